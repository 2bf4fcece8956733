#!/bin/bash
# usage (GPU box, repo root): tools/exp/ab_lib_bench.sh "<bench args>" libA.so libB.so [rounds]  -> bench.py alternating between two
# builds (GKI_LIB; "product" = the in-tree library), same box: ms_per_step and the finder kernels' times of every run
args="$1"; A="$2"; B="$3"; rounds="${4:-3}"; R="$(pwd)"
for i in $(seq 1 "$rounds"); do
  for lib in "$A" "$B"; do
    if [ "$lib" = product ]; then unset GKI_LIB; else export GKI_LIB="$R/$lib"; fi
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --reads 0 --steps 10 --warmup 3 $args > /tmp/ab_line.json 2> /tmp/ab_line.err || { echo "$lib FAILED"; tail -3 /tmp/ab_line.err; continue; }
    python3 -c "
import json; d = json.loads(open('/tmp/ab_line.json').readline()); k = d['kernels_ms_rank0_last_step']
print('%-45s step %.3f ms   count %.3f  interior %.3f  boundary %.3f  scans %.3f' % ('$lib', d['ms_per_step'], k['count_boundary'], k['emit_interior'], k['emit_boundary'], k['setup_scans']))"
  done
done
