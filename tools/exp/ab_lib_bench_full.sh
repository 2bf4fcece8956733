#!/bin/bash
# usage (GPU box, repo root): tools/exp/ab_lib_bench_full.sh libA.so libB.so [rounds] -> the secondary records of bench.py
# (full_index phases, index_build, early_stop_search) alternating between two builds (GKI_LIB; "product" = in-tree), one box
A="$1"; B="$2"; rounds="${3:-2}"; R="$(pwd)"
for i in $(seq 1 "$rounds"); do
  for lib in "$A" "$B"; do
    if [ "$lib" = product ]; then unset GKI_LIB; else export GKI_LIB="$R/$lib"; fi
    timeout -k 10 400 python3 bench.py --no-cpu-baseline --reads 2e6 --steps 5 --warmup 2 > /tmp/abf_line.json 2> /tmp/abf_line.err || { echo "$lib FAILED"; tail -3 /tmp/abf_line.err; continue; }
    python3 -c "
import json; d = json.loads(open('/tmp/abf_line.json').readline()); f = d['full_index']
print('%-42s step %.2f  full_index %.0f ms (find %.1f, partition %.1f, builds %.1f) ok=%s  index_build %.2f  scalar get %.0f/s  early_stop %.2f ms' % ('$lib', d['ms_per_step'], f['ms'], f['find_ms'], f['partition_ms'], f['build_slices_ms'], f['payload_equals_flat_multiset'], d['index_build']['ms'], d['index_build']['scalar_get_calls_per_s'], d['early_stop_search']['ms']))"
  done
done
