#!/bin/bash
# usage: tools/exp/ab_libs.sh "<bench args>" <lib1.so> <lib2.so> ...  -> same-box comparison of several product builds on
# bench.py with the given arguments, cycling through the builds three times: count / emit_boundary / step.
set -u
R="$(pwd)"; L="$R/graph_kmer_index_amd/libgki_hip.so"; ARGS="$1"; shift
cp "$L" /tmp/gki_keep.so
for i in 1 2 3; do
  for lib in "$@"; do
    cp "$lib" "$L"
    timeout -k 10 200 python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --reads 0 $ARGS 2>/dev/null \
     | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernels_ms_rank0_last_step']; print('%-28s [%s] count %.3f ms   emit_boundary %.3f ms   step %.2f ms   records %d' % ('$(basename $lib)', '$ARGS', k['count_boundary'], k['emit_boundary'], d['ms_per_step'], d['config']['records_per_step']))"
  done
done
cp /tmp/gki_keep.so "$L"
