#!/bin/bash
# usage: tools/exp/ab_gki_lib_forward.sh "<bench_forward args>" libA.so libB.so ... -> tools/bench_forward.py with each build (GKI_LIB;
# "product" = in-tree), cycling three times on one box.  Never touches the product library.
R="$(pwd)"; ARGS="$1"; shift
for i in 1 2 3; do
  for lib in "$@"; do
    if [ "$lib" = product ]; then unset GKI_LIB; else export GKI_LIB="$R/$lib"; fi
    timeout -k 10 300 python3 "$R/tools/bench_forward.py" $ARGS 2>/dev/null \
     | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%-34s [%s] %.3f ms/step  %.3g starts/s  %.3g records/s  (%d starts, %d records)' % ('$lib', '$ARGS', d['ms_per_step'], d['value'], d['records_per_s'], d['config']['start_positions'], d['config']['records']))"
  done
done
