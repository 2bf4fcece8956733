#!/bin/bash
# usage: tools/exp/ab_libs_forward.sh "<bench_forward args>" a.so b.so ...  -> tools/bench_forward.py with each product
# build, cycling three times on one box.
set -u
R="$(pwd)"; L="$R/graph_kmer_index_amd/libgki_hip.so"; ARGS="$1"; shift
cp "$L" /tmp/gki_keep.so
for i in 1 2 3; do
  for lib in "$@"; do
    cp "$lib" "$L"
    timeout -k 10 300 python3 "$R/tools/bench_forward.py" $ARGS 2>/dev/null \
     | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%-22s [%s] %.3f ms/step  %.3g starts/s  %.3g records/s  (%d starts, %d records)' % ('$(basename $lib)', '$ARGS', d['ms_per_step'], d['value'], d['records_per_s'], d['config']['start_positions'], d['config']['records']))"
  done
done
cp /tmp/gki_keep.so "$L"
