#!/bin/bash
# usage: tools/exp/ab_all_nodes.sh  -> same-box A/B of two builds of libgki_hip.so on the all-nodes bench
# (tools/exp/_ab/libgki_base.so = the build to compare against; alternates base/new so box drift hits both)
set -u
R="$(pwd)"; L="$R/graph_kmer_index_amd/libgki_hip.so"
cp "$L" /tmp/gki_new.so
run() {  # $1 = tag
  timeout -k 10 200 python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --reads 0 --all-nodes 2>/dev/null \
   | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernels_ms_rank0_last_step']; print('$1', round(d['ms_per_step'],2), {a: round(b,2) for a,b in k.items()})"
}
for i in 1 2 3; do
  cp "$R/tools/exp/_ab/libgki_base.so" "$L"; run base
  cp /tmp/gki_new.so "$L"; run new
done
cp /tmp/gki_new.so "$L"
