#!/bin/bash
# usage: tools/exp/bnd_grid.sh -> k_emit_boundary_one's grid: the product's 2048 resident-and-a-half workgroups walking the
# node groups grid-stride, against one workgroup per g groups of 256 nodes with no cap (GKI_BND_BLOCKS=-g, tuning build
# through GKI_LIB), on the whole graph and on one rank's shard of eight.
set -u
R="$(pwd)"; export GKI_LIB="$R/graph_kmer_index_amd/libgki_hip_tuning.so"
run() {  # $1 tag, $2 knob, $3.. bench args
  local tag="$1"; export GKI_BND_BLOCKS="$2"; shift 2
  timeout -k 10 200 python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --reads 0 --no-full-index "$@" 2>/dev/null \
   | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernels_ms_rank0_last_step']; print('%-40s step %.3f ms   boundary %.3f' % ('$tag', d['ms_per_step'], k['emit_boundary']))"
}
for i in 1 2; do
  for g in 0 -1 -2 -4 -8; do
    run "whole graph, knob $g" $g
    run "shard 3/8, knob $g" $g --pretend-shard 3/8
    run "whole graph, all nodes, knob $g" $g --all-nodes
  done
done
