#!/bin/bash
# usage: tools/exp/overlap_emit2.sh -> can the boundary emit kernel (latency-bound) run UNDER the interior stream (HBM-bound)?
# The interior kernel holds 5 workgroups x 29 KB of LDS per CU, which leaves no room for a 40 KB boundary workgroup: with
# GKI_RUN_BLOCKS the interior kernel is held to 3 or 4 workgroups per CU and the two kernels are put on two streams
# (GKI_OVERLAP_EMIT).  Tuning build through GKI_LIB, alternating on one box; ms_per_step is what counts.
set -u
R="$(pwd)"; export GKI_LIB="$R/graph_kmer_index_amd/libgki_hip_tuning.so"
run() {  # $1 tag, $2 overlap, $3 interior blocks, $4 boundary first
  export GKI_OVERLAP_EMIT="$2" GKI_RUN_BLOCKS="$3" GKI_BOUNDARY_FIRST="$4"
  timeout -k 10 200 python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --reads 0 2>/dev/null \
   | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernels_ms_rank0_last_step']; print('%-44s step %.3f ms   (interior %.2f, boundary %.2f)' % ('$1', d['ms_per_step'], k['emit_interior'], k['emit_boundary']))"
}
for i in 1 2; do
  run "sequential, 5 interior workgroups per CU" 0 1280 0
  run "sequential, 4" 0 1024 0
  run "sequential, 3" 0 768 0
  run "two streams, 4" 1 1024 0
  run "two streams, 3" 1 768 0
  run "two streams, 3, boundary first" 1 768 1
  run "two streams, 2, boundary first" 1 512 1
done
