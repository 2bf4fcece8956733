#!/bin/bash
# usage: tools/exp/overlap_emit.sh -> the interior (HBM-bound) and boundary (issue-bound) emit kernels one after the
# other (product order), boundary first, and on two streams (tuning knobs GKI_BOUNDARY_FIRST / GKI_OVERLAP_EMIT);
# tuning build, alternating on one box.  ms_per_step is what counts: per-kernel brackets overlap in the last mode.
set -u
R="$(pwd)"
export GKI_LIB="$R/graph_kmer_index_amd/libgki_hip_tuning.so"     # (the product library is not touched)
run() {  # $1 tag, $2 overlap, $3 boundary_first
  export GKI_OVERLAP_EMIT="$2" GKI_BOUNDARY_FIRST="$3"
  timeout -k 10 200 python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --reads 0 2>/dev/null \
   | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernels_ms_rank0_last_step']; print('%-30s step %.3f ms   (interior %.2f, boundary %.2f)' % ('$1', d['ms_per_step'], k['emit_interior'], k['emit_boundary']))"
}
for i in 1 2; do
  run "sequential (product order)" 0 0
  run "boundary first" 0 1
  run "two streams" 1 0
  run "two streams, boundary first" 1 1
done
