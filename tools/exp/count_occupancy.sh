#!/bin/bash
# usage: tools/exp/count_occupancy.sh -> does occupancy explain the general count kernel?  The default count kernel
# (53 VGPRs, 8 waves/SIMD) padded with unused dynamic LDS down to 6 and 4 waves/SIMD, next to the general variant
# (105 VGPRs, 4 waves/SIMD) on the same graph; tuning build, alternating, one box.
set -u
R="$(pwd)"
cp "$R/graph_kmer_index_amd/libgki_hip_tuning.so" "$R/graph_kmer_index_amd/libgki_hip.so" || exit 1
run() {  # $1 tag, $2 pad, $3.. bench args
  local tag="$1"; export GKI_CNT_LDS_PAD="$2"; shift 2
  timeout -k 10 200 python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --reads 0 "$@" 2>/dev/null \
   | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernels_ms_rank0_last_step']; print('%-46s count_boundary %.3f ms' % ('$tag', k['count_boundary']))"
}
for i in 1 2 3; do
  run "default kernel, 8 waves/SIMD (no pad)" 0
  run "default kernel, 6 waves/SIMD (pad 8000 B)" 8000
  run "default kernel, 4 waves/SIMD (pad 17000 B)" 17000
  run "general kernel, 4 waves/SIMD (105 VGPRs)" 0 --general
done
