#!/bin/bash
# usage: tools/exp/emit_occupancy.sh -> how much the default boundary emit kernel (LDS-limited to 4 workgroups per CU)
# depends on its occupancy: padded with unused dynamic LDS (tuning knob GKI_EMIT_LDS_PAD) down to 3 and 2 workgroups per
# CU; tuning build, alternating, one box.  Grid = 2048 workgroups: 2.0 / 2.67 / 4.0 rounds at 4 / 3 / 2 per CU.
set -u
R="$(pwd)"
cp "$R/graph_kmer_index_amd/libgki_hip_tuning.so" "$R/graph_kmer_index_amd/libgki_hip.so" || exit 1
run() {  # $1 tag, $2 pad
  export GKI_EMIT_LDS_PAD="$2"
  timeout -k 10 200 python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --reads 0 2>/dev/null \
   | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernels_ms_rank0_last_step']; print('%-40s emit_boundary %.3f ms' % ('$1', k['emit_boundary']))"
}
for i in 1 2 3; do
  run "4 workgroups per CU (no pad)" 0
  run "3 workgroups per CU (pad 1000 B)" 1000
  run "2 workgroups per CU (pad 15000 B)" 15000
done
