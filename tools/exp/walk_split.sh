#!/bin/bash
# usage (GPU box, repo root; needs `make tuning`): tools/exp/walk_split.sh -> k_emit_boundary_one's time with and without the
# expansion, both modes, alternating on one box (tuning build selected through GKI_LIB; the product library is not touched)
set -u
R="$(pwd)"; export GKI_LIB="$R/graph_kmer_index_amd/libgki_hip_tuning.so"
run() {  # $1 tag, $2 knob value, $3.. bench args
  local tag="$1"; export GKI_DBG_SKIP_EXPAND="$2"; shift 2
  timeout -k 10 200 python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --reads 0 "$@" 2>/dev/null \
   | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernels_ms_rank0_last_step']; print('%-32s emit_boundary %.2f ms   (count %.2f, interior %.2f)' % ('$tag', k['emit_boundary'], k['count_boundary'], k['emit_interior']))"
}
for i in 1 2; do
  run "one-node walk alone" 1
  run "all-nodes walk alone" 1 --all-nodes
  run "all-nodes walk, no node lists" 3 --all-nodes
  run "all-nodes, stores folded into L2" 5 --all-nodes
  run "one-node, stores folded into L2" 5
  run "all-nodes whole kernel" 0 --all-nodes
  run "one-node whole kernel" 0
done
