#!/bin/bash
# usage: tools/exp/lossy_variant_cost.sh -> what the lossy-restart variant of the boundary kernels costs on a graph with
# no such point (it loads lossy[q] from global memory at every predecessor step): default kernels, the lossy variants
# forced (tuning knob GKI_FORCE_LOSSY=1), and the general variants (which always include the lossy logic); one box.
set -u
R="$(pwd)"
cp "$R/graph_kmer_index_amd/libgki_hip_tuning.so" "$R/graph_kmer_index_amd/libgki_hip.so" || exit 1
run() {  # $1 tag, $2 force, $3.. bench args
  local tag="$1"; export GKI_FORCE_LOSSY="$2"; shift 2
  timeout -k 10 200 python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --reads 0 "$@" 2>/dev/null \
   | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernels_ms_rank0_last_step']; print('%-40s count %.3f ms   emit_boundary %.3f ms   step %.2f ms' % ('$tag', k['count_boundary'], k['emit_boundary'], d['ms_per_step']))"
}
for i in 1 2 3; do
  run "default  <lossy=0, general=0>" 0
  run "forced   <lossy=1, general=0>" 1
  run "general  <lossy=1, general=1>" 0 --general
done
