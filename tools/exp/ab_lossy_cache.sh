#!/bin/bash
# usage: tools/exp/ab_lossy_cache.sh -> same-box A/B of two TUNING builds (tools/exp/_ab/libgki_base_tuning.so against
# graph_kmer_index_amd/libgki_hip_tuning.so) with the lossy-restart variants forced on the default graph
# (GKI_FORCE_LOSSY=1) and on the general variants (--general); alternating.
set -u
R="$(pwd)"; L="$R/graph_kmer_index_amd/libgki_hip.so"
cp "$L" /tmp/gki_product_keep.so
run() {  # $1 tag, $2 lib, $3 force, $4.. bench args
  local tag="$1"; cp "$2" "$L"; export GKI_FORCE_LOSSY="$3"; shift 3
  timeout -k 10 200 python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --reads 0 "$@" 2>/dev/null \
   | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernels_ms_rank0_last_step']; print('%-34s count %.3f ms   emit_boundary %.3f ms   step %.2f ms' % ('$tag', k['count_boundary'], k['emit_boundary'], d['ms_per_step']))"
}
BASE="$R/tools/exp/_ab/libgki_base_tuning.so"; NEW="$R/graph_kmer_index_amd/libgki_hip_tuning.so"
for i in 1 2 3; do
  run "base  forced lossy variants" "$BASE" 1
  run "new   forced lossy variants" "$NEW" 1
  run "base  --general" "$BASE" 0 --general
  run "new   --general" "$NEW" 0 --general
  run "new   default (no lossy)" "$NEW" 0
done
cp /tmp/gki_product_keep.so "$L"
