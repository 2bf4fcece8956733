#!/bin/bash
# usage: tools/exp/ab_modes.sh "<bench args A>" "<bench args B>" ...  -> same-box A/B of two builds
# (tools/exp/_ab/libgki_base.so against graph_kmer_index_amd/libgki_hip.so, both product builds) on bench.py with each
# argument set, alternating base/new, three rounds: count / emit_boundary / step.
set -u
R="$(pwd)"; L="$R/graph_kmer_index_amd/libgki_hip.so"
cp "$L" /tmp/gki_new_keep.so
run() {  # $1 tag, $2 lib, $3.. bench args
  local tag="$1"; cp "$2" "$L"; shift 2
  timeout -k 10 200 python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --reads 0 "$@" 2>/dev/null \
   | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernels_ms_rank0_last_step']; print('%-30s count %.3f ms   emit_boundary %.3f ms   step %.2f ms' % ('$tag', k['count_boundary'], k['emit_boundary'], d['ms_per_step']))"
}
for i in 1 2 3; do
  for extra in "$@"; do
    run "base [$extra]" "$R/tools/exp/_ab/libgki_base.so" $extra
    run "new  [$extra]" /tmp/gki_new_keep.so $extra
  done
done
cp /tmp/gki_new_keep.so "$L"
