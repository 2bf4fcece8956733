#!/bin/bash
# usage: tools/exp/ab_tuning_env.sh "VAR=val <bench args>" ...  -> same-box A/B of two TUNING builds
# (tools/exp/_ab/libgki_base_tuning.so against graph_kmer_index_amd/libgki_hip_tuning.so) per case; a case is an optional
# leading environment assignment (a tuning knob) followed by bench.py arguments.  Alternating, three rounds.
set -u
R="$(pwd)"; L="$R/graph_kmer_index_amd/libgki_hip.so"
cp "$L" /tmp/gki_product_keep.so
run() {  # $1 tag, $2 lib, $3 case
  local tag="$1" lib="$2" c="$3" envs="" args=""
  for w in $c; do case "$w" in *=*) envs="$envs $w";; *) args="$args $w";; esac; done
  cp "$lib" "$L"
  env $envs timeout -k 10 200 python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --reads 0 $args 2>/dev/null \
   | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernels_ms_rank0_last_step']; print('%-5s [%-38s] count %.3f ms   emit_boundary %.3f ms   step %.2f ms' % ('$tag', '$c', k['count_boundary'], k['emit_boundary'], d['ms_per_step']))"
}
for i in 1 2 3; do
  for c in "$@"; do
    run base "$R/tools/exp/_ab/libgki_base_tuning.so" "$c"
    run new "$R/graph_kmer_index_amd/libgki_hip_tuning.so" "$c"
  done
done
cp /tmp/gki_product_keep.so "$L"
