#!/bin/bash
# usage: tools/exp/sweep.sh "<bench args>" "ENV1=.. ENV2=.." "ENV1=.. " ...
args="$1"; shift
for envs in "$@"; do
  r=$(env $envs timeout -k 5 200 python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); k=d['kernels_ms_rank0_last_step']; print('%.1f G/s  %.3f ms/step  cnt %.3f int %.3f bnd %.3f scan %.3f' % (d['value']/1e9, d['ms_per_step'], k['count_boundary'], k['emit_interior'], k['emit_boundary'], k['setup_scans']))")
  echo "$envs => $r"
done
