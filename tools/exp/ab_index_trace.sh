#!/bin/bash
# usage: tools/exp/ab_index_trace.sh -> same-box A/B of two product builds (tools/exp/_ab/libgki_base.so against the built
# library) on the index build inside bench.py: index_build ms of the bench line (unprofiled) and, from a rocprofv3 kernel
# trace, the average of the radix kernels.  Alternating, two rounds.
set -u
R="$(pwd)"; L="$R/graph_kmer_index_amd/libgki_hip.so"; export TMPDIR=/tmp
cp "$L" /tmp/gki_new_keep.so
one() {  # $1 tag, $2 lib
  cp "$2" "$L"
  timeout -k 10 300 python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null \
   | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1  index_build %.2f ms   reverse_index %.2f ms' % (d['index_build']['ms'], d['index_build']['reverse_index']['ms']))"
}
trace() {  # $1 tag, $2 lib
  cp "$2" "$L"; local d="$R/gpurun_out/idx_trace_$1"
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$d.json" 2> "$d.err") || { echo "trace $1 failed"; return; }
  python3 - "$d" "$1" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_radix_hist" in r["Name"] or "k_radix_scatter" in r["Name"]:
            print("%s  %-18s calls %3s avg %.3f ms" % (sys.argv[2], r["Name"].split("(")[0].replace("(anonymous namespace)::", "")[-18:], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
}
for i in 1 2; do one base "$R/tools/exp/_ab/libgki_base.so"; one new /tmp/gki_new_keep.so; done
trace base "$R/tools/exp/_ab/libgki_base.so"; trace new /tmp/gki_new_keep.so
cp /tmp/gki_new_keep.so "$L"
