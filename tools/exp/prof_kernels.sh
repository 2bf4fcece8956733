#!/bin/bash
# usage: tools/exp/prof_kernels.sh <tag> "<bench args>" [ENV=VAL ...]   -> per-kernel average ms of one profiled bench run
tag="$1"; args="$2"; shift 2
R="$(pwd)"; export TMPDIR=/tmp
for e in "$@"; do export "$e"; done
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_$tag" -- python3 "$R/bench.py" $args --no-cpu-baseline > "$R/gpurun_out/prof_$tag.json" 2> "$R/gpurun_out/prof_$tag.err"
cd "$R"
python3 - "$tag" <<'PY'
import csv, glob, sys, json
tag = sys.argv[1]
f = glob.glob("gpurun_out/prof_%s/*/*kernel_stats.csv" % tag)[0]
print("==", tag, json.load(open("gpurun_out/prof_%s.json" % tag))["ms_per_step"], "ms/step")
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) < 0.8: break
    print("  %-45s x%-3s %8.3f ms" % (r["Name"].replace("(anonymous namespace)::", "")[:45], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
