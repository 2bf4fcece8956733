#!/bin/bash
# usage (GPU box, repo root): tools/exp/forward_trace.sh <tag> "<bench_forward args>" [lib.so] -> per-kernel averages of the early-stop search
tag="$1"; args="$2"; lib="${3:-product}"
R="$(pwd)"; export TMPDIR=/tmp; mkdir -p "$R/gpurun_out/r4"
if [ "$lib" = product ]; then unset GKI_LIB; else export GKI_LIB="$R/$lib"; fi
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r4/fwd_$tag" -- python3 "$R/tools/bench_forward.py" $args > "$R/gpurun_out/r4/fwd_$tag.json" 2> "$R/gpurun_out/r4/fwd_$tag.err"
rc=$?
cd "$R"
[ $rc -ne 0 ] && { echo "== $tag FAILED rc=$rc"; tail -5 "gpurun_out/r4/fwd_$tag.err"; exit $rc; }
python3 - "$tag" <<'PY'
import csv, glob, json, sys
tag = sys.argv[1]
d = json.loads(open("gpurun_out/r4/fwd_%s.json" % tag).readline())
print("==", tag, "%.3f ms/step, %d starts, %d records" % (d["ms_per_step"], d["config"]["start_positions"], d["config"]["records"]))
f = glob.glob("gpurun_out/r4/fwd_%s/*/*kernel_stats.csv" % tag)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) < 0.3: break
    print("  %-72s x%-3s %8.3f ms" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:72], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
