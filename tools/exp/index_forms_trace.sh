#!/bin/bash
# usage (GPU box, repo root): tools/exp/index_forms_trace.sh <tag> [lib.so] [n] [modulo] -> timing of both build forms + per-kernel averages
tag="$1"; lib="${2:-}"; n="${3:-310000000}"; mod="${4:-452930477}"
R="$(pwd)"; export TMPDIR=/tmp
[ -n "$lib" ] && export GKI_LIB="$R/$lib"
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r3/idx_$tag" -- python3 "$R/tools/exp/index_forms_time.py" "$n" "$mod" 3 > "$R/gpurun_out/r3/idx_$tag.json" 2> "$R/gpurun_out/r3/idx_$tag.err"
rc=$?
cd "$R"
[ $rc -ne 0 ] && { echo "== $tag FAILED rc=$rc"; tail -5 "gpurun_out/r3/idx_$tag.err"; exit $rc; }
python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
print("==", tag, open("gpurun_out/r3/idx_%s.json" % tag).read().strip())
f = glob.glob("gpurun_out/r3/idx_%s/*/*kernel_stats.csv" % tag)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) < 0.5: break
    print("  %-50s x%-3s %8.3f ms" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:50], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
