#!/bin/bash
# usage (GPU box, repo root): tools/exp/grouped_trace.sh <tag> [lib.so] [n] [modulo] [group_bits] -> grouped vs plain slice build + per-kernel averages
tag="$1"; lib="${2:-product}"; n="${3:-395000000}"; mod="${4:-56616313}"; g="${5:-7}"
R="$(pwd)"; export TMPDIR=/tmp; mkdir -p "$R/gpurun_out/r4"
if [ "$lib" = product ]; then unset GKI_LIB; else export GKI_LIB="$R/$lib"; fi
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r4/grp_$tag" -- python3 "$R/tools/exp/grouped_build_time.py" "$n" "$mod" "$g" 3 > "$R/gpurun_out/r4/grp_$tag.json" 2> "$R/gpurun_out/r4/grp_$tag.err"
rc=$?
cd "$R"
[ $rc -ne 0 ] && { echo "== $tag FAILED rc=$rc"; tail -5 "gpurun_out/r4/grp_$tag.err"; exit $rc; }
python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
print("==", tag, open("gpurun_out/r4/grp_%s.json" % tag).read().strip())
f = glob.glob("gpurun_out/r4/grp_%s/*/*kernel_stats.csv" % tag)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) < 0.3: break
    print("  %-64s x%-3s %8.3f ms" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:64], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
