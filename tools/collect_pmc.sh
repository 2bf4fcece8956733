#!/bin/bash
# Collect HBM traffic counters of the bench kernels with rocprofv3 (separate --pmc passes, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage (on the GPU box, from the repo root): tools/collect_pmc.sh <outdir> [bench args...]
set -o pipefail
out="$1"; shift
R="$(pwd)"
export TMPDIR=/tmp
mkdir -p "$R/$out"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d "$R/$out/$c" -- python3 "$R/bench.py" "$@" --no-cpu-baseline > "$R/$out/$c.json" 2> "$R/$out/$c.err" || echo "pass $c failed"
done
cd "$R"
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("%s/%s/*/*counter_collection.csv" % (out, c)):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for k, v in per.items():
            res[k][c] = {"per_launch_avg_KB": sum(v) / len(v), "launches": len(v)}
json.dump(res, open("%s/summary.json" % out, "w"), indent=1)
for k, d in res.items():
    if "emit" in k or "count_b" in k:
        print(k[:70], {c: round(x["per_launch_avg_KB"] / 1e6, 3) for c, x in d.items()}, "GB")
PY
