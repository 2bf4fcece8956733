#!/bin/bash
# usage: tools/exp/trace_modes.sh "<bench args A>" "<bench args B>" ...  -> rocprofv3 --kernel-trace --stats of bench.py
# for each argument set (product build), per-kernel calls / average / total of the finder kernels.
set -u
R="$(pwd)"; export TMPDIR=/tmp
cd /tmp
i=0
for extra in "$@"; do
  i=$((i+1)); d="$R/gpurun_out/trace_mode_$i"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --reads 0 $extra > "$d.json" 2> "$d.err" || { echo "trace [$extra] failed"; tail -5 "$d.err"; exit 1; }
  echo "== bench.py $extra   ms_per_step $(python3 -c "import json; print(round(json.loads(open('$d.json').readline())['ms_per_step'], 2))")"
  python3 - "$d" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f)) if any(x in r["Name"] for x in ("k_emit", "k_count", "k_node_emit", "k_block_scan"))]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows:
        print("   %-62s calls %3s  avg %8.3f ms" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:62], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
done
