#!/bin/bash
# usage: tools/exp/pmc_flat.sh "<bench args>" ...  -> executed FLAT instructions of the boundary kernels next to their
# vector-memory reads (one rocprofv3 --pmc pass per argument set): is the out-of-line history_ok, the only code with
# FLAT loads left, hot?
set -u
R="$(pwd)"; export TMPDIR=/tmp
cd /tmp
i=0
for extra in "$@"; do
  i=$((i+1)); d="$R/gpurun_out/flat_$i"
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES --output-format csv -d "$d" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --reads 0 $extra > "$d.json" 2> "$d.err" || { echo "pass [$extra] failed"; tail -3 "$d.err"; exit 1; }
  echo "== bench.py $extra"
  python3 - "$d" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_emit_boundary_one" in k or "k_count_boundary" in k:
            acc[k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print("   %-46s FLAT %.3g   VMEM_RD %.3g   VMEM_WR %.3g   LDS %.3g   FLAT_LDS_ONLY %.3g" % (k, m["SQ_INSTS_FLAT"], m["SQ_INSTS_VMEM_RD"], m["SQ_INSTS_VMEM_WR"], m["SQ_INSTS_LDS"], m.get("SQ_INSTS_FLAT_LDS_ONLY", float("nan"))))
PY
done
