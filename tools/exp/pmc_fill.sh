#!/bin/bash
# usage: tools/exp/pmc_fill.sh "<bench args A>" "<bench args B>" ...  -> one rocprofv3 --pmc pass per argument set:
# wave-cycles, busy-cycles, waiting and instruction activity of the boundary kernels.  wave-cycles / busy-cycles is the
# average number of resident waves (arbitrary unit, comparable between runs): a kernel whose duration grows much faster
# than its wave-cycles is waiting for a few long waves.
set -u
R="$(pwd)"; export TMPDIR=/tmp
cd /tmp
i=0
for extra in "$@"; do
  i=$((i+1)); d="$R/gpurun_out/fill_$i"
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d "$d" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --reads 0 $extra > "$d.json" 2> "$d.err" || { echo "pass [$extra] failed"; tail -5 "$d.err"; exit 1; }
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
    print("   %-46s waves %6d  wave-cyc %.3g  busy %.3g  resident %.1f  wait %.0f%%  VALU %.3g SALU %.3g VMEM_RD %.3g LDS %.3g" % (
        k, m["SQ_WAVES"], m["SQ_WAVE_CYCLES"], m["SQ_BUSY_CYCLES"], m["SQ_WAVE_CYCLES"] / m["SQ_BUSY_CYCLES"],
        100 * m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_INSTS_VALU"], m["SQ_INSTS_SALU"], m["SQ_INSTS_VMEM_RD"], m["SQ_INSTS_LDS"]))
PY
done
