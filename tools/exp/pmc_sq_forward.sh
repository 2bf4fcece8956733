#!/bin/bash
# usage (GPU box, repo root): tools/exp/pmc_sq_forward.sh <tag> "<bench_forward args>" -> SQ / L2 counters of the early-stop search's kernels
# (separate rocprofv3 --pmc passes, no trace domains; SQ counters only: a pass asking for TCC_* / TCP_* names hung for its whole
# timeout on this pool)
tag="$1"; args="$2"
R="$(pwd)"; export TMPDIR=/tmp
cd /tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
B="SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD"


i=0
for grp in "$A" "$B"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d "$R/gpurun_out/sqf_${tag}_$i" -- python3 "$R/tools/bench_forward.py" $args > "$R/gpurun_out/sqf_${tag}_$i.json" 2> "$R/gpurun_out/sqf_${tag}_$i.err" || { echo "pass $i failed"; tail -3 "$R/gpurun_out/sqf_${tag}_$i.err"; }
done
cd "$R"
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/sqf_%s_*/*/*counter_collection.csv" % tag):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_forward" in k:
            acc[k.replace("(anonymous namespace)::", "").split("(")[0][-48:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print("==", tag, k)
    for c, v in sorted(d.items()):
        print("   %-32s %.4g   (x%d)" % (c, sum(v) / len(v), len(v)))
PY
