#!/bin/bash
# usage: tools/exp/pmc_sq.sh <tag>  -> SQ counters of the finder kernels for one bench run (two rocprofv3 --pmc passes)
tag="$1"; shift
R="$(pwd)"; export TMPDIR=/tmp
cd /tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
B="SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD"
i=0
for grp in "$A" "$B"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d "$R/gpurun_out/sq_${tag}_$i" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --reads 0 "$@" > "$R/gpurun_out/sq_${tag}_$i.json" 2> "$R/gpurun_out/sq_${tag}_$i.err" || echo "pass $i failed"
done
cd "$R"
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/sq_%s_*/*/*counter_collection.csv" % tag):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_emit_boundary_one" in k or "k_count_boundary" in k or "k_emit_interior_runs" in k:
            acc[k.replace("(anonymous namespace)::", "").split("(")[0][-48:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print("==", tag, k)
    for c, v in sorted(d.items()):
        print("   %-32s %.4g" % (c, sum(v) / len(v)))
PY
