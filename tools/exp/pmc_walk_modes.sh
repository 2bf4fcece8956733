#!/bin/bash
# usage: tools/exp/pmc_walk_modes.sh  -> SQ counters of k_emit_boundary_one with the expansion skipped (walk alone),
# one-node mode against all-nodes mode on the same box.  Needs `make tuning` (libgki_hip_tuning.so); on the box the
# tuning build is copied over libgki_hip.so (the box's copy of the repo is scratch).  rocprofv3 --pmc passes only.
set -u
R="$(pwd)"; export TMPDIR=/tmp
cp "$R/graph_kmer_index_amd/libgki_hip_tuning.so" "$R/graph_kmer_index_amd/libgki_hip.so" || exit 1
export GKI_DBG_SKIP_EXPAND=1
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
B="SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD"
cd /tmp
for mode in one all; do
  extra=""; [ "$mode" = all ] && extra="--all-nodes"
  i=0
  for grp in "$A" "$B"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d "$R/gpurun_out/walk_${mode}_$i" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --reads 0 $extra > "$R/gpurun_out/walk_${mode}_$i.json" 2> "$R/gpurun_out/walk_${mode}_$i.err" || { echo "pass $mode $i failed"; exit 1; }
    echo "pass $mode $i done"
  done
done
cd "$R"
python3 - <<'PY'
import csv, glob, collections
for mode in ("one", "all"):
    acc = collections.defaultdict(list); n = 0
    for f in glob.glob("gpurun_out/walk_%s_*/*/*counter_collection.csv" % mode):
        for r in csv.DictReader(open(f)):
            if "k_emit_boundary_one" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== walk alone,", mode + "-node mode (mean per launch)")
    for c, v in sorted(acc.items()):
        print("   %-28s %.5g   (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
