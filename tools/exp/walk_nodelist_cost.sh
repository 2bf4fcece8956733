#!/bin/bash
# usage: tools/exp/walk_nodelist_cost.sh -> what the node-list construction (rank sort over path[]) costs the all-nodes
# walk: k_emit_boundary_one with the expansion skipped, (a) one-node mode, (b) all-nodes mode, (c) all-nodes mode with
# no node lists built (GKI_DBG_SKIP_EXPAND=3), alternating on one box; then the instruction counters of (c).
# Needs `make tuning`; copies the tuning build over libgki_hip.so in the box's scratch copy of the repo.
set -u
R="$(pwd)"; export TMPDIR=/tmp
cp "$R/graph_kmer_index_amd/libgki_hip_tuning.so" "$R/graph_kmer_index_amd/libgki_hip.so" || exit 1
run() {  # $1 tag, $2 knob value, $3.. bench args
  local tag="$1"; export GKI_DBG_SKIP_EXPAND="$2"; shift 2
  timeout -k 10 200 python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --reads 0 "$@" 2>/dev/null \
   | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernels_ms_rank0_last_step']; print('%-28s emit_boundary %.2f ms   (count %.2f, interior %.2f)' % ('$tag', k['emit_boundary'], k['count_boundary'], k['emit_interior']))"
}
for i in 1 2 3; do
  run "one-node walk alone" 1
  run "all-nodes walk alone" 1 --all-nodes
  run "all-nodes walk, no node lists" 3 --all-nodes
  run "all-nodes walk, no >NLQ windows" 4 --all-nodes
  run "all-nodes whole kernel" 0 --all-nodes
  run "one-node whole kernel" 0
done
export GKI_DBG_SKIP_EXPAND=3
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d "$R/gpurun_out/walk_all3" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --reads 0 --all-nodes > "$R/gpurun_out/walk_all3.json" 2> "$R/gpurun_out/walk_all3.err" || { echo "pmc pass failed"; exit 1; }
cd "$R"
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/walk_all3/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_emit_boundary_one" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== all-nodes walk alone, no node lists (mean per launch)")
for c, v in sorted(acc.items()):
    print("   %-22s %.5g" % (c, sum(v) / len(v)))
PY
