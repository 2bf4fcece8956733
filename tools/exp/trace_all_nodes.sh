#!/bin/bash
# usage: tools/exp/trace_all_nodes.sh -> rocprofv3 --kernel-trace --stats of `bench.py --all-nodes`: (a) product build,
# (b) tuning build with the expansion skipped (GKI_DBG_SKIP_EXPAND=1).  Prints per-kernel calls / average / total.
set -u
R="$(pwd)"; export TMPDIR=/tmp
cp "$R/graph_kmer_index_amd/libgki_hip.so" /tmp/gki_product.so
show() {
  python3 - "$1" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:9]:
        print("   %-58s calls %4s  avg %9.3f ms  total %9.2f ms" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:58], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
PY
}
cd /tmp
unset GKI_DBG_SKIP_EXPAND
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/trace_all_product" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --reads 0 --all-nodes > "$R/gpurun_out/trace_all_product.json" 2> "$R/gpurun_out/trace_all_product.err" || { echo "product trace failed"; exit 1; }
echo "== all-nodes, product build"; show "$R/gpurun_out/trace_all_product"
cp "$R/graph_kmer_index_amd/libgki_hip_tuning.so" "$R/graph_kmer_index_amd/libgki_hip.so"
export GKI_DBG_SKIP_EXPAND=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/trace_all_walk" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --reads 0 --all-nodes > "$R/gpurun_out/trace_all_walk.json" 2> "$R/gpurun_out/trace_all_walk.err" || { echo "walk trace failed"; exit 1; }
echo "== all-nodes, expansion skipped (tuning build)"; show "$R/gpurun_out/trace_all_walk"
cp /tmp/gki_product.so "$R/graph_kmer_index_amd/libgki_hip.so"
