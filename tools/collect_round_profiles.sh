#!/bin/bash
# usage (on the GPU box, repo root): tools/collect_round_profiles.sh <tag e.g. r02>  -> regenerates the round's headline
# artefacts under gpurun_out/<tag>_final/ from the CURRENT binary, each the way profiles/README.md says:
#   1. rocprofv3 --kernel-trace --stats -- python3 bench.py            -> bench line + kernel stats
#   2. tools/collect_pmc.sh (FETCH_SIZE / WRITE_SIZE, separate passes)  -> pmc json (via tools/make_pmc_profile.py)
#   3. tools/exp/pmc_sq.sh                                              -> SQ counters of the finder kernels
# Copy what you want judged into profiles/ afterwards.
set -u
tag="$1"; R="$(pwd)"; O="$R/gpurun_out/${tag}_final"; mkdir -p "$O"; export TMPDIR=/tmp
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python3 "$R/bench.py" > "$O/bench.json" 2> "$O/bench.err" || { echo "bench under rocprofv3 failed"; tail -5 "$O/bench.err"; exit 1; }
cd "$R"
cp "$O"/trace/*/*kernel_stats.csv "$O/kernel_stats.csv"
python3 - "$O" <<'PY'
import csv, json, sys
o = sys.argv[1]
d = json.loads(open(o + "/bench.json").readline())
print("bench: %.2f ms/step, %.3g k-mers/s, roofline frac %.3f (of measured ceiling %.3f), index_build %.1f ms (reverse %.1f), full_index %.0f ms, read_mapping %.3g k-mers/s (%d reads), early_stop %.3g starts/s" % (
    d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["frac_of_ceiling"], d["index_build"]["ms"], d["index_build"]["reverse_index"]["ms"], d["full_index"]["ms"], d["read_mapping"]["kmers_per_s"], d["read_mapping"]["reads"], d["early_stop_search"]["start_positions_per_s"]))
rows = list(csv.DictReader(open(o + "/kernel_stats.csv")))
keep = [r for r in rows if any(x in r["Name"] for x in ("k_radix", "k_gather_rows", "k_pack_rows", "k_frequencies", "k_directory", "k_bucket_keys", "k_probe", "k_get_small", "k_random_loads", "k_reverse", "k_forward", "k_partition_rows", "k_group_", "k_digit_hist", "k_part_ids", "k_simulate_reads", "k_store_columns", "k_walk_", "k_cls_"))]
with open(o + "/index_kernel_stats.csv", "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:8]:
    print("   %-60s calls %4s avg %8.3f ms" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
tools/collect_pmc.sh "gpurun_out/${tag}_final/pmc" --steps 2 --warmup 1 --reads 0 > "$O/pmc.log" 2>&1 || echo "pmc collection failed"
python3 tools/make_pmc_profile.py "gpurun_out/${tag}_final/pmc" "$O/pmc_3gbp.json" && python3 -c "
import json; d = json.load(open('$O/pmc_3gbp.json')); print('pmc per kernel (GB):', {k[:28]: v for k, v in d['per_kernel_traffic_GB'].items()})"
tools/exp/pmc_sq.sh "${tag}final" > "$O/sq_counters_3gbp.txt" 2>&1 || echo "sq counters failed"
tail -3 "$O/sq_counters_3gbp.txt"
