#!/bin/bash
# HBM traffic of the index build's kernels (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes), on the two inputs DESIGN.md 4.3 quotes: the variant index (3.1e8 random
# records, default modulo) and one dense slice of the whole-genome index (3.95e8 records, 7 per bucket).
# usage (GPU box, repo root): tools/collect_pmc_index.sh <outdir>   -> <outdir>/pmc_index.json
set -o pipefail
out="$1"; R="$(pwd)"; export TMPDIR=/tmp; mkdir -p "$R/$out"
cd /tmp
for shape in "sparse 310000000 452930477" "dense 395000000 56616313"; do
  set -- $shape
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d "$R/$out/$1_$c" -- python3 "$R/tools/exp/index_forms_time.py" "$2" "$3" 1 rows > "$R/$out/$1_$c.json" 2> "$R/$out/$1_$c.err" || echo "pass $1 $c failed"
  done
done
cd "$R"
python3 - "$out" <<'PY'
import csv, glob, json, re, sys, collections, os
out = sys.argv[1]
res = {}
for shape, n in (("sparse", 310000000), ("dense", 395000000)):
    per = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("%s/%s_%s/*/*counter_collection.csv" % (out, shape, c)):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
            for k, v in acc.items():
                m = re.search(r"(k_\w+)(<[^>]*>)?", k)
                if m and any(x in k for x in ("k_partition", "k_group", "k_digit", "k_kmer_digit", "k_block_scan", "k_block_sums")):
                    per[m.group(0)][c] = {"per_launch_avg_KB": sum(v) / len(v), "launches": len(v)}
    kernels = {}
    read = written = 0.0
    for k, d in per.items():
        r2 = 2 * d.get("FETCH_SIZE", {}).get("per_launch_avg_KB", 0.0) * 1024      # gfx950: FETCH_SIZE counts streaming reads at half
        w = d.get("WRITE_SIZE", {}).get("per_launch_avg_KB", 0.0) * 1024
        # kernels launched once per build: their per-launch average is their share of one build; the scans run twice per pass
        launches_per_build = max(1, round(d.get("WRITE_SIZE", d.get("FETCH_SIZE"))["launches"] / 2))     # index_forms_time builds twice (1 warm + 1)
        kernels[k] = {"read_x2_GB_per_launch": round(r2 / 1e9, 3), "written_GB_per_launch": round(w / 1e9, 3), "launches_per_build": launches_per_build}
        read += r2 * launches_per_build; written += w * launches_per_build
    res[shape] = {"records": n, "kernels": kernels, "traffic_bytes_per_build": read + written,
                  "bytes_per_record": round((read + written) / n, 1)}
res["commit"] = os.environ.get("GKI_COMMIT", "unknown")
res["command"] = "tools/collect_pmc_index.sh (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM)"
json.dump(res, open("%s/pmc_index.json" % out, "w"), indent=1)
for shape in ("sparse", "dense"):
    print(shape, "traffic per build %.1f GB = %.0f B/record" % (res[shape]["traffic_bytes_per_build"] / 1e9, res[shape]["bytes_per_record"]))
    for k, d in res[shape]["kernels"].items():
        print("   %-52s read %.2f  written %.2f GB  x%d" % (k[:52], d["read_x2_GB_per_launch"], d["written_GB_per_launch"], d["launches_per_build"]))
PY
