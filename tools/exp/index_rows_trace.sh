#!/bin/bash
# usage (GPU box, repo root): tools/exp/index_rows_trace.sh <outdir-tag> <n> <modulo> lib1.so lib2.so ...
# The row-carrying index build on n random device-resident records per library (GKI_LIB; "product" = libgki_hip.so), one
# after the other on one box, under rocprofv3 --kernel-trace for the per-kernel split.  Dense slices of a whole-genome
# index: n = 395000000, modulo = 56616313 (7 records per bucket); the variant index: 310000000 452930477.
tag0="$1"; n="$2"; mod="$3"; shift 3
R="$(pwd)"; export TMPDIR=/tmp
mkdir -p "$R/gpurun_out/r4"
for lib in "$@"; do
  tag="${tag0}_$(basename "$lib" .so)"
  if [ "$lib" = product ]; then unset GKI_LIB; else export GKI_LIB="$R/$lib"; fi
  cd /tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r4/$tag" -- python3 "$R/tools/exp/index_forms_time.py" "$n" "$mod" 3 rows > "$R/gpurun_out/r4/$tag.json" 2> "$R/gpurun_out/r4/$tag.err" || { echo "== $tag FAILED"; tail -3 "$R/gpurun_out/r4/$tag.err"; cd "$R"; continue; }
  cd "$R"
  python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
print("==", tag, open("gpurun_out/r4/%s.json" % tag).read().strip())
f = glob.glob("gpurun_out/r4/%s/*/*kernel_stats.csv" % tag)[0]
for r in csv.DictReader(open(f)):
    if any(x in r["Name"] for x in ("k_partition", "k_group", "k_digit", "k_kmer_digit", "k_bucket_keys_hist", "k_block_scan<unsigned int, long, unsigned int>")):
        print("  %-60s x%-3s %8.3f ms" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
done
