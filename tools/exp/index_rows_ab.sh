#!/bin/bash
# usage (GPU box, repo root): tools/exp/index_rows_ab.sh <n> <modulo> lib1.so lib2.so ...  -> row-carrying build time + its
# kernels per library (GKI_LIB), same box, one after the other; "product" = graph_kmer_index_amd/libgki_hip.so
n="$1"; mod="$2"; shift 2
R="$(pwd)"; export TMPDIR=/tmp
for lib in "$@"; do
  tag="$(basename "$lib" .so)"
  if [ "$lib" = product ]; then unset GKI_LIB; else export GKI_LIB="$R/$lib"; fi
  cd /tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r3/ab_$tag" -- python3 "$R/tools/exp/index_forms_time.py" "$n" "$mod" 3 rows > "$R/gpurun_out/r3/ab_$tag.json" 2> "$R/gpurun_out/r3/ab_$tag.err" || { echo "== $tag FAILED"; tail -3 "$R/gpurun_out/r3/ab_$tag.err"; cd "$R"; continue; }
  cd "$R"
  python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
print("==", tag, open("gpurun_out/r3/ab_%s.json" % tag).read().strip())
f = glob.glob("gpurun_out/r3/ab_%s/*/*kernel_stats.csv" % tag)[0]
for r in csv.DictReader(open(f)):
    if any(x in r["Name"] for x in ("k_partition", "k_group", "k_digit", "k_bucket_keys_hist", "k_kmer_digit")):
        print("  %-50s x%-3s %8.3f ms" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:50], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
done
