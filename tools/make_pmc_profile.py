#!/usr/bin/env python3
"""profiles/rNN_pmc_3gbp.json from the output of tools/collect_pmc.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
passes).  usage: python tools/make_pmc_profile.py <collect_pmc outdir> <out.json> [records_per_launch_interior]
Applies the gfx950 correction of MI355X_MICROARCH.md (section HBM): FETCH_SIZE reports wide coalesced streaming reads at
half their bytes, so it is doubled; WRITE_SIZE is taken as is."""
import json
import os
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
with open(src + "/summary.json") as fh:
    raw = json.load(fh)
kernels = {}
for name, d in raw.items():
    m = re.search(r"(k_\w+)(<[^>]*>)?", name)
    if m:
        kernels[m.group(0)] = d
dom = [k for k in kernels if k.startswith("k_emit_interior_runs")][0]
kb = 1024.0


def bytes_of(k):
    d = kernels[k]
    return 2 * d.get("FETCH_SIZE", {}).get("per_launch_avg_KB", 0.0) * kb, d.get("WRITE_SIZE", {}).get("per_launch_avg_KB", 0.0) * kb


out = {
    "commit": os.environ.get("GKI_COMMIT", "unknown"),      # the source tree the counters were taken from (set by the caller)
    "command": "tools/collect_pmc.sh %s --steps 2 --warmup 1 --reads 0   (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in "
               "separate passes, python3 bench.py default 3 Gbp workload)" % src,
    "units": "per_launch_avg_KB: counter value per launch in KiB as rocprofv3 reports it",
    "gfx950_corrections": "MI355X_MICROARCH.md section HBM: FETCH_SIZE counts wide coalesced streaming reads at half their "
                          "bytes -> doubled for `traffic`; WRITE_SIZE taken as is",
    "kernels": kernels,
    "dominant_kernel": dom,
    "dominant_kernel_traffic_bytes_per_launch": sum(bytes_of(dom)),
    "per_kernel_traffic_GB": {k: {"read_x2": round(bytes_of(k)[0] / 1e9, 3), "written": round(bytes_of(k)[1] / 1e9, 3)}
                              for k in kernels if "emit" in k or "count" in k or "scan" in k or "block_sums" in k},
    "workload": {"n_ref_bases": 3000000000, "n_snp_bubbles": 5000000, "k": 31,
                 "records_per_launch_interior": int(sys.argv[3]) if len(sys.argv) > 3 else 2848565762},
}
with open(dst, "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps(out["per_kernel_traffic_GB"], indent=1))
