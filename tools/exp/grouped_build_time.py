#!/usr/bin/env python3
"""Time the grouped flow of the single-GPU whole-genome index on n random device-resident records of ONE slice-sized index:
partition_by_bucket_range(group_bits=g) + DeviceIndex.build(group_start=...) against the plain DeviceIndex.build, output
checksums compared.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split.
usage: python tools/exp/grouped_build_time.py [n] [modulo] [group_bits] [reps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from graph_kmer_index_amd import _lib
from graph_kmer_index_amd.flat_kmers import DeviceFlatKmers
from graph_kmer_index_amd.collision_free_kmer_index import (DeviceIndex, DeviceRows, PartitionedDeviceIndex, partition_by_bucket_range,
                                                            partition_rows_by_bucket_range)

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 395_000_000
modulo = int(sys.argv[2]) if len(sys.argv) > 2 else 56616313
g = int(sys.argv[3]) if len(sys.argv) > 3 else 7
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
lib = _lib.load(); _lib.require_device()
rng = np.random.default_rng(7)
d = DeviceFlatKmers.allocate(n)
CH = 1 << 26
for a in range(0, n, CH):
    m = min(CH, n - a)
    for col, arr in ((d.hashes, rng.integers(0, 4 ** 31, size=m, dtype=np.uint64)), (d.nodes, rng.integers(0, 1 << 24, size=m, dtype=np.uint32)),
                     (d.ref_offsets, rng.integers(0, 3 * 10 ** 9, size=m, dtype=np.uint64)), (d.allele_frequencies, rng.random(m, dtype=np.float32))):
        _lib.check(lib.gki_memcpy_h2d(col.view(a, m).ptr, _lib.hptr(arr), arr.nbytes))
sync = lambda: _lib.check(lib.gki_device_synchronize())
res = {"n": n, "modulo": modulo, "group_bits": g, "lib": os.path.basename(_lib.LIB_PATH)}
part = DeviceFlatKmers.allocate(n)
rows = DeviceRows(n)
n_parts = int(sys.argv[5]) if len(sys.argv) > 5 else 1
sums = {}
for form in ("plain", "grouped", "grouped_rows"):
    tp, tb = [], []
    for r in range(reps + 1):
        sync(); t = time.perf_counter()
        if form == "grouped":
            _, start = partition_by_bucket_range(d, modulo, 1, out=part, group_bits=g)
            sync(); tp.append(time.perf_counter() - t); t = time.perf_counter()
            idx = PartitionedDeviceIndex.build_slice(part, start, modulo, 1, 0, g)
        elif form == "grouped_rows":
            _, start = partition_rows_by_bucket_range(d, modulo, 1, group_bits=g, out=rows)
            sync(); tp.append(time.perf_counter() - t); t = time.perf_counter()
            idx = PartitionedDeviceIndex.build_slice(rows, start, modulo, 1, 0, g)
        else:
            idx = DeviceIndex.build(d, modulo)
        sync(); tb.append(time.perf_counter() - t)
        if r < reps:
            idx.free()
    sums[form] = [a.checksum() for a in (idx.hashes_to_index, idx.n_kmers, idx.kmers, idx.nodes, idx.ref_offsets, idx.allele_frequencies, idx.frequencies)]
    idx.free()
    res[form + "_build_ms"] = [round(1e3 * x, 2) for x in tb[1:]]
    if tp:
        res[form + "_partition_ms"] = [round(1e3 * x, 2) for x in tp[1:]]
res["forms_agree"] = sums["plain"] == sums["grouped"] == sums["grouped_rows"]
print(json.dumps(res))
