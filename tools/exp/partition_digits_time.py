#!/usr/bin/env python3
"""Time gki_partition_rows_by_bucket_range / gki_partition_by_bucket_range on n random device-resident records for several
(n_parts, group_bits): how does the pass depend on the number of digits and on the footprint it scatters over?
usage: python tools/exp/partition_digits_time.py [n] [modulo]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from graph_kmer_index_amd import _lib
from graph_kmer_index_amd.flat_kmers import DeviceFlatKmers
from graph_kmer_index_amd.collision_free_kmer_index import DeviceRows, partition_by_bucket_range, partition_rows_by_bucket_range

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 395_000_000
modulo = int(sys.argv[2]) if len(sys.argv) > 2 else 452930477
lib = _lib.load(); _lib.require_device()
rng = np.random.default_rng(7)
d = DeviceFlatKmers.allocate(n)
CH = 1 << 26
base = {}
for a in range(0, n, CH):
    m = min(CH, n - a)
    if m not in base:                        # (one chunk of random records, repeated: the keys stay uniformly spread)
        base[m] = (rng.integers(0, 4 ** 31, size=m, dtype=np.uint64), rng.integers(0, 1 << 24, size=m, dtype=np.uint32),
                   rng.integers(0, 3 * 10 ** 9, size=m, dtype=np.uint64), rng.random(m, dtype=np.float32))
    k, nd, rf, af = base[m]
    k = k ^ np.uint64(a * 2654435761 % (1 << 40))
    for col, arr in ((d.hashes, k), (d.nodes, nd), (d.ref_offsets, rf), (d.allele_frequencies, af)):
        _lib.check(lib.gki_memcpy_h2d(col.view(a, m).ptr, _lib.hptr(np.ascontiguousarray(arr)), arr.nbytes))
sync = lambda: _lib.check(lib.gki_device_synchronize())
rows, cols = DeviceRows(n), DeviceFlatKmers.allocate(n)
res = {"n": n, "modulo": modulo}
for n_parts, g in ((8, 0), (8, 2), (8, 4), (8, 5), (8, 7), (1, 7), (1, 10)):
    for form in ("rows", "cols"):
        ts = []
        for r in range(3):
            sync(); t = time.perf_counter()
            if form == "rows":
                partition_rows_by_bucket_range(d, modulo, n_parts, group_bits=g, out=rows)
            else:
                partition_by_bucket_range(d, modulo, n_parts, out=cols, group_bits=g)
            sync(); ts.append(1e3 * (time.perf_counter() - t))
        res["%d_parts_%d_bits_%s_ms" % (n_parts, g, form)] = round(min(ts[1:]), 2)
print(json.dumps(res))
