#!/usr/bin/env python3
"""Time the two forms of the index build on n random records (device resident), default modulo, and check that their
outputs agree (device checksums of every output array).  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel
split.  usage: python tools/exp/index_forms_time.py [n] [modulo] [reps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from graph_kmer_index_amd import _lib
from graph_kmer_index_amd.flat_kmers import DeviceFlatKmers
from graph_kmer_index_amd.collision_free_kmer_index import DeviceIndex

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 310_000_000
modulo = int(sys.argv[2]) if len(sys.argv) > 2 else 452930477
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
forms = sys.argv[4].split(",") if len(sys.argv) > 4 else ["rows", "pairs"]
lib = _lib.load(); _lib.require_device()
rng = np.random.default_rng(7)
t = time.perf_counter()
d = DeviceFlatKmers.allocate(n)
CH = 1 << 26
for a in range(0, n, CH):
    m = min(CH, n - a)
    for col, arr in ((d.hashes, rng.integers(0, 4 ** 31, size=m, dtype=np.uint64)), (d.nodes, rng.integers(0, 1 << 24, size=m, dtype=np.uint32)),
                     (d.ref_offsets, rng.integers(0, 3 * 10 ** 9, size=m, dtype=np.uint64)), (d.allele_frequencies, rng.random(m, dtype=np.float32))):
        _lib.check(lib.gki_memcpy_h2d(col.view(a, m).ptr, _lib.hptr(arr), arr.nbytes))
print("generated %d records in %.1f s" % (n, time.perf_counter() - t), file=sys.stderr, flush=True)
res = {"n": n, "modulo": modulo, "lib": os.path.basename(_lib.LIB_PATH)}
sums = {}
for form in forms:
    times = []
    for r in range(reps + 1):
        _lib.check(lib.gki_device_synchronize())
        t = time.perf_counter()
        idx = DeviceIndex.build(d, modulo, pairs_form=(form == "pairs"))
        _lib.check(lib.gki_device_synchronize())
        times.append(time.perf_counter() - t)
        if r < reps:
            idx.free()
    sums[form] = [a.checksum() for a in (idx.hashes_to_index, idx.n_kmers, idx.kmers, idx.nodes, idx.ref_offsets,
                                        idx.allele_frequencies, idx.frequencies)]
    idx.free()
    res[form + "_ms"] = [round(1e3 * x, 2) for x in times[1:]]
if len(forms) == 2:
    res["forms_agree"] = sums[forms[0]] == sums[forms[1]]
print(json.dumps(res))
