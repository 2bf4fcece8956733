#!/usr/bin/env python3
"""Experiment: how much of the index build is the random row gather?  Builds the same 3.1e8 random records twice: in
random order, and pre-partitioned into 256 bucket ranges (the gather of the build then reads rows from a ~40 MB window
at a time, which the Infinity Cache holds)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from graph_kmer_index_amd import _lib
from graph_kmer_index_amd.flat_kmers import DeviceFlatKmers, FlatKmers
from graph_kmer_index_amd.collision_free_kmer_index import DeviceIndex, partition_by_bucket_range
lib = _lib.load()
n, modulo = 310_000_000, 452930477
rng = np.random.default_rng(1)
k = rng.integers(0, 4 ** 31, size=n, dtype=np.int64).view(np.uint64)
d = DeviceFlatKmers.from_flat_kmers(FlatKmers(k, np.zeros(n, np.uint32), np.arange(n, dtype=np.uint64), np.ones(n, np.float32)))
t = time.perf_counter(); part, start = partition_by_bucket_range(d, modulo, 256); lib.gki_device_synchronize()
print("partition into 256 ranges: %.1f ms" % (1e3 * (time.perf_counter() - t)))
for name, src in (("random order", d), ("pre-partitioned", part)):
    for rep in range(3):
        lib.gki_device_synchronize(); t = time.perf_counter()
        idx = DeviceIndex.build(src, modulo); lib.gki_device_synchronize()
        dt = time.perf_counter() - t
        idx.free()
    print("%s: build %.1f ms" % (name, 1e3 * dt))
