import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from graph_kmer_index_amd import _lib
from graph_kmer_index_amd.flat_kmers import DeviceFlatKmers
from graph_kmer_index_amd.collision_free_kmer_index import DeviceIndex, partition_by_bucket_range
lib = _lib.load(); _lib.require_device()
n = int(float(sys.argv[1]))
rng = np.random.default_rng(0)
d = DeviceFlatKmers(n, _lib.DeviceArray.from_host(rng.integers(0, 4**31, size=n, dtype=np.uint64)),
                    _lib.DeviceArray.from_host(rng.integers(0, 10**7, size=n).astype(np.uint32)),
                    _lib.DeviceArray.from_host(np.arange(n, dtype=np.uint64)), _lib.DeviceArray.from_host(np.ones(n, np.float32)))
def sync(): _lib.check(lib.gki_device_synchronize())
for rep in range(3):
    t = time.perf_counter(); out, st = partition_by_bucket_range(d, 452930477, 8); sync(); print("partition", rep, time.perf_counter() - t, flush=True); out.free()
for rep in range(3):
    t = time.perf_counter(); ix = DeviceIndex.build(d, 452930477); sync(); print("build", rep, time.perf_counter() - t, flush=True); ix.free()
for rep in range(3):
    t = time.perf_counter(); a = _lib.DeviceArray(n * 4, np.uint64); print("malloc %.1f GB" % (n * 32 / 1e9), rep, time.perf_counter() - t, flush=True)
    t = time.perf_counter(); a.free(); print("free", rep, time.perf_counter() - t, flush=True)
