import sys, time, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np
from graph_kmer_index_amd import _lib
from graph_kmer_index_amd.flat_kmers import DeviceFlatKmers, FlatKmers
from graph_kmer_index_amd.collision_free_kmer_index import DeviceIndex
lib=_lib.load()
n=310_000_000
rng=np.random.default_rng(1)
k=rng.integers(0,4**31,size=n,dtype=np.int64).view(np.uint64)
d=DeviceFlatKmers.from_flat_kmers(FlatKmers(k, np.zeros(n,np.uint32), np.arange(n,dtype=np.uint64), np.ones(n,np.float32)))
for rep in range(3):
    lib.gki_device_synchronize(); t=time.perf_counter()
    idx=DeviceIndex.build(d, 452930477); lib.gki_device_synchronize()
    print("build %.1f ms" % (1e3*(time.perf_counter()-t))); idx.free()
