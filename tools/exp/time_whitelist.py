import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from graph_kmer_index_amd import _lib, DenseKmerFinder, CriticalGraphPaths
from graph_kmer_index_amd.flat_kmers import DeviceFlatKmers
from graph_kmer_index_amd.collision_free_kmer_index import DeviceIndex
from graph_kmer_index_amd.graph import synthetic_snp_graph
lib = _lib.load(); _lib.require_device()
def sync(): _lib.check(lib.gki_device_synchronize())
k = 31
g = synthetic_snp_graph(int(3e9), int(5e6), k=k, seed=1234)
cp = CriticalGraphPaths.from_graph(g, k)
f = DenseKmerFinder(g, k, critical_graph_paths=cp, only_save_one_node_per_kmer=True, max_variant_nodes=5)
for rep in range(3):
    t = time.perf_counter(); flat = f.find_flat_on_device(); sync(); print("find+alloc", rep, time.perf_counter() - t, flush=True)
    if rep < 2: flat.free()
n_int = f.interior_records(); nb = flat.n - n_int
bnd = DeviceFlatKmers(nb, flat.hashes.view(n_int, nb), flat.nodes.view(n_int, nb), flat.ref_offsets.view(n_int, nb), flat.allele_frequencies.view(n_int, nb))
idx = DeviceIndex.build(bnd, 452930477); idx.probe_table(); sync()
for rep in range(3):
    t = time.perf_counter(); flags = idx.contains(flat.hashes.view(0, flat.n)); sync(); t1 = time.perf_counter()
    c = flags.checksum(flat.n); t2 = time.perf_counter()
    kept = flat.compacted(flags); sync(); t3 = time.perf_counter()
    print("contains %.4f  checksum %.4f  compact %.4f  kept %d" % (t1 - t, t2 - t1, t3 - t2, kept.n), flush=True)
    kept.free(); flags.free()
t = time.perf_counter(); flat.free(); print("free 76 GB", time.perf_counter() - t)
