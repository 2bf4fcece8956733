#!/usr/bin/env python3
"""Experiment: does the interior kernel's rate depend on how the four output columns are placed relative to each other?
  python tools/exp/column_skew.py <shard r/w> <skew bytes between column starts>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from graph_kmer_index_amd import _lib, DenseKmerFinder, CriticalGraphPaths, DeviceGraph
from graph_kmer_index_amd.flat_kmers import DeviceFlatKmers
from graph_kmer_index_amd.graph import synthetic_snp_graph
from graph_kmer_index_amd.sharding import shard_range

r, w = (int(x) for x in sys.argv[1].split("/"))
skews = [int(x) for x in sys.argv[2:]]
k = 31
g = synthetic_snp_graph(int(3e9), int(5e6), k=k, seed=1234)
cp = CriticalGraphPaths.from_graph(g, k)
g._device = DeviceGraph(g)
a, b = shard_range(g, cp, r, w)
f = DenseKmerFinder(g, k, critical_graph_paths=cp, only_save_one_node_per_kmer=True, max_variant_nodes=5,
                    start_at_critical_path_number=a if w > 1 else None, stop_at_critical_path_number=b if w > 1 else None)
n = f._count(layout=1)
for skew in skews:
    pad = 4 * skew + 4096
    cols = []
    for i, dt in enumerate((np.uint64, np.uint32, np.uint64, np.float32)):
        item = np.dtype(dt).itemsize
        raw = _lib.DeviceArray(n + pad // item, dt)
        cols.append((raw, raw.view((i * skew) // item, n)))
    out = DeviceFlatKmers(n, *[c[1] for c in cols])
    ts = []
    for _ in range(6):
        out = f.find_flat_on_device(out)
        f.synchronize()
        ts.append((f.kernel_ms(1), f.kernel_ms(2)))
    print("shard %d/%d records %d skew %7d B: interior %.2f ms  boundary %.2f ms   ptrs %s" % (
        r, w, n, skew, np.median([t[0] for t in ts[1:]]), np.median([t[1] for t in ts[1:]]),
        [hex(c[1].ptr.value % (1 << 22)) for c in cols]), flush=True)
    for raw, _ in cols:
        raw.free()
