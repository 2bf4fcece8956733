#!/usr/bin/env python3
"""Secondary benchmark (not the headline): BASELINE configs[3] rehearsed on ONE GPU.
  python tools/bench_partitioned.py --bases 3e9 --sites 5e6 --world 8
The bucket-range partitioned build with the `world` ranks run one after the other on the single GPU:
every rank enumerates its critical-path shard and partitions its records by owning rank; the all-to-all is stood in for
by device copies (rank r's slice p appended to part p's receive buffer, in rank order: exactly what
gki_comm_alltoall_flat delivers); every part then builds its directory slice.  The result is the index of ALL
3.16e9 records of the 3 Gbp graph in `world` slices -- more than one int32 directory can address -- resident on one
MI355X.  Checks: record counts add up, the per-column checksums of the index payload equal those of the FlatKmers
columns (same multiset), every slice accepted its records (gki_index_build_range rejects a foreign bucket).
Prints one JSON object."""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from graph_kmer_index_amd import _lib, CriticalGraphPaths
from graph_kmer_index_amd.flat_kmers import DeviceFlatKmers
from graph_kmer_index_amd.collision_free_kmer_index import DeviceIndex, PartitionedDeviceIndex, bucket_range, \
    partition_by_bucket_range
from graph_kmer_index_amd.graph import synthetic_snp_graph, synthetic_haplotype_sequence
from graph_kmer_index_amd.parallel import find_sharded

MASK = (1 << 64) - 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bases", type=float, default=3e9)
    ap.add_argument("--sites", type=float, default=5e6)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--modulo", type=int, default=452930477)
    ap.add_argument("--reads", type=float, default=2e6)
    args = ap.parse_args()
    lib = _lib.load(); _lib.require_device()
    k, W, M = 31, args.world, args.modulo
    g = synthetic_snp_graph(int(args.bases), int(args.sites), k=k, seed=1234)
    cp = CriticalGraphPaths.from_graph(g, k)
    kw = dict(only_save_one_node_per_kmer=True, max_variant_nodes=5)
    t_find = t_part = t_copy = t_build = 0.0
    shards, starts = [], []
    sums, xors = [0] * 4, [0] * 4
    for r in range(W):
        t = time.perf_counter()
        mine = find_sharded(g, k, cp, r, W, **kw)
        t_find += time.perf_counter() - t
        for i, c in enumerate((mine.hashes, mine.nodes, mine.ref_offsets, mine.allele_frequencies)):
            s, x = c.checksum(mine.n)
            sums[i] = (sums[i] + s) & MASK; xors[i] ^= x
        t = time.perf_counter()
        by_dest, send_start = partition_by_bucket_range(mine, M, W)
        _lib.check(lib.gki_device_synchronize())
        t_part += time.perf_counter() - t
        mine.free()
        shards.append(by_dest); starts.append(send_start)
    total = sum(s.n for s in shards)
    parts = []
    isums, ixors = [0] * 4, [0] * 4
    for p in range(W):
        t = time.perf_counter()
        n_p = sum(starts[r][p + 1] - starts[r][p] for r in range(W))
        recv = DeviceFlatKmers.allocate(n_p)
        at = 0
        for r in range(W):                      # stands in for gki_comm_alltoall_flat: slices land in rank order
            a, b = starts[r][p], starts[r][p + 1]
            for src, dst in ((shards[r].hashes, recv.hashes), (shards[r].nodes, recv.nodes),
                             (shards[r].ref_offsets, recv.ref_offsets), (shards[r].allele_frequencies, recv.allele_frequencies)):
                if b > a:
                    _lib.check(lib.gki_memcpy_d2d(dst.view(at, b - a).ptr, src.view(a, b - a).ptr, (b - a) * src.dtype.itemsize))
            at += b - a
        _lib.check(lib.gki_device_synchronize())
        t_copy += time.perf_counter() - t
        t = time.perf_counter()
        lo, hi = bucket_range(M, W, p)
        idx = DeviceIndex.build(recv, M, bucket_begin=lo, n_buckets=hi - lo)
        _lib.check(lib.gki_device_synchronize())
        t_build += time.perf_counter() - t
        recv.free()
        for i, c in enumerate((idx.kmers, idx.nodes, idx.ref_offsets, idx.allele_frequencies)):
            s, x = c.checksum(idx.n)
            isums[i] = (isums[i] + s) & MASK; ixors[i] ^= x
        parts.append(idx)
    for s in shards:
        s.free()
    index = PartitionedDeviceIndex(M, parts)
    free_b, total_b = C.c_int64(0), C.c_int64(0)
    _lib.check(lib.gki_mem_info(C.byref(free_b), C.byref(total_b)))
    res = {"config": "BASELINE configs[3] rehearsed on 1 GPU: %d ranks in turn" % W, "records": total,
           "records_per_slice": [p.n for p in parts], "exceeds_int32_directory": total >= 2 ** 31,
           "find_s": t_find, "partition_s": t_part, "exchange_standin_copy_s": t_copy, "build_slices_s": t_build,
           "build_records_per_s": total / (t_part + t_build),
           "payload_equals_flat_multiset": [(a, b) for a, b in zip(sums, xors)] == [(a, b) for a, b in zip(isums, ixors)],
           "hbm_used_gb": (total_b.value - free_b.value) / 1e9}
    # reads against the sliced index of the whole graph: every k-mer of an error-free read must hit
    n_reads = int(args.reads)
    path = synthetic_haplotype_sequence(g)
    rng = np.random.default_rng(99)
    starts_r = rng.integers(0, len(path) - 150, size=n_reads)
    letters = np.frombuffer(b"ACGT", np.uint8)[np.lib.stride_tricks.sliding_window_view(path, 150)[starts_r]].reshape(-1)
    read_start = np.arange(n_reads + 1, dtype=np.int64) * 150
    d_l, d_s = _lib.DeviceArray.from_host(letters), _lib.DeviceArray.from_host(read_start)
    for p in parts:
        p.probe_table()
    t = time.perf_counter()
    counts, n_kmers, hits = index.count_nodes_from_reads(d_l, d_s, k, g.n_nodes, strands=1, max_hits=2 ** 62)
    dt = time.perf_counter() - t
    res.update({"reads": n_reads, "forward_kmers": n_kmers, "hits": hits, "every_forward_kmer_hits": hits >= n_kmers,
                "map_reads_s": dt, "map_kmers_per_s": n_kmers / dt})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
