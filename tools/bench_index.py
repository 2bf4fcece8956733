#!/usr/bin/env python3
"""Secondary benchmark (not the headline): CollisionFreeKmerIndex build and batched lookup on MI355X.
  python tools/bench_index.py --bases 3e8 --sites 5e5 --reads 2e6
Builds the index from the FlatKmers of a synthetic SNP graph (columns stay in HBM), then hashes synthetic 150-bp reads
(both strands) and probes the index.  Prints one JSON object."""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from graph_kmer_index_amd import _lib, DenseKmerFinder, CriticalGraphPaths, DeviceGraph
from graph_kmer_index_amd.collision_free_kmer_index import DeviceIndex
from graph_kmer_index_amd.graph import synthetic_snp_graph, synthetic_haplotype_sequence


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bases", type=float, default=3e8)
    ap.add_argument("--sites", type=float, default=5e5)
    ap.add_argument("--reads", type=float, default=2e6)
    ap.add_argument("--modulo", type=int, default=452930477)
    ap.add_argument("--skip-frequencies", action="store_true")
    args = ap.parse_args()
    lib = _lib.load(); _lib.require_device()
    k = 31
    g = synthetic_snp_graph(int(args.bases), int(args.sites), k=k, seed=1234)
    cp = CriticalGraphPaths.from_graph(g, k)
    f = DenseKmerFinder(g, k, critical_graph_paths=cp, only_save_one_node_per_kmer=True, max_variant_nodes=5)
    flat = f.find_flat_on_device(); f.synchronize()
    n = flat.n
    res = {"records": n, "modulo": args.modulo}
    for rep in range(2):
        t = time.perf_counter()
        idx = DeviceIndex.build(flat, args.modulo, args.skip_frequencies)
        _lib.check(lib.gki_device_synchronize())
        dt = time.perf_counter() - t
        if rep == 0:
            idx.free()
    res["index_build_s"] = dt
    res["index_build_records_per_s"] = n / dt
    # reads: 90 % sampled from a random path of the graph, 10 % random, forward strand letters
    rng = np.random.default_rng(99)
    n_reads = int(args.reads)
    path = synthetic_haplotype_sequence(g)
    starts = rng.integers(0, len(path) - 150, size=n_reads)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)[path[(starts[:, None] + np.arange(150)[None, :]).ravel()]].copy()
    rnd = rng.random(n_reads) < 0.1
    letters.reshape(n_reads, 150)[rnd] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(int(rnd.sum()), 150))]
    read_start = (np.arange(n_reads + 1, dtype=np.int64) * 150)
    d_letters = _lib.DeviceArray.from_host(letters); d_start = _lib.DeviceArray.from_host(read_start)
    d_out_start = _lib.DeviceArray(n_reads + 1, np.int64)
    nq = n_reads * (150 - k + 1)
    d_q = _lib.DeviceArray(nq, np.uint64)
    n_out = C.c_int64(0)
    tot_hash = tot_probe = 0.0
    hits = 0
    view = idx.view()
    d_hs = _lib.DeviceArray(nq + 1, np.int64)
    for strand in (0, 1):
        for rep in range(2):
            t = time.perf_counter()
            _lib.check(lib.gki_hash_reads(d_letters.ptr, d_start.ptr, n_reads, k, strand, d_out_start.ptr, d_q.ptr, nq, C.byref(n_out)))
            dt_h = time.perf_counter() - t
            nh = C.c_int64(0)
            t = time.perf_counter()
            _lib.check(lib.gki_index_lookup_count(C.byref(view), d_q.ptr, nq, 10, d_hs.ptr, C.byref(nh)))
            m = nh.value
            bufs = [_lib.DeviceArray(max(m, 1), dt_) for dt_ in (np.uint32, np.uint64, np.int64, np.uint16, np.float32)]
            _lib.check(lib.gki_index_lookup_emit(C.byref(view), d_q.ptr, nq, 10, d_hs.ptr, *[b.ptr for b in bufs], None))
            dt_p = time.perf_counter() - t
            for b in bufs:
                b.free()
        tot_hash += dt_h; tot_probe += dt_p; hits += m
    res.update({"reads": n_reads, "kmers_per_strand": nq, "hash_reads_kmers_per_s": 2 * nq / tot_hash,
                "lookup_queries_per_s": 2 * nq / tot_probe, "hits": hits})
    # the same batched get on the probe table: per hit the query index and the payload position
    table = idx.probe_table()
    tot = 0.0
    hits_t = 0
    for strand in (0, 1):
        _lib.check(lib.gki_hash_reads(d_letters.ptr, d_start.ptr, n_reads, k, strand, d_out_start.ptr, d_q.ptr, nq, C.byref(n_out)))
        for rep in range(2):
            nh = C.c_int64(0)
            t = time.perf_counter()
            _lib.check(lib.gki_probe_lookup_count(table, d_q.ptr, nq, 10, d_hs.ptr, C.byref(nh)))
            m = nh.value
            qi, pos = _lib.DeviceArray(max(m, 1), np.int64), _lib.DeviceArray(max(m, 1), np.int64)
            _lib.check(lib.gki_probe_lookup_emit(table, d_q.ptr, nq, 10, d_hs.ptr, qi.ptr, pos.ptr))
            dt = time.perf_counter() - t
            qi.free(); pos.free()
        tot += dt; hits_t += m
    res.update({"probe_table_lookup_queries_per_s": 2 * nq / tot, "probe_table_hits": hits_t})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
