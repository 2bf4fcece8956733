#!/usr/bin/env python3
"""Secondary benchmark (not the headline): BASELINE configs[4], the read-side loop on one MI355X.
  python tools/bench_reads.py --bases 3e9 --sites 5e6 --reads 2e7
Index = the FlatKmers records of the synthetic SNP graph whose window crosses a node boundary (the KAGE-like variant
index of SURVEY.md 8d C5: the boundary section of the finder's split layout), default modulo.  Reads: 150 bp, 90 %
sampled from a random path of the graph (either strand) with 1 % substitutions, 10 % uniform random, seed 99.  Three ways to get node counts
of all read k-mers on both strands, checked equal:
  two_pass_reference_layout  gki_hash_reads -> gki_index_count_nodes   (k-mers materialised, five-array probe)
  two_pass_probe_table       gki_hash_reads -> gki_probe_count_nodes   (k-mers materialised, probe table)
  fused                      gki_probe_reads_count_nodes               (letters -> counts)
Prints one JSON object."""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from graph_kmer_index_amd import _lib, DenseKmerFinder, CriticalGraphPaths
from graph_kmer_index_amd.flat_kmers import DeviceFlatKmers
from graph_kmer_index_amd.collision_free_kmer_index import DeviceIndex
from graph_kmer_index_amd.graph import synthetic_snp_graph, synthetic_haplotype_sequence


def make_reads(seq, n_reads, rng, chunk=2_000_000):
    """ASCII letters uint8[n_reads * 150]"""
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = np.empty((n_reads, 150), dtype=np.uint8)
    win = np.lib.stride_tricks.sliding_window_view(seq, 150)
    for a in range(0, n_reads, chunk):
        b = min(n_reads, a + chunk)
        codes = win[rng.integers(0, len(win), size=b - a)]
        sub = rng.random(codes.shape) < 0.01
        codes = np.where(sub, (codes + rng.integers(1, 4, size=codes.shape, dtype=np.uint8)) & 3, codes)
        rnd = rng.random(b - a) < 0.1
        codes[rnd] = rng.integers(0, 4, size=(int(rnd.sum()), 150), dtype=np.uint8)
        rc = rng.random(b - a) < 0.5                     # half the reads come from the reverse strand
        codes[rc] = 3 - codes[rc][:, ::-1]
        np.take(lut, codes, out=out[a:b])
    return out.reshape(-1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bases", type=float, default=3e9)
    ap.add_argument("--sites", type=float, default=5e6)
    ap.add_argument("--reads", type=float, default=2e7)
    ap.add_argument("--batch", type=float, default=1e7, help="reads per launch")
    ap.add_argument("--modulo", type=int, default=452930477)
    ap.add_argument("--max-hits", type=int, default=10)
    ap.add_argument("--skip-two-pass", action="store_true")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--whitelist-find", action="store_true",
                    help="also time DenseKmerFinder.find_flat_on_device with the variant index as whitelist")
    args = ap.parse_args()
    lib = _lib.load(); _lib.require_device()
    k = 31
    t0 = time.perf_counter()
    g = synthetic_snp_graph(int(args.bases), int(args.sites), k=k, seed=1234)
    cp = CriticalGraphPaths.from_graph(g, k)
    f = DenseKmerFinder(g, k, critical_graph_paths=cp, only_save_one_node_per_kmer=True, max_variant_nodes=5)
    flat = f.find_flat_on_device(); f.synchronize()
    n_int = f.interior_records()
    nb = flat.n - n_int
    bnd = DeviceFlatKmers(nb, flat.hashes.view(n_int, nb), flat.nodes.view(n_int, nb), flat.ref_offsets.view(n_int, nb),
                          flat.allele_frequencies.view(n_int, nb))
    res = {"config": "BASELINE configs[4] on 1 GPU", "graph_bases": int(args.bases), "snp_sites": int(args.sites),
           "flat_records": flat.n, "index_records": nb, "modulo": args.modulo, "max_hits": args.max_hits,
           "setup_s": time.perf_counter() - t0}
    t = time.perf_counter()
    idx = DeviceIndex.build(bnd, args.modulo)
    res["index_build_s"] = time.perf_counter() - t
    flat.free(); f = None
    if args.whitelist_find:
        # the CLI `index --whitelist` flow (command_line_interface.py:559-565, 634) in HBM: emit, membership probe of
        # every record against the whitelist index, stable compaction
        fw = DenseKmerFinder(g, k, critical_graph_paths=cp, only_save_one_node_per_kmer=True, max_variant_nodes=5, whitelist=idx)
        for rep in range(2):
            t = time.perf_counter()
            kept = fw.find_flat_on_device()
            _lib.check(lib.gki_device_synchronize())
            res["whitelisted_find_s"] = time.perf_counter() - t
            res["whitelisted_records"] = kept.n
            kept.free()
        fw.close(); fw = None
    t = time.perf_counter()
    idx.probe_table()
    res["probe_table_build_s"] = time.perf_counter() - t
    n_nodes = len(g.node_size)
    n_reads, batch = int(args.reads), int(args.batch)
    t = time.perf_counter()
    letters = make_reads(synthetic_haplotype_sequence(g), n_reads, np.random.default_rng(99))
    res["make_reads_host_s"] = time.perf_counter() - t
    res["reads"] = n_reads
    per_read = 150 - k + 1
    view = idx.view()
    mh = args.max_hits

    def batches():
        for a in range(0, n_reads, batch):
            b = min(n_reads, a + batch)
            yield (_lib.DeviceArray.from_host(letters[a * 150:b * 150]),
                   _lib.DeviceArray.from_host(np.arange(b - a + 1, dtype=np.int64) * 150), b - a)

    dev_batches = list(batches())          # letters resident in HBM before any timed region
    results = {}

    def run_two_pass(use_table):
        counts = _lib.DeviceArray(n_nodes, np.uint32); counts.zero()
        t_hash = t_probe = 0.0
        hits = 0
        for d_letters, d_start, nr in dev_batches:
            nq = nr * per_read
            d_q = _lib.DeviceArray(nq, np.uint64); d_os = _lib.DeviceArray(nr + 1, np.int64)
            n_out = C.c_int64(0)
            for strand in (0, 1):
                t = time.perf_counter()
                _lib.check(lib.gki_hash_reads(d_letters.ptr, d_start.ptr, nr, k, strand, d_os.ptr, d_q.ptr, nq, C.byref(n_out)))
                t_hash += time.perf_counter() - t
                t = time.perf_counter()
                if use_table:
                    h = C.c_int64(0)
                    _lib.check(lib.gki_probe_count_nodes(idx.probe_table(), d_q.ptr, nq, mh, counts.ptr, n_nodes, C.byref(h)))
                    hits += h.value
                else:
                    _lib.check(lib.gki_index_count_nodes(C.byref(view), d_q.ptr, nq, mh, counts.ptr, n_nodes))
                t_probe += time.perf_counter() - t
            d_q.free(); d_os.free()
        return counts, t_hash, t_probe, hits

    def run_fused():
        counts = _lib.DeviceArray(n_nodes, np.uint32); counts.zero()
        tt = 0.0
        kmers = hits = 0
        for d_letters, d_start, nr in dev_batches:
            nk_, nh_ = C.c_int64(0), C.c_int64(0)
            t = time.perf_counter()
            _lib.check(lib.gki_probe_reads_count_nodes(idx.probe_table(), d_letters.ptr, d_start.ptr, nr, k, 3, mh,
                                                       counts.ptr, n_nodes, C.byref(nk_), C.byref(nh_)))
            tt += time.perf_counter() - t
            kmers += nk_.value; hits += nh_.value
        return counts, tt, kmers, hits

    total_kmers = 2 * n_reads * per_read
    ref_counts = None
    for rep in range(args.reps):
        c, tt, kmers, hits = run_fused()
        results["fused"] = {"s": tt, "kmers_per_s": kmers / tt, "reads_per_s": n_reads / tt, "kmers": kmers, "hits": hits}
        if ref_counts is None:
            ref_counts = c.to_host()
        c.free()
    if not args.skip_two_pass:
        for name, use_table in (("two_pass_probe_table", True), ("two_pass_reference_layout", False)):
            for rep in range(args.reps):
                c, th, tp, hits = run_two_pass(use_table)
                results[name] = {"hash_s": th, "probe_s": tp, "kmers_per_s": total_kmers / (th + tp),
                                 "probe_queries_per_s": total_kmers / tp, "hash_kmers_per_s": total_kmers / th}
                if use_table:
                    results[name]["hits"] = hits
                same = bool(np.array_equal(c.to_host(), ref_counts))
                results[name]["counts_equal_fused"] = same
                c.free()
    res["hit_fraction"] = results["fused"]["hits"] / max(results["fused"]["kmers"], 1)
    res["nodes_with_hits"] = int((ref_counts > 0).sum())
    res.update(results)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
