#!/usr/bin/env python3
"""Randomised soak test (not part of the suite): GPU DenseKmerFinder.find() against the oracle on many random graphs,
k in [2, 31], every max_variant_nodes regime, both node modes, random critical-path chunks, optional whitelist.
  python tools/soak_parity.py --seconds 120 --seed 1
Exits non-zero at the first mismatch (prints the seed and parameters to reproduce)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from graph_kmer_index_amd import DenseKmerFinder, CriticalGraphPaths, GraphArrays
from graph_kmer_index_amd.graph import synthetic_indel_graph
from graphgen import random_bubble_graph, overlapping_bubble_graph, nested_bubble_graph, deep_nested_graph
from gpu_util import assert_same_records, finder_cols
from oracle import oracle


_last = [time.time()]


def progress(msg):
    """one line a minute: a silent GPU job is taken to be hung"""
    if time.time() - _last[0] > 45:
        _last[0] = time.time()
        print(msg, flush=True)


def make_graph(rng, k):
    mode = rng.choice(["bubble", "overlap", "chain", "indel", "nested", "deep"])
    if mode == "nested":               # a variant inside an alternative allele: nodes with no linear-ref predecessor
        seqs, edges, lin, af = nested_bubble_graph(rng, n_var=int(rng.integers(2, 12)), min_ref=1, max_ref=int(rng.integers(2, 2 * k + 3)),
                                                   p_nest=float(rng.choice([0.3, 0.7])), p_chain=float(rng.choice([0.0, 0.4])))
        return mode, GraphArrays.from_dicts(seqs, edges, lin, af)
    if mode == "deep":
        seqs, edges, lin, af = deep_nested_graph(rng, n_var=int(rng.integers(1, 8)), max_depth=int(rng.integers(1, 4)), min_ref=1,
                                                 max_ref=int(rng.integers(2, 2 * k + 3)), max_allele=int(rng.integers(1, 8)))
        return mode, GraphArrays.from_dicts(seqs, edges, lin, af)
    if mode == "bubble":
        seqs, edges, lin, af = random_bubble_graph(rng, n_var=int(rng.integers(2, 40)), min_ref=1, max_ref=int(rng.integers(2, 3 * k + 3)),
                                                   p_indel=float(rng.choice([0.0, 0.3, 0.7])), with_af=True)
        return mode, GraphArrays.from_dicts(seqs, edges, lin, af)
    if mode == "overlap":
        seqs, edges, lin, af = overlapping_bubble_graph(rng, n_var=int(rng.integers(3, 14)), min_ref=2, max_ref=max(3, k))
        return mode, GraphArrays.from_dicts(seqs, edges, lin, af)
    if mode == "chain":
        nv = int(rng.integers(1, 8))
        seqs, edges, lin, af = random_bubble_graph(rng, n_var=nv, min_ref=1, max_ref=3 * k + 8, p_indel=0.3,
                                                   chain_after={int(rng.integers(-1, nv)): int(rng.integers(1, k + 2))})
        return mode, GraphArrays.from_dicts(seqs, edges, lin, af)
    G = int(rng.integers(40 * k, 400 * k))
    return mode, synthetic_indel_graph(G, max(2, G // int(rng.integers(8, 60))), k=k, seed=int(rng.integers(0, 1 << 30)),
                                       p_del=float(rng.uniform(0, 0.4)), p_ins=float(rng.uniform(0, 0.4)))


def soak_forward(args):
    """find_only_kmers_starting_at_position (batched), reference order, with only_follow / only_store node sets."""
    t_end = time.time() + args.seconds
    it = checked = 0
    while time.time() < t_end:
        seed = args.seed * 1_000_003 + it
        it += 1
        progress("... %d drawn, %d compared" % (it, checked))
        rng = np.random.default_rng(seed)
        k = int(rng.integers(2, 32))
        M = int(rng.choice([0, 1, 2, 3, 4, 5, 100]))
        one = bool(rng.integers(0, 2))
        try:
            mode, g = make_graph(rng, k)
        except AssertionError:
            continue
        kw = {}
        if rng.random() < 0.5:
            variant = np.nonzero(g.is_ref == 0)[0]
            if len(variant):
                chosen = set(int(x) for x in rng.choice(variant, size=min(len(variant), int(rng.integers(1, 4))), replace=False))
                kw = dict(only_store_nodes=chosen, only_follow_nodes=chosen) if rng.random() < 0.5 else dict(only_follow_nodes=chosen)
        nodes = rng.integers(0, g.n_nodes, size=16)
        offs = [int(rng.integers(0, max(1, g.node_size[n]))) for n in nodes]
        try:
            exp = [oracle.find_from_position(g, k, int(n), int(o), one, M, **kw) for n, o in zip(nodes, offs)]
        except oracle.OracleError:
            continue
        exp = {key: np.concatenate([e[key] for e in exp]) for key in exp[0]}
        f = DenseKmerFinder(g, k, only_save_one_node_per_kmer=one, max_variant_nodes=M, **kw)
        try:
            f.find_kmers_starting_at_positions(nodes, offs)
        except NotImplementedError:
            continue
        try:
            assert_same_records(finder_cols(f), exp, exact_order=True)
        except AssertionError as e:
            print("MISMATCH forward: seed %d %s k=%d M=%d one=%s %s" % (seed, mode, k, M, one, kw), e); sys.exit(1)
        f.close()
        checked += 1
    print("soak forward ok: %d graphs x 16 start positions" % checked)


def soak_index(args):
    """from_flat_kmers element-wise against the oracle's stable build; get / batched getters against loops of get."""
    from graph_kmer_index_amd import CollisionFreeKmerIndex, FlatKmers
    t_end = time.time() + args.seconds
    it = 0
    while time.time() < t_end:
        seed = args.seed * 1_000_003 + it
        it += 1
        progress("... %d indexes" % it)
        rng = np.random.default_rng(seed)
        n = int(rng.integers(1, 30000))
        modulo = int(rng.choice([1, 2, 3, 97, 1009, 65537, 452930477]))
        pool = rng.integers(0, 4 ** int(rng.integers(1, 32)), size=int(rng.integers(1, n + 1)), dtype=np.int64)
        kmers = pool[rng.integers(0, len(pool), size=n)]
        nodes = rng.integers(0, 1000, size=n).astype(np.uint32)
        refs = rng.integers(0, int(rng.integers(1, 50)), size=n).astype(np.uint64)
        af = rng.random(n).astype(np.float32)
        skip_freq, skip_single = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        desc = "seed %d n=%d modulo=%d skip_freq=%s skip_singletons=%s" % (seed, n, modulo, skip_freq, skip_single)
        idx = CollisionFreeKmerIndex.from_flat_kmers(FlatKmers(kmers, nodes, refs, af), modulo=modulo, skip_frequencies=skip_freq,
                                                     skip_singletons=skip_single)
        orc = oracle.index_build(kmers, nodes, refs, af, modulo=modulo, skip_frequencies=skip_freq, skip_singletons=skip_single)
        for name in ("_hashes_to_index", "_n_kmers", "_kmers", "_nodes", "_ref_offsets", "_allele_frequencies"):
            if not np.array_equal(np.asarray(getattr(idx, name)).astype(np.int64) if name != "_allele_frequencies" else getattr(idx, name),
                                  np.asarray(orc[name]).astype(np.int64) if name != "_allele_frequencies" else orc[name]):
                print("MISMATCH index build", name, desc); sys.exit(1)
        if not skip_freq and not np.array_equal(idx._frequencies, orc["_frequencies"]):
            print("MISMATCH frequencies", desc); sys.exit(1)
        queries = np.concatenate([pool[rng.integers(0, len(pool), size=200)], rng.integers(0, 4 ** 31, size=50, dtype=np.int64)])
        max_hits = int(rng.choice([1, 2, 10, 2 ** 62]))
        got_nodes, got_refs, got_q, got_af = idx.get_nodes_and_ref_offsets_from_multiple_kmers(queries.astype(np.uint64), max_hits=max_hits) \
            if hasattr(idx, "get_nodes_and_ref_offsets_from_multiple_kmers") else (None,) * 4
        e_nodes, e_refs = [], []
        for q in queries:
            r = oracle.index_get(orc, int(q), max_hits)
            if r[0] is not None:
                e_nodes.append(np.asarray(r[0])); e_refs.append(np.asarray(r[1]))
        e_nodes = np.concatenate(e_nodes) if e_nodes else np.zeros(0, np.uint32)
        e_refs = np.concatenate(e_refs) if e_refs else np.zeros(0, np.uint64)
        if got_nodes is not None and not (np.array_equal(np.asarray(got_nodes).astype(np.int64), e_nodes.astype(np.int64))
                                          and np.array_equal(np.asarray(got_refs).astype(np.int64), e_refs.astype(np.int64))):
            print("MISMATCH batched get", desc, "max_hits", max_hits); sys.exit(1)
    print("soak index ok: %d indexes" % it)


def soak_reads(args):
    """map_reads (fused read hashing + probe + node histogram), partitioned index slices and ReverseKmerIndex against
    the oracle / NumPy on random indexes and reads (N, lower case, ragged lengths)."""
    from graph_kmer_index_amd import CollisionFreeKmerIndex, FlatKmers, ReverseKmerIndex, DeviceFlatKmers
    from graph_kmer_index_amd.collision_free_kmer_index import PartitionedDeviceIndex
    comp = str.maketrans("ACGTacgt", "TGCAtgca")
    t_end = time.time() + args.seconds
    it = 0
    while time.time() < t_end:
        seed = args.seed * 1_000_003 + it
        it += 1
        progress("... %d rounds" % it)
        rng = np.random.default_rng(seed)
        k = int(rng.integers(1, 32))
        L = int(rng.integers(k + 5, 4000))
        genome = "".join("ACGT"[i] for i in rng.integers(0, 4, size=L))
        hashes = oracle.hash_sequence(oracle.letter_sequence_to_numeric(genome), k).astype(np.int64)
        n = len(hashes)
        nodes = rng.integers(0, int(rng.integers(1, 300)), size=n).astype(np.uint32)
        n_nodes = int(nodes.max()) + 1
        refs = rng.integers(0, 20, size=n).astype(np.uint64)
        af = np.ones(n, np.float32)
        modulo = int(rng.choice([1, 7, 101, 10007, 452930477]))
        idx = CollisionFreeKmerIndex.from_flat_kmers(FlatKmers(hashes, nodes, refs, af), modulo=modulo)
        orc = oracle.index_build(hashes, nodes, refs, af, modulo=modulo)
        reads = []
        for _ in range(int(rng.integers(1, 40))):
            a = int(rng.integers(0, L))
            r = list(genome[a:a + int(rng.integers(0, 260))])
            for p in rng.integers(0, max(1, len(r)), size=int(rng.integers(0, 4))):
                if r:
                    r[int(p)] = "NnacgtRYACGT"[int(rng.integers(0, 12))]
            s_ = "".join(r)
            reads.append(s_.translate(comp)[::-1] if rng.random() < 0.5 else s_)
        max_hits = int(rng.choice([1, 3, 2 ** 62]))
        both = bool(rng.integers(0, 2))
        exp = np.zeros(n_nodes, np.int64)
        for r in reads:
            for strand_read in ([r, r.translate(comp)[::-1]] if both else [r]):
                for q in oracle.read_kmers(strand_read, k):
                    got = oracle.index_get(orc, int(q), max_hits)[0]
                    if got is not None:
                        np.add.at(exp, np.asarray(got, np.int64), 1)
        desc = "seed %d k=%d modulo=%d max_hits=%d both=%s" % (seed, k, modulo, max_hits, both)
        got = idx.map_reads(reads, k, n_nodes, max_hits=max_hits, include_reverse_complement=both)
        if not np.array_equal(got, exp):
            print("MISMATCH map_reads", desc); sys.exit(1)
        # partitioned slices count like the whole index
        parts = PartitionedDeviceIndex.build(DeviceFlatKmers.from_flat_kmers(FlatKmers(hashes, nodes, refs, af)), modulo,
                                             int(rng.integers(1, min(modulo, 9) + 1)))
        enc = [r.encode() for r in reads]
        start = np.concatenate([[0], np.cumsum([len(e) for e in enc])]).astype(np.int64)
        letters = np.frombuffer(b"".join(enc), np.uint8) if start[-1] else np.zeros(1, np.uint8)
        c, _, _ = parts.count_nodes_from_reads(letters, start, k, n_nodes, 3 if both else 1, max_hits)
        if not np.array_equal(c.to_host(n_nodes), exp):
            print("MISMATCH partitioned map_reads", desc); sys.exit(1)
        parts.free()
        # reverse index
        rev = ReverseKmerIndex.from_flat_kmers(FlatKmers(hashes, nodes, refs, af))
        order = np.argsort(nodes, kind="stable")
        if not (np.array_equal(rev.hashes, hashes[order]) and np.array_equal(rev.ref_positions, refs[order])):
            print("MISMATCH reverse index", desc); sys.exit(1)
    print("soak reads ok: %d rounds" % it)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--what", choices=["find", "forward", "index", "reads"], default="find")
    args = ap.parse_args()
    if args.what != "find":
        return {"forward": soak_forward, "index": soak_index, "reads": soak_reads}[args.what](args)
    t_end = time.time() + args.seconds
    it = checked = refused = asserted = 0
    why = {}
    while time.time() < t_end:
        seed = args.seed * 1_000_003 + it
        it += 1
        progress("... %d drawn, %d compared" % (it, checked))
        rng = np.random.default_rng(seed)
        k = int(rng.integers(2, 32))
        M = int(rng.choice([0, 1, 2, 3, 4, 5, 100]))
        one = bool(rng.integers(0, 2))
        try:
            mode, g = make_graph(rng, k)
        except AssertionError:
            continue
        desc = "seed %d mode k=%d M=%d one=%s" % (seed, k, M, one)
        try:
            crit = oracle.critical_paths(g, k)
        except oracle.OracleError:
            continue                                              # reference crash E2
        kw = {}
        if len(crit[0]) > 2 and rng.random() < 0.4:
            a = int(rng.integers(0, len(crit[0])))
            b = int(rng.integers(a, len(crit[0]) + 1))
            kw = dict(start_at_critical_path_number=a, stop_at_critical_path_number=b)
        if rng.random() < 0.15:                                   # kmer_finder.py:386-388 in find()
            variant = np.nonzero(g.is_ref == 0)[0]
            if len(variant):
                kw = dict(kw, only_follow_nodes=set(int(x) for x in rng.choice(variant, size=max(1, len(variant) // 4), replace=False)))
        try:
            full, flags = oracle.find(g, k, crit, one, M, return_flags=True, **kw)
        except oracle.OracleError as e:
            if e.code != 3:
                continue                                          # the reference hits its recursion limit here
            # the reference's `assert len(next_nodes) == 1` (kmer_finder.py:402): the library must raise too
            f = DenseKmerFinder(g, k, critical_graph_paths=CriticalGraphPaths(crit[0], crit[1]), only_save_one_node_per_kmer=one,
                                max_variant_nodes=M, **kw)
            try:
                f.find()
            except AssertionError:
                asserted += 1
                f.close()
                continue
            except ValueError:
                continue                                          # (also undefined in the reference)
            print("library did not raise where the reference asserts:", desc, mode, kw); sys.exit(1)
        wl = None
        if len(full["kmers"]) and rng.random() < 0.25:
            wl = set(int(x) for x in full["kmers"][rng.random(len(full["kmers"])) < 0.5])
            full = oracle.find(g, k, crit, one, M, whitelist=wl, **kw)
        elif rng.random() < 0.15 and "only_follow_nodes" not in kw:
            osn = set(int(x) for x in np.nonzero(rng.random(g.n_nodes) < 0.4)[0])
            kw = dict(kw, only_store_nodes=osn)                   # kmer_finder.py:153 (the bulk path ignores it, :370-374)
            full = oracle.find(g, k, crit, one, M, **kw)
        cp = CriticalGraphPaths(crit[0], crit[1])
        f = DenseKmerFinder(g, k, critical_graph_paths=cp, only_save_one_node_per_kmer=one, max_variant_nodes=M, whitelist=wl, **kw)
        try:
            f.find()
        except (ValueError, NotImplementedError) as e:
            if flags & oracle.ORC_FLAG_UNDEFINED_BULK or isinstance(e, NotImplementedError):
                refused += 1
                reason = type(e).__name__ + ": " + ("reference output undefined (bulk path entered with < k bases)"
                                                    if isinstance(e, ValueError) else str(e)[:70])
                why[reason] = why.get(reason, 0) + 1
                continue
            print("UNEXPECTED refusal:", desc, mode, kw, e); sys.exit(1)
        if flags & oracle.ORC_FLAG_UNDEFINED_BULK:
            print("library accepted a graph the oracle flags as undefined:", desc, mode, kw); sys.exit(1)
        try:
            assert_same_records(finder_cols(f), full)
        except AssertionError as e:
            print("MISMATCH:", desc, mode, kw, "whitelist" if wl else "", e); sys.exit(1)
        if rng.random() < 0.35 and wl is None:
            # the FlatKmers emit path (other kernels' format, split and by-node layouts) holds the same records
            pos = g.position_id_base()[full["start_nodes"]] + full["start_offsets"]
            want = (full["kmers"].astype(np.uint64), full["nodes"].astype(np.uint32), pos.astype(np.uint64),
                    full["allele_frequencies"].astype(np.float32))
            ow = np.lexsort((want[3], want[1], want[0], want[2]))
            for split in (False, True):
                d = f.find_flat_on_device(split_layout=split)
                fl = d.to_flat_kmers()
                got = (fl._hashes, fl._nodes, fl._ref_offsets, fl._allele_frequencies)
                og = np.lexsort((got[3], got[1], got[0], got[2]))
                if len(got[0]) != len(want[0]) or not all(np.array_equal(a_[og], b_[ow]) for a_, b_ in zip(got, want)):
                    print("MISMATCH flat layout split=%s:" % split, desc, mode, kw); sys.exit(1)
                d.free()
        if not kw and wl is None and rng.random() < 0.25:        # (whole runs without node sets only)
            # the multi-GPU shards (sharding.critical_path_cuts) partition the full run
            from graph_kmer_index_amd.sharding import critical_path_cuts
            world = int(rng.integers(2, 7))
            cuts = critical_path_cuts(g, cp, world)
            parts = []
            try:
                for a_, b_ in zip(cuts[:-1], cuts[1:]):
                    fs = DenseKmerFinder(g, k, critical_graph_paths=cp, only_save_one_node_per_kmer=one, max_variant_nodes=M,
                                         start_at_critical_path_number=a_, stop_at_critical_path_number=b_)
                    fs.find()
                    parts.append(finder_cols(fs))
                    fs.close()
            except NotImplementedError:
                parts = None                                      # chunking needs node ids increasing along edges
            if parts:
                cat = {key: np.concatenate([p_[key] for p_ in parts]) for key in parts[0]}
                try:
                    assert_same_records(cat, full)
                except AssertionError as e:
                    print("MISMATCH shards do not partition the run:", desc, mode, "world", world, cuts, e); sys.exit(1)
        f.close()
        checked += 1
    print("soak ok: %d graphs compared, %d raised the reference's linear-successor assertion like the oracle, %d refused "
          "(undefined in the reference / unsupported), %d drawn" % (checked, asserted, refused, it))
    for reason, count in sorted(why.items(), key=lambda kv: -kv[1]):
        print("   refused %5d x %s" % (count, reason))


if __name__ == "__main__":
    main()
