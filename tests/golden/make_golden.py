"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference; the test-only stand-ins under
tests/standins/ replace the absent third-party packages obgraph / npstructures / Bio / pyfaidx):

    python tests/golden/make_golden.py

Fixtures are data only: graph literals / generator parameters and the reference's outputs.
Nothing of the reference's source is stored.  The GPU box has no /root/reference; tests there
read these files.
"""
import hashlib
import json
import logging
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "tests", "standins"), "/root/reference", ROOT, os.path.join(ROOT, "tests")]
logging.disable(logging.CRITICAL)

import numpy as np  # noqa: E402
from graph_kmer_index.kmer_finder import DenseKmerFinder  # noqa: E402
from graph_kmer_index.critical_graph_paths import CriticalGraphPaths  # noqa: E402
from graph_kmer_index.flat_kmers import FlatKmers  # noqa: E402
from graph_kmer_index.collision_free_kmer_index import CollisionFreeKmerIndex  # noqa: E402
from graph_kmer_index.kmer_hashing import (kmer_hashes_to_reverse_complement_hash,  # noqa: E402
                                           kmer_hashes_to_complement_hashes)
from graph_kmer_index import sequence_to_kmer_hash, kmer_hash_to_sequence, ReadKmers  # noqa: E402
from graph_kmer_index.kmer_hashing import power_array  # noqa: E402
from obgraph import Graph  # noqa: E402

from graph_kmer_index_amd.graph import GraphArrays, synthetic_linear_graph, synthetic_snp_graph  # noqa: E402
from graphgen import (random_bubble_graph, overlapping_bubble_graph, nested_bubble_graph, deep_nested_graph,  # noqa: E402
                      empty_chain_graph)
from golden_cases import REFERENCE_TEST_GRAPHS, canonical_digest  # noqa: E402


def run_finder(graph, k, **kw):
    early = kw.pop("from_position", None)
    cp = None
    if early is None:
        cp = CriticalGraphPaths.from_graph(graph, k)
    f = DenseKmerFinder(graph, k, critical_graph_paths=cp, **kw)
    if early is None:
        f.find()
    else:
        f.find_only_kmers_starting_at_position(*early)
    fl = f.get_flat_kmers()
    crit = ([], []) if cp is None else (cp.nodes.tolist(), cp.offsets.tolist())
    return fl, crit


def toy_cases():
    cases = []
    rng = np.random.default_rng(20240501)

    def add(name, seqs, edges, lin, k, af=None, **kw):
        g = Graph.from_dicts(seqs, edges, lin, af)
        kw2 = dict(kw)
        if "only_store_nodes" in kw2:
            kw2["only_store_nodes"] = set(kw2["only_store_nodes"])
        try:
            fl, crit = run_finder(g, k, **kw2)
        except OverflowError:
            cases.append(dict(name=name, seqs={str(a): b for a, b in seqs.items()},
                              edges={str(a): b for a, b in edges.items()}, linear=lin, k=k,
                              af=None if af is None else {str(a): b for a, b in af.items()},
                              kw=kw, raises="E2"))
            return
        cases.append(dict(
            name=name, seqs={str(a): b for a, b in seqs.items()}, edges={str(a): b for a, b in edges.items()},
            linear=lin, k=k, af=None if af is None else {str(a): b for a, b in af.items()}, kw=kw,
            crit_nodes=crit[0], crit_offsets=crit[1],
            kmers=fl._hashes.tolist(), nodes=fl._nodes.tolist(), start_nodes=fl._start_nodes.tolist(),
            start_offsets=fl._start_offsets.tolist(), allele_frequencies=fl._allele_frequencies.tolist()))

    for name, (seqs, edges, lin, k, kw) in REFERENCE_TEST_GRAPHS.items():
        add(name, seqs, edges, lin, k, **kw)
    for i in range(40):
        k = int(rng.integers(3, 8))
        seqs, edges, lin, af = random_bubble_graph(rng, p_indel=float(rng.choice([0.0, 0.5])), with_af=True)
        add("rand_bubble_%d" % i, seqs, edges, lin, k, af,
            max_variant_nodes=int(rng.choice([0, 1, 2, 3, 4, 100])),
            only_save_one_node_per_kmer=bool(rng.integers(0, 2)))
    for i in range(15):
        k = int(rng.integers(3, 8))
        seqs, edges, lin, af = overlapping_bubble_graph(rng)
        add("rand_overlap_%d" % i, seqs, edges, lin, k, af,
            max_variant_nodes=int(rng.choice([1, 2, 4])), only_save_one_node_per_kmer=bool(rng.integers(0, 2)))
    for i in range(30):
        k = int(rng.integers(3, 8))
        nv = int(rng.integers(1, 4))
        seqs, edges, lin, af = random_bubble_graph(
            rng, n_var=nv, min_ref=1, max_ref=3 * k + 8, p_indel=0.3,
            chain_after={int(rng.integers(-1, nv)): int(rng.integers(1, k + 2))})
        add("rand_chain_%d" % i, seqs, edges, lin, k, af,
            max_variant_nodes=int(rng.choice([1, 4])), only_save_one_node_per_kmer=bool(rng.integers(0, 2)))
    # chunked runs (command_line_interface.py:588-601): same graph, three critical-path ranges
    for i in range(6):
        k = int(rng.integers(3, 6))
        seqs, edges, lin, af = random_bubble_graph(rng, n_var=7, min_ref=k, max_ref=3 * k, p_indel=0.3)
        g = Graph.from_dicts(seqs, edges, lin)
        n_crit = len(CriticalGraphPaths.from_graph(g, k))
        cuts = [0, n_crit // 3, 2 * n_crit // 3, n_crit]
        for a, b in zip(cuts[:-1], cuts[1:]):
            add("rand_chunk_%d_%d_%d" % (i, a, b), seqs, edges, lin, k,
                start_at_critical_path_number=a, stop_at_critical_path_number=b,
                only_save_one_node_per_kmer=True, max_variant_nodes=5)
    # critical points at offset 0 (a single-edge chain of exactly k-1 bases before a critical node): the run before such a
    # point passes through it, the run starting there is not rewound (kmer_finder.py:231-232, 334-341) -- every chunk
    # [a, b) of two such graphs
    rng0 = np.random.default_rng(77)
    letters = "ACGT"

    def rand(n):
        return "".join(letters[i] for i in rng0.integers(0, 4, size=n))
    for gi, (k, sizes) in enumerate([(23, [22, 22, 1, 1, 77, 2, 0, 27]), (4, [3, 5, 1, 1, 9, 1, 0, 3, 6])]):
        if gi == 0:
            edges = {0: [1], 1: [3, 2], 2: [4], 3: [4], 4: [6, 5], 5: [7], 6: [7]}
            lin = [0, 1, 2, 4, 5, 7]
        else:
            edges = {0: [1], 1: [2, 3], 2: [4], 3: [4], 4: [5, 6], 5: [7], 6: [7], 7: [8]}
            lin = [0, 1, 2, 4, 5, 7, 8]
        seqs = {n: rand(sz) for n, sz in enumerate(sizes)}
        g = Graph.from_dicts(seqs, edges, lin)
        n_crit = len(CriticalGraphPaths.from_graph(g, k))
        for a in range(n_crit + 1):
            for b in range(a, n_crit + 1):
                for one in (True, False):
                    add("chunk_offset0_%d_%d_%d_%d" % (gi, a, b, int(one)), seqs, edges, lin, k,
                        start_at_critical_path_number=a, stop_at_critical_path_number=b,
                        only_save_one_node_per_kmer=one, max_variant_nodes=2)
    return cases


def nested_cases():
    """Graphs with nodes that have no linear-ref predecessor (a variant inside an alternative allele, multi-node
    alleles), k 3-7, every max_variant_nodes regime, both node modes, some with only_follow_nodes.  The reference's
    `assert len(next_nodes) == 1` (kmer_finder.py:402) fires on many of them at small limits: recorded as raises."""
    cases = []
    rng = np.random.default_rng(20241004)

    def add(name, seqs, edges, lin, k, M, one, follow=None):
        g = Graph.from_dicts(seqs, edges, lin)
        base = dict(name=name, seqs={str(a): b for a, b in seqs.items()}, edges={str(a): b for a, b in edges.items()},
                    linear=lin, k=k, M=M, one=one, follow=None if follow is None else sorted(follow))
        kw = dict(max_variant_nodes=M, only_save_one_node_per_kmer=one)
        if follow is not None:
            kw["only_follow_nodes"] = set(follow)
        try:
            fl, crit = run_finder(g, k, **kw)
        except AssertionError:
            cases.append(dict(base, raises=True))
            return
        cases.append(dict(base, raises=False, kmers=fl._hashes.tolist(), nodes=fl._nodes.tolist(),
                          start_nodes=fl._start_nodes.tolist(), start_offsets=fl._start_offsets.tolist(),
                          allele_frequencies=fl._allele_frequencies.tolist()))

    limits = [0, 1, 2, 3, 4, 100]
    for i in range(36):
        seqs, edges, lin, _ = nested_bubble_graph(rng, n_var=int(rng.integers(2, 5)), p_nest=0.7)
        add("nested_%d" % i, seqs, edges, lin, 3 + i % 5, limits[i % 6], bool((i // 6) % 2))
    for i in range(24):
        seqs, edges, lin, _ = deep_nested_graph(rng, n_var=int(rng.integers(1, 4)), max_depth=int(rng.integers(1, 4)))
        add("deep_%d" % i, seqs, edges, lin, 3 + i % 5, limits[(i + 3) % 6], bool(i % 2))
    for i in range(12):
        gen = nested_bubble_graph if i % 2 else random_bubble_graph
        seqs, edges, lin = gen(rng)[:3]
        g = Graph.from_dicts(seqs, edges, lin)
        cand = [n for n in seqs if not g.is_linear_ref_node_or_linear_ref_dummy_node(n)]
        follow = [int(x) for x in rng.choice(cand, size=max(1, len(cand) // 3), replace=False)]
        add("follow_%d" % i, seqs, edges, lin, 3 + i % 5, [1, 2, 100][i % 3], bool(i % 2), follow)
    # single-edge chains behind nested bubbles: critical points with a lossy restart (SURVEY.md 8a' E1) next to nodes
    # that carry the assertion
    for i in range(16):
        seqs, edges, lin, _ = nested_bubble_graph(rng, n_var=int(rng.integers(2, 6)), min_ref=1, max_ref=12, p_nest=0.6, p_chain=0.5)
        try:
            add("nested_chain_%d" % i, seqs, edges, lin, 4 + i % 10, [1, 2, 3, 4][i % 4], bool(i % 2))
        except OverflowError:
            pass                                      # reference crash E2 (critical offset -1)
    return cases


def medium_cases():
    """Generator-defined graphs at k=31; full columns for the linear one, digests for the SNP ones."""
    out = {}
    g = synthetic_linear_graph(20000, node_len=3000, seed=1234)
    for one in (False, True):
        fl, crit = run_finder(g, 31, only_save_one_node_per_kmer=one)
        tag = "linear20k_one%d" % one
        out[tag + "_kmers"] = fl._hashes
        out[tag + "_nodes"] = fl._nodes
        out[tag + "_start_nodes"] = fl._start_nodes
        out[tag + "_start_offsets"] = fl._start_offsets
        out[tag + "_af"] = fl._allele_frequencies
    out["linear20k_crit_nodes"] = np.array(crit[0], dtype=np.uint32)
    out["linear20k_crit_offsets"] = np.array(crit[1], dtype=np.uint16)
    meta = {}
    for name, (G, S, M, one) in {"snp100k_one1_M5": (100000, 400, 5, True),
                                 "snp100k_one0_M4": (100000, 400, 4, False),
                                 "snp30k_dense_one1_M2": (30000, 1500, 2, True),
                                 "snp30k_dense_one0_M5": (30000, 1500, 5, False)}.items():
        g = synthetic_snp_graph(G, S, k=31, seed=77)
        fl, crit = run_finder(g, 31, only_save_one_node_per_kmer=one, max_variant_nodes=M)
        cols = dict(kmers=fl._hashes, nodes=fl._nodes, start_nodes=fl._start_nodes,
                    start_offsets=fl._start_offsets, allele_frequencies=fl._allele_frequencies)
        meta[name] = dict(G=G, S=S, seed=77, k=31, M=M, one=one, n_records=int(len(fl._hashes)),
                          digest=canonical_digest(cols), n_crit=len(crit[0]),
                          graph_digest=hashlib.sha256(g.seq.tobytes() + g.node_size.tobytes()
                                                      + g.edges.tobytes()).hexdigest())
        out[name + "_head_kmers"] = fl._hashes[:3000]
        out[name + "_head_nodes"] = fl._nodes[:3000]
        out[name + "_head_start_nodes"] = fl._start_nodes[:3000]
        out[name + "_head_start_offsets"] = fl._start_offsets[:3000]
        out[name + "_head_af"] = fl._allele_frequencies[:3000]
        out[name + "_crit_nodes"] = np.array(crit[0], dtype=np.uint32)
        out[name + "_crit_offsets"] = np.array(crit[1], dtype=np.uint16)
    return out, meta


def hashing_cases():
    rng = np.random.default_rng(99)
    out = {}
    for k in (3, 9, 16, 31):
        h = rng.integers(0, 4 ** k, size=1000, dtype=np.uint64)
        out["rc_in_k%d" % k] = h
        out["rc_out_k%d" % k] = kmer_hashes_to_reverse_complement_hash(h.copy(), k).astype(np.uint64)
        out["comp_out_k%d" % k] = kmer_hashes_to_complement_hashes(h.copy(), k).astype(np.uint64)
    reads = ["".join("acgtACGTnN"[i] for i in rng.integers(0, 10, size=150)) for _ in range(20)]
    reads += ["ACGT" * 8, "a" * 31, "T" * 40]
    out["reads"] = np.array(reads)
    for k in (5, 31):
        pv = power_array(k)
        out["read_kmers_k%d" % k] = np.concatenate(
            [ReadKmers.get_kmers_from_read_dynamic(r, pv).astype(np.uint64) for r in reads])
    seqs = ["ACTG", "T" * 31, "CAtgAACAtttggtAATCTACAtgAACAttt", "G", "atg", "Acacatacgactacg"]
    out["kat_sequences"] = np.array(seqs)
    out["kat_hashes"] = np.array([sequence_to_kmer_hash(s) for s in seqs], dtype=np.uint64)
    assert all(kmer_hash_to_sequence(int(h), len(s)).lower() == s.lower() for s, h in zip(seqs, out["kat_hashes"]))
    return out


def index_cases():
    out = {}
    rng = np.random.default_rng(7)

    def add(tag, hashes, nodes, ref_offsets, af, modulo, **kw):
        flat = FlatKmers(hashes.astype(np.int64), nodes, ref_offsets, af)
        idx = CollisionFreeKmerIndex.from_flat_kmers(flat, modulo=modulo, **kw)
        out[tag + "_in_hashes"] = hashes.astype(np.int64)
        out[tag + "_in_nodes"] = nodes
        out[tag + "_in_ref_offsets"] = ref_offsets
        out[tag + "_in_af"] = af
        out[tag + "_modulo"] = np.int64(modulo)
        for name in ("_hashes_to_index", "_n_kmers", "_nodes", "_ref_offsets", "_kmers", "_frequencies",
                     "_allele_frequencies"):
            out[tag + name] = np.asarray(getattr(idx, name))
        # probes: every distinct kmer + misses, with two max_hits settings
        queries = np.concatenate([np.unique(hashes), rng.integers(0, 4 ** 15, size=50)]).astype(np.int64)
        out[tag + "_queries"] = queries
        for mh in (10, 1):
            hit_n, hit_nodes, hit_ro, hit_fr, hit_af = [], [], [], [], []
            for q in queries:
                r = idx.get(int(q), max_hits=mh)
                if r[0] is None:
                    hit_n.append(-1)
                    continue
                hit_n.append(len(r[0]))
                hit_nodes.append(r[0]); hit_ro.append(r[1]); hit_fr.append(r[2]); hit_af.append(r[3])
            cat = (lambda x, dt: np.concatenate(x).astype(dt) if x else np.zeros(0, dt))
            out[tag + "_get%d_n" % mh] = np.array(hit_n, dtype=np.int64)
            out[tag + "_get%d_nodes" % mh] = cat(hit_nodes, np.int64)
            out[tag + "_get%d_ref_offsets" % mh] = cat(hit_ro, np.int64)
            out[tag + "_get%d_frequencies" % mh] = cat(hit_fr, np.int64)
            out[tag + "_get%d_af" % mh] = cat(hit_af, np.float64)

    # fixture of tests/test_collision_free_kmer_index.py:6-15 (with int64 hashes, SURVEY.md 8c caveat 1)
    add("kat", np.array([1, 1, 2, 2, 4, 5, 3]), np.array([5, 6, 7, 8, 10, 11, 100]),
        np.array([1, 1, 2, 3, 10, 11, 100]), np.ones(7, dtype=np.float32), 4)
    n = 4000
    hashes = rng.integers(0, 600, size=n).astype(np.int64) * 7919
    nodes = rng.integers(0, 5000, size=n).astype(np.uint32)
    ref = rng.integers(0, 40, size=n).astype(np.uint64) + (hashes % 11).astype(np.uint64)
    af = rng.uniform(0, 1, size=n).astype(np.float32)
    add("rand", hashes, nodes, ref, af, 1009)
    add("rand_skipfreq", hashes, nodes, ref, af, 1009, skip_frequencies=True)
    add("rand_nosingle", hashes, nodes, ref, af, 257, skip_singletons=True)
    return out


def two_chromosome_cases():
    """Graphs with two chromosomes (two components, `chromosome_start_nodes` lists both starts).  The reference walks
    every chromosome for critical points (critical_graph_paths.py:52-53) but prepends the extra start point only for
    the graph's first node (kmer_finder.py:208-211): a later chromosome whose start node is shorter than k begins to
    emit at its first critical point."""
    cases = []
    rng = np.random.default_rng(20261004)

    def shifted(part, shift):
        seqs, edges, lin, af = part
        return ({n + shift: s for n, s in seqs.items()}, {n + shift: [m + shift for m in e] for n, e in edges.items()},
                [n + shift for n in lin], {n + shift: f for n, f in (af or {m: 1.0 for m in seqs}).items()})

    def add(name, parts, k, **kw):
        seqs, edges, lin, af, starts = {}, {}, [], {}, []
        for part in parts:
            s2, e2, l2, a2 = shifted(part, len(seqs))
            starts.append(min(s2))
            seqs.update(s2); edges.update(e2); lin += l2; af.update(a2)
        g = Graph.from_dicts(seqs, edges, lin, af, chromosome_start_nodes=starts)
        base = dict(name=name, seqs={str(a): b for a, b in seqs.items()}, edges={str(a): b for a, b in edges.items()},
                    linear=lin, k=k, af={str(a): b for a, b in af.items()}, kw=kw, chromosome_start_nodes=starts)
        try:
            fl, crit = run_finder(g, k, **kw)
        except OverflowError:
            cases.append(dict(base, raises="E2"))
            return
        cases.append(dict(base, crit_nodes=crit[0], crit_offsets=crit[1], kmers=fl._hashes.tolist(), nodes=fl._nodes.tolist(),
                          start_nodes=fl._start_nodes.tolist(), start_offsets=fl._start_offsets.tolist(),
                          allele_frequencies=fl._allele_frequencies.tolist()))

    i = 0
    for k in (3, 4, 5, 7):
        for first2 in (1, 2, k - 1, k, k + 4):             # start node of chromosome 2: shorter than k, exactly k, longer
            if first2 < 1:
                continue
            for variant in range(2):
                c1 = random_bubble_graph(rng, n_var=int(rng.integers(1, 4)), min_ref=k, max_ref=2 * k + 2, p_indel=0.3,
                                         first_ref=k + 2, with_af=True)
                chain = None if variant == 0 else {-1: int(rng.integers(1, max(2, first2)))}
                c2 = random_bubble_graph(rng, n_var=int(rng.integers(1, 5)), min_ref=1, max_ref=2 * k + 2, p_indel=0.4,
                                         first_ref=first2, with_af=True, chain_after=chain)
                for one in (True, False):
                    add("two_chrom_%d_k%d_f%d_v%d_%d" % (i, k, first2, variant, int(one)), [c1, c2], k,
                        only_save_one_node_per_kmer=one, max_variant_nodes=int(rng.integers(1, 6)))
                i += 1
    # three chromosomes, the middle one a single short node (never emits), and chunked runs over a two-chromosome graph
    for j in range(4):
        k = int(rng.integers(3, 6))
        c1 = random_bubble_graph(rng, n_var=2, min_ref=k, max_ref=2 * k, first_ref=k + 1)
        mid = ({0: "ACGT"[:k - 1][:max(1, k - 2)]}, {}, [0], {0: 1.0})
        c3 = random_bubble_graph(rng, n_var=3, min_ref=1, max_ref=2 * k, first_ref=int(rng.integers(1, k)))
        add("three_chrom_%d" % j, [c1, mid, c3], k, only_save_one_node_per_kmer=bool(j % 2), max_variant_nodes=4)
        seqs = {}
        parts = [c1, c3]
        g_parts = []
        for part in parts:
            g_parts.append(shifted(part, sum(len(p[0]) for p in g_parts)))
        all_seqs = {}
        all_edges, all_lin = {}, []
        for sq, ed, li, _ in g_parts:
            all_seqs.update(sq); all_edges.update(ed); all_lin += li
        n_crit = len(CriticalGraphPaths.from_graph(Graph.from_dicts(all_seqs, all_edges, all_lin,
                                                                    chromosome_start_nodes=[min(p[0]) for p in g_parts]), k))
        for a, b in ((0, n_crit // 2), (n_crit // 2, n_crit)):
            add("two_chrom_chunk_%d_%d_%d" % (j, a, b), parts, k, start_at_critical_path_number=a,
                stop_at_critical_path_number=b, only_save_one_node_per_kmer=True, max_variant_nodes=5)
    return cases


def deep_window_cases():
    """k-windows crossing far more nodes than the GPU kernels' scratch stacks hold (GKI_MAX_WINDOW_NODES = 48): runs of
    insertion sites / empty nodes with nothing between them (graphgen.empty_chain_graph).  The reference recurses per node
    (sys.setrecursionlimit(20000), kmer_finder.py); the GPU library answers these through its slow path."""
    cases = []
    rng = np.random.default_rng(20261005)

    def add(name, graph, k, **kw):
        seqs, edges, lin, af = graph
        g = Graph.from_dicts(seqs, edges, lin, af)
        base = dict(name=name, seqs={str(a): b for a, b in seqs.items()}, edges={str(a): b for a, b in edges.items()},
                    linear=lin, k=k, af=None if af is None else {str(a): b for a, b in af.items()}, kw=kw)
        fl, crit = run_finder(g, k, **kw)
        cols = dict(kmers=fl._hashes, nodes=fl._nodes, start_nodes=fl._start_nodes, start_offsets=fl._start_offsets,
                    allele_frequencies=fl._allele_frequencies)
        rec = dict(base, crit_nodes=crit[0], crit_offsets=crit[1], n_records=int(len(fl._hashes)), digest=canonical_digest(cols))
        if len(fl._hashes) <= 1500:                    # small cases carry their records, the others count + digest
            rec.update({name2: np.asarray(v).tolist() for name2, v in cols.items()})
        cases.append(rec)

    i = 0
    for n_sites, k, m in ((20, 31, 2), (46, 31, 1), (47, 12, 2), (50, 31, 1), (70, 31, 1), (70, 6, 2), (120, 31, 1), (120, 9, 1),
                          (200, 31, 1), (200, 4, 2), (400, 31, 0), (400, 20, 1)):
        for one in (True, False):
            graph = empty_chain_graph(rng, n_sites, first_ref=int(rng.integers(3, 40)), last_ref=int(rng.integers(k, 3 * k)),
                                      with_af=bool(i % 2), p_plain=float(rng.choice([0.0, 0.2, 0.6])))
            add("deep_%d_sites%d_k%d_m%d_%d" % (i, n_sites, k, m, int(one)), graph, k, only_save_one_node_per_kmer=one, max_variant_nodes=m)
            i += 1
    # early-stop searches (find_only_kmers_starting_at_position, kmer_finder.py:170) whose first k-mer lies beyond a deep run
    for j, (n_sites, k, m, one) in enumerate(((60, 31, 1, True), (60, 31, 2, False), (100, 20, 1, False), (150, 31, 0, True),
                                              (150, 12, 2, False), (300, 31, 1, True))):
        first = int(rng.integers(6, k))
        graph = empty_chain_graph(rng, n_sites, first_ref=first, last_ref=2 * k, with_af=True, p_plain=float(rng.choice([0.0, 0.4])))
        for off in (0, first - 1):
            add("deep_from_position_%d_%d" % (j, off), graph, k, from_position=[0, off], only_save_one_node_per_kmer=one, max_variant_nodes=m)
    # two deep runs in one graph with ordinary bubbles between them, chunked by critical points
    for j in range(3):
        k = int(rng.choice([7, 15, 31]))
        a = empty_chain_graph(rng, 60, first_ref=k + 3, last_ref=2 * k, with_af=True)
        b = random_bubble_graph(rng, n_var=4, min_ref=2, max_ref=2 * k, first_ref=k, with_af=True)
        c = empty_chain_graph(rng, 90, first_ref=5, last_ref=k + 5, with_af=True, p_plain=0.5)
        seqs, edges, lin, af = {}, {}, [], {}
        last_tail = None
        for part in (a, b, c):
            shift = len(seqs)
            ps, pe, pl, pa = part
            for n2, sq in ps.items(): seqs[n2 + shift] = sq
            for n2, e2 in pe.items(): edges[n2 + shift] = [m2 + shift for m2 in e2]
            lin += [n2 + shift for n2 in pl]
            for n2, f2 in pa.items(): af[n2 + shift] = f2
            if last_tail is not None:
                edges[last_tail] = [shift]              # the previous part's end node continues into this part's first
            last_tail = shift + len(ps) - 1
        g = Graph.from_dicts(seqs, edges, lin, af)
        n_crit = len(CriticalGraphPaths.from_graph(g, k))
        add("deep_mixed_%d" % j, (seqs, edges, lin, af), k, only_save_one_node_per_kmer=bool(j % 2), max_variant_nodes=1)
        for lo, hi in ((0, max(1, n_crit // 2)), (max(1, n_crit // 2), n_crit)):
            add("deep_mixed_chunk_%d_%d_%d" % (j, lo, hi), (seqs, edges, lin, af), k, start_at_critical_path_number=lo,
                stop_at_critical_path_number=hi, only_save_one_node_per_kmer=True, max_variant_nodes=2)
    return cases


def main():
    if sys.argv[1:] == ["deep"]:
        deep = deep_window_cases()
        with open(os.path.join(HERE, "finder_deep.json"), "w") as f:
            json.dump(deep, f, separators=(",", ":"))
        print("deep-window cases:", len(deep), "records:", [c["n_records"] for c in deep])
        return
    if sys.argv[1:] == ["two_chrom"]:
        two = two_chromosome_cases()
        with open(os.path.join(HERE, "finder_two_chrom.json"), "w") as f:
            json.dump(two, f, separators=(",", ":"))
        print("two-chromosome cases:", len(two), "raising:", sum("raises" in c for c in two))
        return
    nested = nested_cases()
    with open(os.path.join(HERE, "finder_nested.json"), "w") as f:
        json.dump(nested, f, separators=(",", ":"))
    print("nested cases:", len(nested), "raising:", sum(c["raises"] for c in nested))
    if sys.argv[1:] == ["nested"]:
        return
    cases = toy_cases()
    with open(os.path.join(HERE, "finder_toy.json"), "w") as f:
        json.dump(cases, f, separators=(",", ":"))
    med, meta = medium_cases()
    np.savez_compressed(os.path.join(HERE, "finder_medium.npz"), **med)
    with open(os.path.join(HERE, "finder_medium_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    np.savez_compressed(os.path.join(HERE, "hashing.npz"), **hashing_cases())
    np.savez_compressed(os.path.join(HERE, "index.npz"), **index_cases())
    print("toy cases:", len(cases), "medium:", list(meta))


if __name__ == "__main__":
    main()
