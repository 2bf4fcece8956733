"""Live cross-check: oracle and brute-force spec against the reference itself (build container only;
skipped where /root/reference is absent).  Uses the test-only third-party stand-ins."""
import logging
import os
import sys
from collections import Counter
import numpy as np
import pytest

from conftest import HAVE_REFERENCE, REFERENCE, ROOT

pytestmark = pytest.mark.reference

if HAVE_REFERENCE:
    sys.path[:0] = [os.path.join(ROOT, "tests", "standins"), REFERENCE]
    logging.disable(logging.CRITICAL)
    from graph_kmer_index.kmer_finder import DenseKmerFinder
    from graph_kmer_index.critical_graph_paths import CriticalGraphPaths
    from obgraph import Graph

from graph_kmer_index_amd.graph import GraphArrays
from graphgen import random_bubble_graph, overlapping_bubble_graph
from spec_bruteforce import spec_rows
from oracle import oracle


def _make(rng, mode, k):
    if mode == "bubble":
        return random_bubble_graph(rng, p_indel=float(rng.choice([0.0, 0.5])), with_af=True)
    if mode == "overlap":
        return overlapping_bubble_graph(rng)
    nv = int(rng.integers(1, 4))
    return random_bubble_graph(rng, n_var=nv, min_ref=1, max_ref=3 * k + 8, p_indel=0.3,
                               chain_after={int(rng.integers(-1, nv)): int(rng.integers(1, k + 2))})


@pytest.mark.parametrize("mode,n,seed", [("bubble", 150, 1), ("overlap", 100, 2), ("chain", 250, 3)])
def test_oracle_and_spec_match_reference(mode, n, seed):
    rng = np.random.default_rng(seed)
    checked = 0
    for _ in range(n):
        k = int(rng.integers(3, 8))
        M = int(rng.choice([0, 1, 2, 3, 4, 100]))
        one = bool(rng.integers(0, 2))
        seqs, edges, lin, af = _make(rng, mode, k)
        g = GraphArrays.from_dicts(seqs, edges, lin, af)
        rg = Graph.from_dicts(seqs, edges, lin, af)
        try:
            cp = CriticalGraphPaths.from_graph(rg, k)
        except OverflowError:                      # reference crash E2 (SURVEY.md 8a')
            with pytest.raises(oracle.OracleError):
                oracle.critical_paths(g, k)
            continue
        ocn, oco = oracle.critical_paths(g, k)
        assert np.array_equal(ocn, cp.nodes) and np.array_equal(oco, cp.offsets)
        f = DenseKmerFinder(rg, k, critical_graph_paths=cp, max_variant_nodes=M, only_save_one_node_per_kmer=one)
        f.find()
        fl = f.get_flat_kmers()
        o, flags = oracle.find(g, k, (cp.nodes, cp.offsets), one, M, return_flags=True)
        # exact emission order, all five columns
        assert np.array_equal(o["kmers"], fl._hashes) and np.array_equal(o["nodes"], fl._nodes)
        assert np.array_equal(o["start_nodes"], fl._start_nodes)
        assert np.array_equal(o["start_offsets"], fl._start_offsets)
        assert np.array_equal(o["allele_frequencies"], fl._allele_frequencies)
        if flags & oracle.ORC_FLAG_UNDEFINED_BULK:
            continue                               # reference emits meaningless hashes here
        crit = {int(a): int(b) for a, b in zip(cp.nodes, cp.offsets)}
        ref_rows = Counter(zip(fl._hashes.tolist(), fl._start_nodes.tolist(), fl._start_offsets.tolist(),
                               fl._nodes.tolist(), fl._allele_frequencies.tolist()))
        assert spec_rows(g, k, M, one, crit) == ref_rows
        checked += 1
    assert checked > n // 2


def test_chunked_find_equals_unchunked():
    # command_line_interface.py:588-601 chunking over critical-path ranges
    rng = np.random.default_rng(11)
    for _ in range(30):
        k = int(rng.integers(3, 6))
        seqs, edges, lin, af = random_bubble_graph(rng, n_var=7, min_ref=k, max_ref=3 * k, p_indel=0.3)
        g = GraphArrays.from_dicts(seqs, edges, lin)
        cn, co = oracle.critical_paths(g, k)
        full = oracle.find(g, k, (cn, co), True, 5)
        n = len(cn)
        cuts = [0, n // 3, 2 * n // 3, n]
        parts = [oracle.find(g, k, (cn, co), True, 5, start_at_critical_path_number=a,
                             stop_at_critical_path_number=b) for a, b in zip(cuts[:-1], cuts[1:])]
        for key in ("kmers", "nodes", "start_nodes", "start_offsets"):
            assert np.array_equal(np.concatenate([p[key] for p in parts]), full[key])


def test_indel_generator_graphs_match_reference():
    # the SNP/indel graph of the full-size parity test (graph_kmer_index_amd.graph.synthetic_indel_graph): the
    # reference on the same graph (through the obgraph stand-in) against the oracle, emission order, all columns
    from graph_kmer_index_amd.graph import synthetic_indel_graph
    letters = "ACGT"
    for seed, k, M, one in [(1, 5, 4, True), (2, 7, 5, True), (3, 6, 2, False), (4, 7, 5, False), (5, 4, 1, True)]:
        g = synthetic_indel_graph(1500, 70, k=k, seed=seed, p_del=0.3, p_ins=0.3)
        seqs = {n: "".join(letters[c] for c in g.seq[g.seq_start[n]:g.seq_start[n + 1]]) for n in range(g.n_nodes)}
        edges = {n: g.edges[g.edge_start[n]:g.edge_start[n + 1]].tolist() for n in range(g.n_nodes)
                 if g.edge_start[n + 1] > g.edge_start[n]}
        lin = [n for n in range(g.n_nodes) if g.is_ref[n] and g.node_size[n] > 0]
        af = {n: float(g.allele_freq[n]) for n in range(g.n_nodes)}
        rg = Graph.from_dicts(seqs, edges, lin, af)
        g2 = GraphArrays.from_dicts(seqs, edges, lin, af)
        assert np.array_equal(g2.is_ref, g.is_ref)                # the adapter infers the same ref-dummy flags
        cp = CriticalGraphPaths.from_graph(rg, k)
        f = DenseKmerFinder(rg, k, critical_graph_paths=cp, max_variant_nodes=M, only_save_one_node_per_kmer=one)
        f.find()
        fl = f.get_flat_kmers()
        o, flags = oracle.find(g, k, (cp.nodes, cp.offsets), one, M, return_flags=True)
        assert flags == 0
        assert np.array_equal(o["kmers"], fl._hashes) and np.array_equal(o["nodes"], fl._nodes)
        assert np.array_equal(o["start_nodes"], fl._start_nodes) and np.array_equal(o["start_offsets"], fl._start_offsets)
        assert np.array_equal(o["allele_frequencies"], fl._allele_frequencies)


@pytest.mark.parametrize("mode,n,seed", [("nested", 80, 5), ("deep", 80, 6), ("follow", 60, 7), ("nested_chain", 120, 8)])
def test_nested_graphs_oracle_and_general_spec_match_reference(mode, n, seed):
    """Graphs with nodes that have no linear-ref predecessor, and find() with only_follow_nodes (kmer_finder.py:386-388):
    the reference itself, the oracle (exact order) and the order-free general rule the kernels implement
    (tests/spec_general.py), including where the reference's `assert len(next_nodes) == 1` (:402) fires."""
    from graphgen import nested_bubble_graph, deep_nested_graph
    import spec_general
    rng = np.random.default_rng(seed)
    n_rows = n_assert = 0
    for it in range(n):
        k = int(rng.integers(3, 8))
        M = int(rng.choice([0, 1, 2, 3, 4, 100]))
        one = bool(rng.integers(0, 2))
        if mode == "nested_chain":           # single-edge chains behind nested bubbles: lossy restarts (E1) next to them
            k = int(rng.integers(3, 16))
            M = int(rng.choice([1, 2, 3, 4]))
            seqs, edges, lin, af = nested_bubble_graph(rng, n_var=int(rng.integers(2, 6)), min_ref=1, max_ref=12, p_nest=0.6,
                                                       p_chain=0.5)
        elif mode == "deep":
            seqs, edges, lin, af = deep_nested_graph(rng, n_var=int(rng.integers(1, 4)), max_depth=int(rng.integers(1, 4)))
        elif mode == "nested" or it % 2:
            seqs, edges, lin, af = nested_bubble_graph(rng, n_var=int(rng.integers(2, 5)), p_nest=0.7)
        else:
            seqs, edges, lin, af = random_bubble_graph(rng)
        g = GraphArrays.from_dicts(seqs, edges, lin, af)
        rg = Graph.from_dicts(seqs, edges, lin, af)
        follow = None
        if mode == "follow":
            cand = [x for x in seqs if not g.is_ref[x]]
            follow = set(int(x) for x in rng.choice(cand, size=max(1, len(cand) // 3), replace=False))
        try:
            cp = CriticalGraphPaths.from_graph(rg, k)
        except (OverflowError, AssertionError):
            continue
        crit = {int(a): int(b) for a, b in zip(cp.nodes, cp.offsets)}
        f = DenseKmerFinder(rg, k, critical_graph_paths=cp, max_variant_nodes=M, only_save_one_node_per_kmer=one,
                            only_follow_nodes=None if follow is None else set(follow))
        try:
            f.find()
        except AssertionError:
            with pytest.raises(oracle.OracleError):
                oracle.find(g, k, (cp.nodes, cp.offsets), one, M, only_follow_nodes=follow)
            with pytest.raises(spec_general.SpecError):
                spec_general.spec_rows_general(g, k, M, one, critical=crit, follow=follow)
            n_assert += 1
            continue
        fl = f.get_flat_kmers()
        o, flags = oracle.find(g, k, (cp.nodes, cp.offsets), one, M, only_follow_nodes=follow, return_flags=True)
        if follow is None:                         # with a follow set the reference iterates a Python set: order unpinned
            assert np.array_equal(o["kmers"], fl._hashes) and np.array_equal(o["nodes"], fl._nodes)
            assert np.array_equal(o["start_nodes"], fl._start_nodes) and np.array_equal(o["start_offsets"], fl._start_offsets)
        ref_rows = Counter(zip(fl._hashes.tolist(), fl._start_nodes.tolist(), fl._start_offsets.tolist(),
                               fl._nodes.tolist(), fl._allele_frequencies.tolist()))
        assert Counter(zip(o["kmers"].tolist(), o["start_nodes"].tolist(), o["start_offsets"].tolist(),
                           o["nodes"].tolist(), o["allele_frequencies"].tolist())) == ref_rows
        if flags & oracle.ORC_FLAG_UNDEFINED_BULK:
            continue
        assert spec_general.spec_rows_general(g, k, M, one, critical=crit, follow=follow) == ref_rows
        n_rows += 1
    assert n_rows > n // 3 and (mode == "follow" or n_assert > 0)
