"""CPU: the order-free rule for general graphs (tests/spec_general.py -- what the GENERAL kernels implement) and the
host-side node classification, against the oracle and the reference-generated nested fixtures."""
import json
import os
from collections import Counter
import numpy as np
import pytest

from graph_kmer_index_amd import GraphArrays
from graph_kmer_index_amd.kmer_finder import classify_nodes
from graphgen import nested_bubble_graph, deep_nested_graph, random_bubble_graph, overlapping_bubble_graph
from oracle import oracle
import spec_general

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rows(rec):
    return Counter(zip(rec["kmers"].tolist(), rec["start_nodes"].tolist(), rec["start_offsets"].tolist(),
                       rec["nodes"].tolist(), rec["allele_frequencies"].tolist()))


def oracle_rows(g, k, M, one, follow=None):
    """(Counter, None) or (None, 'assert'); None if the graph has no critical paths the reference accepts."""
    try:
        crit = oracle.critical_paths(g, k)
    except oracle.OracleError:
        return None
    critd = {int(n): int(c) for n, c in zip(*crit)}
    try:
        return rows(oracle.find(g, k, crit, one, M, only_follow_nodes=follow)), critd
    except oracle.OracleError as e:
        assert e.code == 3
        return "assert", critd


def spec(g, k, M, one, critd, follow=None):
    try:
        return spec_general.spec_rows_general(g, k, M, one, critical=critd, follow=follow)
    except spec_general.SpecError:
        return "assert"


def test_oracle_and_spec_match_reference_on_nested_fixtures():
    with open(os.path.join(GOLD, "finder_nested.json")) as fh:
        cases = json.load(fh)
    assert len(cases) >= 40 and {c["M"] for c in cases} == {0, 1, 2, 3, 4, 100} and {c["k"] for c in cases} >= {3, 4, 5, 6, 7}
    for case in cases:
        g = GraphArrays.from_dicts({int(a): b for a, b in case["seqs"].items()},
                                   {int(a): b for a, b in case["edges"].items()}, case["linear"])
        follow = None if case["follow"] is None else set(case["follow"])
        got, critd = oracle_rows(g, case["k"], case["M"], case["one"], follow)
        sp = spec(g, case["k"], case["M"], case["one"], critd, follow)
        if case["raises"]:
            assert got == "assert" and sp == "assert", case["name"]
            continue
        exp = Counter(zip(case["kmers"], case["start_nodes"], case["start_offsets"], case["nodes"],
                          case["allele_frequencies"]))
        assert got == exp, case["name"]
        assert sp == exp, case["name"]


@pytest.mark.parametrize("gen,seed", [("nested", 1), ("deep", 2), ("bubble", 3), ("overlap", 4)])
def test_spec_equals_oracle_on_random_graphs(gen, seed):
    rng = np.random.default_rng(seed)
    make = {"nested": lambda: nested_bubble_graph(rng, n_var=int(rng.integers(2, 5)), p_nest=0.7),
            "deep": lambda: deep_nested_graph(rng, n_var=int(rng.integers(1, 4)), max_depth=int(rng.integers(1, 4))),
            "bubble": lambda: random_bubble_graph(rng),
            "overlap": lambda: overlapping_bubble_graph(rng, n_var=int(rng.integers(3, 7)))}[gen]
    n_ok = n_assert = 0
    for it in range(12):
        seqs, edges, lin, af = make()
        g = GraphArrays.from_dicts(seqs, edges, lin, af)
        for k, M in ((3, 0), (4, 1), (5, 2), (6, 3), (7, 100), (5, 4)):
            one = bool(rng.integers(0, 2))
            follow = None
            if it % 3 == 2:
                cand = [n for n in seqs if not g.is_ref[n]]
                follow = set(int(x) for x in rng.choice(cand, size=max(1, len(cand) // 3), replace=False))
            res = oracle_rows(g, k, M, one, follow)
            if res is None:
                continue
            got, critd = res
            assert spec(g, k, M, one, critd, follow) == got, (seqs, edges, lin, k, M, one, follow)
            n_assert += got == "assert"
            n_ok += got != "assert"
    assert n_ok > 20 and (gen in ("bubble", "overlap") or n_assert > 0)


def _max_variant_nodes_before(g, flags, n, k):
    """max over backward paths (reachable nodes only) of the distinct non-linear-ref nodes within the k bases before n"""
    best = 0
    stack = [(n, 0, 0)]
    while stack:
        node, bases, cnt = stack.pop()
        best = max(best, cnt)
        if bases >= k:
            continue
        for p in g.rev_edges[g.rev_start[node]:g.rev_start[node + 1]].tolist():
            if flags[p] & spec_general.DEAD:
                continue
            stack.append((p, bases + int(g.node_size[p]), cnt + (0 if g.is_ref[p] else 1)))
    return best


def test_host_classification_equals_spec():
    rng = np.random.default_rng(7)
    general = 0
    for it in range(80):
        gen = [nested_bubble_graph, random_bubble_graph, overlapping_bubble_graph, deep_nested_graph][it % 4]
        seqs, edges, lin, af = gen(rng)
        g = GraphArrays.from_dicts(seqs, edges, lin, af)
        for k in (3, 6):
            for M in (0, 3):
                follow = None
                if it % 3 == 0:
                    cand = [n for n in seqs if not g.is_ref[n]]
                    follow = set(int(x) for x in rng.choice(cand, size=max(1, len(cand) // 3), replace=False))
                words, gen_needed = classify_nodes(g, k, M, follow)
                flags, bound = (words & 0xFF).astype(np.uint8), words >> 8
                assert np.array_equal(flags, np.array(spec_general.classify(g, k, M, follow), dtype=np.uint8))
                # the history bound is an upper bound on the variant nodes of the k bases before a node (brute force)
                if it % 5 == 0 and M == 3:
                    for n in range(g.n_nodes):
                        if not (flags[n] & spec_general.DEAD):
                            assert bound[n] >= _max_variant_nodes_before(g, flags, n, k)
                interesting = spec_general.NESTED | spec_general.CHECK | spec_general.HFS | spec_general.FORCED
                if np.any(flags & interesting):
                    assert gen_needed
                if gen is random_bubble_graph and follow is None:
                    assert not gen_needed            # SNP / indel bubbles: the simple class, also at limit 0
                general += gen_needed
    assert general > 50


def test_simple_graphs_stay_on_the_fast_path():
    from graph_kmer_index_amd.graph import synthetic_snp_graph, synthetic_indel_graph
    for g in (synthetic_snp_graph(50000, 500, k=31, seed=1), synthetic_indel_graph(50000, 500, k=31, seed=2)):
        flags, general = classify_nodes(g, 31, 5)
        assert not general
        assert np.all(((flags & spec_general.T) != 0) == (g.is_ref != 0))
        assert np.all((flags >> 8) <= 5)


def test_nested_generator_at_scale_is_wellformed_and_matches_the_oracle():
    from graph_kmer_index_amd.graph import synthetic_nested_graph
    g = synthetic_nested_graph(200000, 2000, k=31, seed=9, p_nest=0.3)
    src = np.repeat(np.arange(g.n_nodes), np.diff(g.edge_start))
    assert np.all(g.edges > src)                                          # topological ids
    assert int(g.node_size[g.is_ref == 1].sum()) == 200000
    flags, general = classify_nodes(g, 31, 8)
    assert general and np.any(flags & spec_general.NESTED) and np.any(flags & spec_general.CHECK)
    assert not np.any(flags & spec_general.DEAD)
    small = synthetic_nested_graph(4000, 80, k=9, seed=4, p_nest=0.5)
    got, critd = oracle_rows(small, 9, 8, True)
    assert got != "assert" and spec(small, 9, 8, True, critd) == got
    assert oracle_rows(small, 9, 1, True)[0] == "assert"                  # Z1's successors are both non-linear
