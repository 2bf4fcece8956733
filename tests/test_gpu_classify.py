"""gki_graph_classify_nodes (relaxation sweeps on the device, csrc/gki_classify.hip) against the host pass in
topological order (gki_classify_nodes): same flag words, same `general` verdict, on every graph family of the suite,
with and without forced successors, for every limit -- including the graphs where a nested non-free node needs its
histories enumerated (rounds of relaxation and k_cls_history on the device; the host pass only beyond the kernel's
stack)."""
import ctypes as C
import numpy as np
import pytest

from graph_kmer_index_amd import _lib, CriticalGraphPaths, DeviceGraph
from graph_kmer_index_amd.graph import GraphArrays, synthetic_snp_graph, synthetic_indel_graph, synthetic_nested_graph
from graph_kmer_index_amd.kmer_finder import classify_nodes, search_roots
from graphgen import random_bubble_graph, nested_bubble_graph, deep_nested_graph, overlapping_bubble_graph
from oracle import oracle

pytestmark = pytest.mark.gpu
GKI_REF, GKI_FORCED, GKI_T, GKI_SIMPLE, GKI_NESTED, GKI_CHECK, GKI_HFS, GKI_DEAD = (1 << i for i in range(8))   # include/gki.h GKI_NODE_*


def device_only(g, k, M, follow, crit):
    """(flags, general, needs_host) of the device call alone."""
    lib = _lib.load()
    f = None
    if follow is not None:
        f = np.zeros(g.n_nodes, dtype=np.uint8)
        f[list(follow)] = 1
    roots = np.ascontiguousarray(np.concatenate([np.asarray(search_roots(g, k), dtype=np.int32), np.asarray(crit, dtype=np.int32)]))
    flags = np.zeros(g.n_nodes, dtype=np.uint16)
    general, needs = C.c_int32(0), C.c_int32(0)
    _lib.check(lib.gki_graph_classify_nodes(DeviceGraph.of(g).handle, _lib.hptr(f), _lib.hptr(roots), len(roots), k, M,
                                            _lib.hptr(flags), 1, C.byref(general), C.byref(needs)))
    return flags, bool(general.value), bool(needs.value)


def test_random_graphs_equal_the_host_pass():
    rng = np.random.default_rng(44)
    n_dev = n_host = n_hist = 0
    for it in range(500):
        kind = it % 5
        k = int(rng.integers(2, 14))
        if kind == 0:
            seqs, edges, lin, af = random_bubble_graph(rng, n_var=int(rng.integers(1, 12)), min_ref=1, max_ref=int(rng.integers(2, 3 * k)),
                                                       p_indel=0.4, chain_after={int(rng.integers(-1, 3)): int(rng.integers(1, k + 2))})
        elif kind == 1:
            seqs, edges, lin, af = nested_bubble_graph(rng, n_var=int(rng.integers(2, 8)), min_ref=1, max_ref=12, p_nest=0.6, p_chain=0.4)
        elif kind == 2:
            seqs, edges, lin, af = deep_nested_graph(rng, n_var=int(rng.integers(1, 5)), max_depth=2)
        elif kind == 3:
            seqs, edges, lin = overlapping_bubble_graph(rng)[:3]
        else:
            seqs, edges, lin, af = random_bubble_graph(rng, n_var=int(rng.integers(2, 9)), min_ref=1, max_ref=2 * k, p_indel=0.5)
        g = GraphArrays.from_dicts(seqs, edges, lin)
        try:
            crit = oracle.critical_paths(g, k)[0]
        except oracle.OracleError:
            continue
        M = int(rng.choice([0, 1, 2, 4, 250]))
        follow = None
        if kind == 4:
            cand = [n for n in seqs if n not in lin]
            follow = set(int(x) for x in rng.choice(cand, size=min(len(cand), int(rng.integers(1, 3))), replace=False)) if cand else None
        want_flags, want_general = classify_nodes(g, k, M, follow, crit, on_device=False)
        flags, general, needs = device_only(g, k, M, follow, crit)
        # graphs with a nested non-free node (its histories are enumerated: round 3 on the device, k_cls_history)
        low = want_flags & 0xFF
        nonfree = (low & (GKI_REF | GKI_FORCED)) == 0
        if M >= 1 and np.any(nonfree & ((low & GKI_NESTED) != 0) | (nonfree & ((low & GKI_DEAD) != 0) & (g.rev_start[1:] > g.rev_start[:-1]))):
            n_hist += 1
        if needs:
            n_host += 1
            # the public entry point hands over and still equals the host pass
            got_flags, got_general = classify_nodes(g, k, M, follow, crit, on_device=True)
            assert got_general == want_general and np.array_equal(got_flags, want_flags), (it, k, M)
            continue
        n_dev += 1
        assert general == want_general, (it, k, M, follow)
        assert np.array_equal(flags, want_flags), (it, k, M, follow, flags.tolist(), want_flags.tolist())
    assert n_dev > 300 and n_host <= 3 and n_hist > 20, (n_dev, n_host, n_hist)


@pytest.mark.parametrize("make,expect_general", [
    (lambda: synthetic_snp_graph(3_000_000, 5000, k=31, seed=5), False),
    (lambda: synthetic_indel_graph(2_000_000, 4000, k=31, seed=6, p_del=0.1, p_ins=0.1), False),
    (lambda: synthetic_nested_graph(1_000_000, 2000, k=31, seed=7, p_nest=0.2), True)])
def test_generator_graphs(make, expect_general):
    g = make()
    cp = CriticalGraphPaths.from_graph(g, 31)
    for M in (5, 8):
        want_flags, want_general = classify_nodes(g, 31, M, None, cp.nodes, on_device=False)
        got_flags, got_general = classify_nodes(g, 31, M, None, cp.nodes, on_device=True)
        assert got_general == want_general == expect_general
        assert np.array_equal(got_flags, want_flags)


def test_a_later_chromosome_with_a_short_start_node_is_dead_until_its_first_critical_point():
    two = GraphArrays.from_dicts({0: "ACGTACGT", 1: "AC", 2: "GGGG"}, {1: [2]}, [0, 1, 2], chromosome_start_nodes=[0, 1])
    flags, general, needs = device_only(two, 4, 4, None, [0, 2])
    assert not needs and general and flags[1] & 128 and not flags[2] & 128 and not flags[0] & 128
