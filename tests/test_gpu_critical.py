"""CriticalGraphPaths.from_graph on the device (csrc/gki_critical.hip: jump tables + prefix sums over the walk's path)
against the reference's known answers, the reference-generated fixtures, the host walk of the library and the oracle
(critical_graph_paths.py:42-104), including the graphs at which the reference raises."""
import json
import os
import numpy as np
import pytest

from golden_cases import CRITICAL_KATS
from graph_kmer_index_amd import CriticalGraphPaths
from graph_kmer_index_amd.graph import GraphArrays, synthetic_snp_graph, synthetic_indel_graph, synthetic_nested_graph
from graphgen import random_bubble_graph, nested_bubble_graph, deep_nested_graph, overlapping_bubble_graph
from oracle import oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def both(g, k):
    """(device result, host result), each (nodes, offsets) or the exception text."""
    out = []
    for dev in (True, False):
        try:
            cp = CriticalGraphPaths.from_graph(g, k, on_device=dev)
            out.append((cp.nodes.tolist(), cp.offsets.tolist()))
        except Exception as e:          # noqa: BLE001 -- the reference raises a bare Exception / OverflowError here
            out.append("raises: " + ("offset -1" if "offset -1" in str(e) else "walk"))
    return out


def test_reference_known_answers():
    for name, ((seqs, edges, lin), k, nodes, offsets) in CRITICAL_KATS.items():     # tests/test_critical_graph_paths.py:6-94
        g = GraphArrays.from_dicts(seqs, edges, lin)
        cp = CriticalGraphPaths.from_graph(g, k, on_device=True)
        assert cp.nodes.tolist() == nodes and cp.offsets.tolist() == offsets, name
        assert cp.nodes.dtype == np.uint32 and cp.offsets.dtype == np.uint16


@pytest.mark.parametrize("fixture", ["finder_toy.json", "finder_two_chrom.json"])
def test_reference_generated_fixtures(fixture):
    with open(os.path.join(GOLD, fixture)) as f:
        cases = json.load(f)
    seen = raised = 0
    for case in cases:
        if ("crit_nodes" not in case and case.get("raises") != "E2") or "from_position" in case.get("kw", {}):
            continue                 # (early-stop cases carry no critical points: the reference runs them without)
        seqs = {int(a): b for a, b in case["seqs"].items()}
        edges = {int(a): b for a, b in case["edges"].items()}
        g = GraphArrays.from_dicts(seqs, edges, case["linear"], chromosome_start_nodes=case.get("chromosome_start_nodes"))
        if case.get("raises") == "E2":
            with pytest.raises(Exception, match="offset -1"):
                CriticalGraphPaths.from_graph(g, case["k"], on_device=True)
            raised += 1
            continue
        cp = CriticalGraphPaths.from_graph(g, case["k"], on_device=True)
        assert cp.nodes.tolist() == case["crit_nodes"] and cp.offsets.tolist() == case["crit_offsets"], case["name"]
        seen += 1
    assert seen > 50


def test_random_graphs_equal_the_host_walk_and_the_oracle():
    rng = np.random.default_rng(12)
    n_ok = n_raise = 0
    for it in range(400):
        kind = it % 4
        k = int(rng.integers(2, 12))
        if kind == 0:
            seqs, edges, lin, af = random_bubble_graph(rng, n_var=int(rng.integers(1, 9)), min_ref=1, max_ref=int(rng.integers(2, 3 * k)),
                                                       p_indel=0.4, chain_after={int(rng.integers(-1, 3)): int(rng.integers(1, k + 2))})
        elif kind == 1:
            seqs, edges, lin, af = nested_bubble_graph(rng, n_var=int(rng.integers(2, 7)), min_ref=1, max_ref=12, p_nest=0.6, p_chain=0.4)
        elif kind == 2:
            seqs, edges, lin, af = deep_nested_graph(rng, n_var=int(rng.integers(1, 5)), max_depth=2)
        else:
            seqs, edges, lin = overlapping_bubble_graph(rng)[:3]
        g = GraphArrays.from_dicts(seqs, edges, lin)
        dev, host = both(g, k)
        assert dev == host, (it, k, dev, host)
        try:
            cn, co = oracle.critical_paths(g, k)
            assert dev == (cn.tolist(), co.tolist())
            n_ok += 1
        except oracle.OracleError:
            assert isinstance(dev, str)
            n_raise += 1
    assert n_ok > 250 and n_raise > 3


def test_branching_node_without_one_linear_successor_raises():
    # node 0 branches into two nodes none of which is on the linear reference: critical_graph_paths.py:96-100
    g = GraphArrays.from_dicts({0: "ACGT", 1: "A", 2: "C", 3: "GGGG"}, {0: [1, 2], 1: [3], 2: [3]}, [0, 3])
    dev, host = both(g, 3)
    assert dev == host and isinstance(dev, str) and "raises" in dev
    with pytest.raises(Exception, match="exactly one linear-ref successor"):
        CriticalGraphPaths.from_graph(g, 3, on_device=True)


@pytest.mark.parametrize("make", [lambda: synthetic_snp_graph(3_000_000, 5000, k=31, seed=5),
                                  lambda: synthetic_indel_graph(2_000_000, 4000, k=31, seed=6, p_del=0.1, p_ins=0.1),
                                  lambda: synthetic_nested_graph(1_000_000, 2000, k=31, seed=7, p_nest=0.2)])
def test_generator_graphs_equal_the_host_walk(make):
    g = make()
    dev, host = both(g, 31)
    assert dev == host and len(dev[0]) > 1000
    cn, co = oracle.critical_paths(g, 31)
    assert dev == (cn.tolist(), co.tolist())
