"""-m gpu: DenseKmerFinder on MI355X (through the C ABI) against the golden vectors generated from the
reference and against the oracle on seeded inputs."""
import json
import os
import numpy as np
import pytest

from golden_cases import canonical_digest
from gpu_util import finder_cols, assert_same_records
from graph_kmer_index_amd import DenseKmerFinder, GraphArrays, CriticalGraphPaths
from graph_kmer_index_amd.graph import synthetic_indel_graph, synthetic_linear_graph, synthetic_snp_graph
from graphgen import random_bubble_graph, overlapping_bubble_graph
from oracle import oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")

with open(os.path.join(GOLD, "finder_toy.json")) as _f:
    TOY = json.load(_f)


def graph_of(case):
    seqs = {int(a): b for a, b in case["seqs"].items()}
    edges = {int(a): b for a, b in case["edges"].items()}
    af = None if case["af"] is None else {int(a): b for a, b in case["af"].items()}
    return GraphArrays.from_dicts(seqs, edges, case["linear"], af)


def is_linear(g):
    return np.all(np.diff(g.edge_start) <= 1)


@pytest.mark.parametrize("case", [c for c in TOY if "from_position" not in c["kw"]], ids=lambda c: c["name"])
def test_toy_graphs_match_reference_records(case):
    g = graph_of(case)
    kw = dict(case["kw"])
    if "only_store_nodes" in kw:
        kw["only_store_nodes"] = set(kw["only_store_nodes"])
    if case.get("raises") == "E2":
        with pytest.raises(Exception):
            CriticalGraphPaths.from_graph(g, case["k"])
        return
    cp = CriticalGraphPaths.from_graph(g, case["k"])
    assert cp.nodes.tolist() == case["crit_nodes"] and cp.offsets.tolist() == case["crit_offsets"]
    # graphs where the reference itself emits meaningless hashes are refused, not imitated
    _, flags = oracle.find(g, case["k"], (cp.nodes, cp.offsets), return_flags=True)
    f = DenseKmerFinder(g, case["k"], critical_graph_paths=cp, **kw)
    if flags & oracle.ORC_FLAG_UNDEFINED_BULK:
        with pytest.raises(ValueError):
            f.find()
        return
    f.find()
    exp = dict(kmers=np.array(case["kmers"], np.int64), nodes=np.array(case["nodes"], np.int32),
               start_nodes=np.array(case["start_nodes"], np.int32),
               start_offsets=np.array(case["start_offsets"], np.int16),
               allele_frequencies=np.array(case["allele_frequencies"], np.float64))
    got = finder_cols(f)
    assert got["kmers"].dtype == np.int64 and got["nodes"].dtype == np.int32
    assert got["start_nodes"].dtype == np.int32 and got["start_offsets"].dtype == np.int16
    assert got["allele_frequencies"].dtype == np.float64
    assert_same_records(got, exp, exact_order=is_linear(g))
    k2, n2 = f.get_found_kmers_and_nodes()
    assert np.array_equal(k2, got["kmers"]) and np.array_equal(n2, got["nodes"])


def test_medium_linear_exact_order():
    med = np.load(os.path.join(GOLD, "finder_medium.npz"))
    g = synthetic_linear_graph(20000, node_len=3000, seed=1234)
    for one in (False, True):
        f = DenseKmerFinder(g, 31, only_save_one_node_per_kmer=one)
        f.find()
        got = finder_cols(f)
        tag = "linear20k_one%d" % one
        exp = dict(kmers=med[tag + "_kmers"], nodes=med[tag + "_nodes"], start_nodes=med[tag + "_start_nodes"],
                   start_offsets=med[tag + "_start_offsets"], allele_frequencies=med[tag + "_af"])
        assert_same_records(got, exp, exact_order=True)


def test_medium_snp_graph_digests():
    with open(os.path.join(GOLD, "finder_medium_meta.json")) as fh:
        meta = json.load(fh)
    for name, m in meta.items():
        g = synthetic_snp_graph(m["G"], m["S"], k=m["k"], seed=m["seed"])
        f = DenseKmerFinder(g, m["k"], only_save_one_node_per_kmer=m["one"], max_variant_nodes=m["M"])
        f.find()
        got = finder_cols(f)
        assert len(got["kmers"]) == m["n_records"], name
        assert canonical_digest(got) == m["digest"], name


@pytest.mark.parametrize("mode,n,seed", [("bubble", 60, 21), ("overlap", 40, 22), ("chain", 80, 23),
                                         ("bubble_bigk", 60, 24), ("overlap_bigk", 30, 25), ("chain_bigk", 70, 26)])
def test_random_graphs_against_oracle(mode, n, seed):
    rng = np.random.default_rng(seed)
    checked = 0
    bigk = mode.endswith("_bigk")
    mode = mode.split("_")[0]
    for _ in range(n):
        k = int(rng.integers(12, 32)) if bigk else int(rng.integers(3, 12))
        M = int(rng.choice([0, 1, 2, 3, 4, 100]))
        one = bool(rng.integers(0, 2))
        if mode == "bubble":
            seqs, edges, lin, af = random_bubble_graph(rng, n_var=int(rng.integers(2, 30)), min_ref=1, max_ref=2 * k,
                                                       p_indel=float(rng.choice([0.0, 0.5])), with_af=True)
        elif mode == "overlap":
            seqs, edges, lin, af = overlapping_bubble_graph(rng, n_var=int(rng.integers(3, 12)))
        else:
            nv = int(rng.integers(1, 6))
            seqs, edges, lin, af = random_bubble_graph(
                rng, n_var=nv, min_ref=1, max_ref=3 * k + 8, p_indel=0.3,
                chain_after={int(rng.integers(-1, nv)): int(rng.integers(1, k + 2))})
        g = GraphArrays.from_dicts(seqs, edges, lin, af)
        try:
            cn, co = oracle.critical_paths(g, k)
        except oracle.OracleError:
            with pytest.raises(Exception):
                CriticalGraphPaths.from_graph(g, k)
            continue
        exp, flags = oracle.find(g, k, (cn, co), one, M, return_flags=True)
        f = DenseKmerFinder(g, k, only_save_one_node_per_kmer=one, max_variant_nodes=M)
        if flags & oracle.ORC_FLAG_UNDEFINED_BULK:
            with pytest.raises(ValueError):
                f.find()
            continue
        f.find()
        assert_same_records(finder_cols(f), exp)
        checked += 1
    assert checked > n // 2


@pytest.mark.parametrize("G,S,M,one", [(400000, 3000, 5, True), (200000, 4000, 3, False), (150000, 6000, 1, True)])
def test_snp_graph_k31_against_oracle(G, S, M, one):
    g = synthetic_snp_graph(G, S, k=31, seed=5)
    exp = oracle.find(g, 31, None, one, M)
    f = DenseKmerFinder(g, 31, only_save_one_node_per_kmer=one, max_variant_nodes=M)
    f.find()
    assert_same_records(finder_cols(f), exp)


@pytest.mark.parametrize("G,S,M,one,p", [(300000, 3500, 5, True, 0.2), (150000, 4000, 2, False, 0.35), (120000, 5000, 0, True, 0.5)])
def test_indel_graph_k31_against_oracle(G, S, M, one, p):
    # SNP / 1-bp deletion / 1-bp insertion sites: empty alt nodes and empty linear-ref dummy nodes at k=31
    g = synthetic_indel_graph(G, S, k=31, seed=G % 97, p_del=p / 2, p_ins=p / 2)
    assert (g.node_size == 0).sum() > S * p * 0.8
    f = DenseKmerFinder(g, 31, only_save_one_node_per_kmer=one, max_variant_nodes=M)
    f.find()
    assert_same_records(finder_cols(f), oracle.find(g, 31, None, one, M))


def test_flat_layout_on_device_matches_v2_columns():
    g = synthetic_snp_graph(120000, 1500, k=31, seed=9)
    f = DenseKmerFinder(g, 31, only_save_one_node_per_kmer=True, max_variant_nodes=5)
    f.find()
    v2 = finder_cols(f)
    pos = g.position_id_base()[v2["start_nodes"]] + v2["start_offsets"]
    fl1 = f.get_flat_kmers(v="1")
    assert np.array_equal(np.asarray(fl1._ref_offsets), pos)
    want = (v2["kmers"].astype(np.uint64), v2["nodes"].astype(np.uint32), pos.astype(np.uint64),
            v2["allele_frequencies"].astype(np.float32))
    for split in (False, True):
        d = f.find_flat_on_device(split_layout=split)
        f.synchronize()
        flat = d.to_flat_kmers()
        assert flat._hashes.dtype == np.uint64 and flat._nodes.dtype == np.uint32
        assert flat._ref_offsets.dtype == np.uint64 and flat._allele_frequencies.dtype == np.float32
        got = (flat._hashes, flat._nodes, flat._ref_offsets, flat._allele_frequencies)
        if split:      # same records; interior ones (by position) first, then the boundary ones
            og = np.lexsort((got[3], got[1], got[0], got[2]))
            ow = np.lexsort((want[3], want[1], want[0], want[2]))
            for a_, b_ in zip(got, want):
                assert np.array_equal(a_[og], b_[ow])
            n_int = f.interior_records()
            assert np.all(np.diff(flat._ref_offsets[:n_int].astype(np.int64)) > 0)
        else:          # by-node layout: identical order to find()
            for a_, b_ in zip(got, want):
                assert np.array_equal(a_, b_)


def test_chunked_find_partitions_the_records():
    # command_line_interface.py:588-601: contiguous ranges of critical-path numbers
    g = synthetic_snp_graph(300000, 2500, k=31, seed=13)
    cp = CriticalGraphPaths.from_graph(g, 31)
    full = DenseKmerFinder(g, 31, critical_graph_paths=cp, only_save_one_node_per_kmer=True, max_variant_nodes=5)
    full.find()
    exp = finder_cols(full)
    n = len(cp)
    cuts = [0] + sorted(np.random.default_rng(1).integers(1, n, size=6).tolist()) + [n]
    parts = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        f = DenseKmerFinder(g, 31, critical_graph_paths=cp, only_save_one_node_per_kmer=True, max_variant_nodes=5,
                            start_at_critical_path_number=a, stop_at_critical_path_number=b)
        f.find()
        parts.append(finder_cols(f))
        # each chunk equals the oracle's chunk (which equals the reference's, tests/test_oracle_vs_reference.py)
        o = oracle.find(g, 31, (cp.nodes, cp.offsets), True, 5, start_at_critical_path_number=a,
                        stop_at_critical_path_number=b)
        assert_same_records(parts[-1], o)
    cat = {k: np.concatenate([p[k] for p in parts]) for k in exp}
    assert_same_records(cat, exp)
    # records are ordered by end node: chunks concatenate to the full run up to the order inside the cut nodes
    assert np.array_equal(np.sort(cat["start_nodes"], kind="stable"), cat["start_nodes"])


def test_chunked_find_on_graphs_whose_node_ids_are_not_topological():
    # overlapping alleles get the largest node id but sit in the middle of the graph: the run of a chunk is then a set of
    # topological ranks, not an id range (gki_find_params.h_node_rank)
    rng = np.random.default_rng(77)
    checked = 0
    for it in range(60):
        k = int(rng.integers(3, 10))
        seqs, edges, lin, af = overlapping_bubble_graph(rng, n_var=int(rng.integers(4, 12)), min_ref=2, max_ref=2 * k)
        g = GraphArrays.from_dicts(seqs, edges, lin, af)
        src = np.repeat(np.arange(g.n_nodes), np.diff(g.edge_start))
        if not np.any(g.edges <= src):
            continue
        try:
            crit = oracle.critical_paths(g, k)
        except oracle.OracleError:
            continue
        n = len(crit[0])
        one, M = bool(rng.integers(0, 2)), int(rng.choice([1, 2, 4]))
        cp = CriticalGraphPaths(crit[0], crit[1])
        for a, b in [(0, n // 2), (n // 2, n), (1, max(1, n - 1)), (0, 0), (n, n)]:
            exp, flags = oracle.find(g, k, crit, one, M, start_at_critical_path_number=a, stop_at_critical_path_number=b,
                                     return_flags=True)
            if flags:
                continue
            f = DenseKmerFinder(g, k, critical_graph_paths=cp, only_save_one_node_per_kmer=one, max_variant_nodes=M,
                                start_at_critical_path_number=a, stop_at_critical_path_number=b)
            f.find()
            assert_same_records(finder_cols(f), exp)
            checked += 1
    assert checked > 50


def test_golden_chunk_cases():
    for case in TOY:
        if not case["name"].startswith(("rand_chunk_", "chunk_offset0_")):
            continue
        g = graph_of(case)
        f = DenseKmerFinder(g, case["k"], **case["kw"])
        f.find()
        exp = dict(kmers=np.array(case["kmers"], np.int64), nodes=np.array(case["nodes"], np.int32),
                   start_nodes=np.array(case["start_nodes"], np.int32),
                   start_offsets=np.array(case["start_offsets"], np.int16),
                   allele_frequencies=np.array(case["allele_frequencies"], np.float64))
        assert_same_records(finder_cols(f), exp)


def test_formerly_unsupported_graphs_match_the_oracle():
    # a variant node whose only predecessor is a variant node (a two-node alternative allele)
    g = GraphArrays.from_dicts({0: "ACGTACGT", 1: "A", 2: "C", 3: "G", 4: "TTTTTTTT"},
                               {0: [1, 2], 2: [3], 1: [4], 3: [4]}, [0, 1, 4])
    for M in (1, 2, 4):
        f = DenseKmerFinder(g, 4, max_variant_nodes=M)
        if M == 1:          # the reference asserts at node 2: window at the limit, its only successor is not linear-ref
            with pytest.raises(AssertionError):
                f.find()
            with pytest.raises(oracle.OracleError):
                oracle.find(g, 4, max_variant_nodes=M)
            continue
        f.find()
        assert_same_records(finder_cols(f), oracle.find(g, 4, max_variant_nodes=M))
    g2 = GraphArrays.from_dicts({0: "ACGTACGT", 1: "A", 2: "C", 3: "TTTTTTTT"}, {0: [1, 2], 1: [3], 2: [3]}, [0, 1, 3])
    f = DenseKmerFinder(g2, 4, only_follow_nodes={2})
    f.find()
    assert_same_records(finder_cols(f), oracle.find(g2, 4, only_follow_nodes={2}))
    assert 1 not in set(finder_cols(f)["nodes"].tolist())        # the forced allele hides its sibling (:386-388)


def test_whitelist_and_only_store_nodes_filters():
    g = synthetic_snp_graph(50000, 600, k=31, seed=3)
    base = oracle.find(g, 31, None, False, 4)
    wl = set(int(x) for x in base["kmers"][::7])
    f = DenseKmerFinder(g, 31, whitelist=wl)
    f.find()
    assert_same_records(finder_cols(f), oracle.find(g, 31, None, False, 4, whitelist=wl))
    variant_nodes = set(np.nonzero(g.is_ref == 0)[0].tolist())
    f = DenseKmerFinder(g, 31, only_store_nodes=variant_nodes)
    f.find()
    assert_same_records(finder_cols(f), oracle.find(g, 31, None, False, 4, only_store_nodes=variant_nodes))


def test_whitelist_on_device_as_set_index_and_flat_columns():
    # kmer_finder.py:130-132, 362-365 with the whitelist a CollisionFreeKmerIndex, as the reference's CLI passes it
    # (command_line_interface.py:634): membership probe + stable compaction in HBM
    from graph_kmer_index_amd import CollisionFreeKmerIndex, FlatKmers
    g = synthetic_snp_graph(90000, 1100, k=31, seed=31)
    base = oracle.find(g, 31, None, True, 5)
    rng = np.random.default_rng(3)
    chosen = np.unique(base["kmers"][rng.random(len(base["kmers"])) < 0.3])
    foreign = rng.integers(0, 4 ** 31, size=5000, dtype=np.int64)
    wl_kmers = np.concatenate([chosen, foreign])
    wl_index = CollisionFreeKmerIndex.from_flat_kmers(
        FlatKmers(wl_kmers, np.zeros(len(wl_kmers), np.uint32), np.zeros(len(wl_kmers), np.uint64),
                  np.ones(len(wl_kmers), np.float32)), modulo=20011)
    want = oracle.find(g, 31, None, True, 5, whitelist=set(int(x) for x in wl_kmers))
    assert 0 < len(want["kmers"]) < len(base["kmers"])
    for wl in (wl_index, set(int(x) for x in wl_kmers)):
        f = DenseKmerFinder(g, 31, only_save_one_node_per_kmer=True, max_variant_nodes=5, whitelist=wl)
        f.find()
        assert_same_records(finder_cols(f), want)
        pos = g.position_id_base()[want["start_nodes"]] + want["start_offsets"]
        for split in (False, True):
            d = f.find_flat_on_device(split_layout=split)
            flat = d.to_flat_kmers()
            assert d.n == len(want["kmers"])
            og = np.lexsort((flat._allele_frequencies, flat._nodes, flat._hashes, flat._ref_offsets))
            ow = np.lexsort((want["allele_frequencies"].astype(np.float32), want["nodes"], want["kmers"], pos))
            assert np.array_equal(flat._hashes[og], want["kmers"].astype(np.uint64)[ow])
            assert np.array_equal(flat._nodes[og], want["nodes"].astype(np.uint32)[ow])
            assert np.array_equal(flat._ref_offsets[og], pos.astype(np.uint64)[ow])
            assert np.array_equal(flat._allele_frequencies[og], want["allele_frequencies"].astype(np.float32)[ow])
            if not split:          # compaction is stable: by-node order of find() survives
                got_v2 = finder_cols(f)
                assert np.array_equal(flat._hashes, got_v2["kmers"].astype(np.uint64))
    # nothing whitelisted -> empty columns
    f = DenseKmerFinder(g, 31, only_save_one_node_per_kmer=True, whitelist={1})
    assert f.find_flat_on_device().n == 0
    f.find()
    assert len(finder_cols(f)["kmers"]) == 0



@pytest.mark.gpu
@pytest.mark.parametrize("k,M,seed", [(16, 100, 41), (24, 3, 42), (31, 100, 43), (31, 5, 44)])
def test_windows_over_more_nodes_than_the_step_queue_carries(k, M, seed):
    # bubbles 1-3 bp apart: with only_save_one_node_per_kmer=False most windows span far more than the 6 nodes a queued
    # step carries (NLQ in csrc/gki_finder.hip), so their records come from the kernel's many-node path -- in the v2
    # layout (find), both flat layouts on the device, and under the only_store_nodes filter
    rng = np.random.default_rng(seed)
    seqs, edges, lin, af = random_bubble_graph(rng, n_var=12, min_ref=1, max_ref=3, p_indel=0.4, with_af=True,
                                               first_ref=k + 3, last_ref=k + 5)
    g = GraphArrays.from_dicts(seqs, edges, lin, af)
    exp = oracle.find(g, k, None, False, M)
    per_window = len(exp["kmers"]) / max(1, len(set(zip(exp["kmers"].tolist(), exp["start_nodes"].tolist(),
                                                           exp["start_offsets"].tolist()))))
    assert per_window > 6.5                      # the case is what it claims to be
    f = DenseKmerFinder(g, k, only_save_one_node_per_kmer=False, max_variant_nodes=M)
    f.find()
    v2 = finder_cols(f)
    assert_same_records(v2, exp)
    pos = g.position_id_base()[v2["start_nodes"]] + v2["start_offsets"]
    want = (v2["kmers"].astype(np.uint64), v2["nodes"].astype(np.uint32), pos.astype(np.uint64),
            v2["allele_frequencies"].astype(np.float32))
    for split in (False, True):
        d = f.find_flat_on_device(split_layout=split)
        f.synchronize()
        flat = d.to_flat_kmers()
        got = (flat._hashes, flat._nodes, flat._ref_offsets, flat._allele_frequencies)
        if split:
            og = np.lexsort((got[3], got[1], got[0], got[2]))
            ow = np.lexsort((want[3], want[1], want[0], want[2]))
            for a_, b_ in zip(got, want):
                assert np.array_equal(a_[og], b_[ow])
        else:
            for a_, b_ in zip(got, want):
                assert np.array_equal(a_, b_)
    keep = set(int(x) for x in range(g.n_nodes) if x % 3 != 1)
    f = DenseKmerFinder(g, k, only_save_one_node_per_kmer=False, max_variant_nodes=M, only_store_nodes=keep)
    f.find()
    assert_same_records(finder_cols(f), oracle.find(g, k, None, False, M, only_store_nodes=keep))

@pytest.mark.parametrize("case", [c for c in TOY if "from_position" in c["kw"]], ids=lambda c: c["name"])
def test_kmers_from_position_reference_cases(case):
    # tests/test_kmer_finder.py:118-129, 300-382 of the reference (early-stop searches), exact order
    g = graph_of(case)
    kw = dict(case["kw"])
    pos = kw.pop("from_position")
    if "only_store_nodes" in kw:
        kw["only_store_nodes"] = set(kw["only_store_nodes"])
    f = DenseKmerFinder(g, case["k"], **kw)
    f.find_only_kmers_starting_at_position(*pos)
    got = finder_cols(f)
    exp = dict(kmers=np.array(case["kmers"], np.int64), nodes=np.array(case["nodes"], np.int32),
               start_nodes=np.array(case["start_nodes"], np.int32),
               start_offsets=np.array(case["start_offsets"], np.int16),
               allele_frequencies=np.array(case["allele_frequencies"], np.float64))
    assert_same_records(got, exp, exact_order=True)


def test_kmers_from_positions_random_vs_oracle():
    rng = np.random.default_rng(31)
    for it in range(40):
        k = int(rng.integers(3, 14))
        M = int(rng.choice([0, 1, 2, 4, 100]))
        one = bool(rng.integers(0, 2))
        seqs, edges, lin, af = random_bubble_graph(rng, n_var=int(rng.integers(2, 12)), min_ref=1, max_ref=2 * k,
                                                   p_indel=0.5, with_af=True)
        g = GraphArrays.from_dicts(seqs, edges, lin, af)
        nodes = rng.integers(0, g.n_nodes, size=12)
        offs = [int(rng.integers(0, max(1, g.node_size[n]))) for n in nodes]
        f = DenseKmerFinder(g, k, only_save_one_node_per_kmer=one, max_variant_nodes=M)
        f.find_kmers_starting_at_positions(nodes, offs)
        got = finder_cols(f)
        exp = [oracle.find_from_position(g, k, int(n), int(o), one, M) for n, o in zip(nodes, offs)]
        exp = {key: np.concatenate([e[key] for e in exp]) for key in exp[0]}
        assert_same_records(got, exp, exact_order=True)


def test_kmers_from_positions_when_node_ids_do_not_grow_along_the_path():
    # an allele that bridges a bubble is numbered after the nodes it skips (overlapping_bubble_graph): a forward path through
    # it visits node ids out of order, and the records of a k-mer still come per distinct node in ascending order
    # (np.unique, kmer_finder.py:134) -- the kernel's selection path, not its ascending-path shortcut
    rng = np.random.default_rng(33)
    out_of_order = 0
    for it in range(40):
        k = int(rng.integers(4, 14))
        M = int(rng.choice([2, 4, 100]))
        seqs, edges, lin, af = overlapping_bubble_graph(rng, n_var=int(rng.integers(3, 10)), min_ref=1, max_ref=k)
        g = GraphArrays.from_dicts(seqs, edges, lin, af)
        nodes = rng.integers(0, g.n_nodes, size=16)
        offs = [int(rng.integers(0, max(1, g.node_size[n]))) for n in nodes]
        f = DenseKmerFinder(g, k, only_save_one_node_per_kmer=False, max_variant_nodes=M)
        try:
            exp = [oracle.find_from_position(g, k, int(n), int(o), False, M) for n, o in zip(nodes, offs)]
        except oracle.OracleError:
            continue
        f.find_kmers_starting_at_positions(nodes, offs)
        got = finder_cols(f)
        exp = {key: np.concatenate([e[key] for e in exp]) for key in exp[0]}
        assert_same_records(got, exp, exact_order=True)
        # did a k-mer's window really contain the bridging allele (the highest id) before a lower one?
        bridge = g.n_nodes - 1
        out_of_order += int(np.any((exp["nodes"] == bridge) & (exp["start_nodes"] < bridge)))
    assert out_of_order >= 5


def test_early_stop_search_after_the_sequence_on_the_device_was_replaced():
    # gki_graph_prepare (include/gki.h) after the caller rewrote its device sequence: the search's per-node records (first
    # bases of every node, built by the first search on the graph) belong to the old sequence and have to be built again --
    # and so do the finder's walk records, whose tails the search reads for its start nodes
    from graph_kmer_index_amd import _lib
    from graph_kmer_index_amd.device_graph import DeviceGraph
    rng = np.random.default_rng(77)
    for it in range(6):
        k = int(rng.integers(4, 14))
        seqs, edges, lin, af = overlapping_bubble_graph(rng, n_var=int(rng.integers(3, 10)), min_ref=1, max_ref=k)
        g = GraphArrays.from_dicts(seqs, edges, lin, af)
        d_seq = _lib.DeviceArray.from_host(np.ascontiguousarray(g.seq, dtype=np.uint8))
        g._device = DeviceGraph(g, d_seq=d_seq)
        nodes = rng.integers(0, g.n_nodes, size=24)
        offs = [int(rng.integers(0, max(1, g.node_size[n]))) for n in nodes]
        for round_ in range(2):
            try:
                exp = [oracle.find_from_position(g, k, int(n), int(o), False, 4) for n, o in zip(nodes, offs)]
            except oracle.OracleError:
                break
            f = DenseKmerFinder(g, k, only_save_one_node_per_kmer=False, max_variant_nodes=4)
            f.find_kmers_starting_at_positions(nodes, offs)
            assert_same_records(finder_cols(f), {key: np.concatenate([e[key] for e in exp]) for key in exp[0]}, exact_order=True)
            f.close()
            # other bases in the same nodes, written over the caller's device buffer
            g.seq[:] = rng.integers(0, 4, size=len(g.seq)).astype(g.seq.dtype)
            _lib.check(_lib.load().gki_memcpy_h2d(d_seq.ptr, _lib.hptr(np.ascontiguousarray(g.seq, dtype=np.uint8)), len(g.seq)))
            g._device.prepare()
        g._device.close()
        g._device = None


def test_early_stop_emit_from_the_script_equals_the_walking_emit():
    # csrc/gki_forward.hip: in all-nodes mode gki_forward_count leaves the finished k-mers in a script and the
    # gki_forward_emit call with the same arguments expands it (start positions with more than four finished k-mers or a
    # path over more than five nodes are walked); the script is released by that call, so a SECOND emit call with the same
    # arguments walks every start position.  Both forms must fill the five columns identically, and as the oracle does.
    import ctypes as C
    from graph_kmer_index_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(35)
    compared = many_paths = 0
    dt = [np.int64, np.int32, np.int16, np.int32, np.float64]
    for it in range(30):
        k = int(rng.integers(3, 14))
        M = int(rng.choice([0, 1, 2, 4, 100]))
        seqs, edges, lin, af = random_bubble_graph(rng, n_var=int(rng.integers(2, 12)), min_ref=1, max_ref=2 * k,
                                                   p_indel=0.5, with_af=True)
        g = GraphArrays.from_dicts(seqs, edges, lin, af)
        f = DenseKmerFinder(g, k, only_save_one_node_per_kmer=False, max_variant_nodes=M)
        nodes = rng.integers(0, g.n_nodes, size=48).astype(np.int32)
        offs = np.array([int(rng.integers(0, max(1, g.node_size[n]))) for n in nodes], dtype=np.int32)
        d_nodes, d_offs = _lib.DeviceArray.from_host(nodes), _lib.DeviceArray.from_host(offs)
        d_start = _lib.DeviceArray(len(nodes) + 1, np.int64)
        n = C.c_int64(0)
        args = (f._device_graph().handle, k, M, 0, None, d_nodes.ptr, d_offs.ptr, len(nodes))
        _lib.check(lib.gki_forward_count(*args, d_start.ptr, C.byref(n)))
        if n.value == 0:
            continue
        first = [_lib.DeviceArray(n.value, d) for d in dt]
        second = [_lib.DeviceArray(n.value, d) for d in dt]
        _lib.check(lib.gki_forward_emit(*args, d_start.ptr, *[b.ptr for b in first]))      # expands the script
        _lib.check(lib.gki_forward_emit(*args, d_start.ptr, *[b.ptr for b in second]))     # no script left: walks
        a, b = [x.to_host() for x in first], [x.to_host() for x in second]
        for col_a, col_b in zip(a, b):
            assert np.array_equal(col_a, col_b)
        exp = [oracle.find_from_position(g, k, int(p), int(o), False, M) for p, o in zip(nodes, offs)]
        exp = {key: np.concatenate([e[key] for e in exp]) for key in exp[0]}
        got = dict(kmers=a[0], start_nodes=a[1], start_offsets=a[2], nodes=a[3], allele_frequencies=a[4])
        assert_same_records(got, exp, exact_order=True)
        per_start = np.diff(d_start.to_host())
        many_paths += int(np.sum(per_start > 20))          # more records than four finished k-mers of five nodes can hold
        compared += 1
        for x in first + second + [d_nodes, d_offs, d_start]:
            x.free()
    assert compared >= 20 and many_paths >= 1


def test_only_follow_nodes_from_position():
    # unique_variant_kmers.py:91-96: only_store_nodes = only_follow_nodes = {variant node}, early-stop search
    rng = np.random.default_rng(41)
    for it in range(30):
        k = int(rng.integers(4, 12))
        seqs, edges, lin, af = random_bubble_graph(rng, n_var=int(rng.integers(2, 8)), min_ref=2, max_ref=2 * k,
                                                   p_indel=0.4, shuffle_succ=False)
        g = GraphArrays.from_dicts(seqs, edges, lin, af)
        variant = np.nonzero((g.is_ref == 0) & (g.exists != 0))[0]
        node = int(rng.choice(variant))
        pred = int(g.rev_edges[g.rev_start[node]])
        off = int(rng.integers(0, max(1, g.node_size[pred])))
        for M in (0, 1, 4):
            f = DenseKmerFinder(g, k, max_variant_nodes=M, only_store_nodes={node}, only_follow_nodes={node})
            f.find_only_kmers_starting_at_position(pred, off)
            exp = oracle.find_from_position(g, k, pred, off, False, M, only_store_nodes={node}, only_follow_nodes={node})
            assert_same_records(finder_cols(f), exp, exact_order=True)


def test_edge_cases_small_and_degenerate_graphs():
    cases = [
        ({0: "ACGT"}, {}, [0], 5),                      # k longer than the graph: nothing
        ({0: "ACGT"}, {}, [0], 4),                      # exactly one k-mer
        ({0: "ACGTACGTAC"}, {}, [0], 1),                # k = 1
        ({0: "AC", 1: "G", 2: "T", 3: "AC"}, {0: [1, 2], 1: [3], 2: [3]}, [0, 1, 3], 2),
        ({3: "ACGTTGCA", 7: "A", 9: "C", 12: "GGTTAACC"}, {3: [7, 9], 7: [12], 9: [12]}, [3, 7, 12], 6),   # sparse ids
        ({0: "ACGTACGTACGTACGTACGTACGTACGTACGTACG", 1: "T", 2: "", 3: "ACGTACGTACGTACGTACGTACGTACGTACGTACGT"},
         {0: [1, 2], 1: [3], 2: [3]}, [0, 1, 3], 31),   # k = 31 across a deletion
    ]
    for seqs, edges, lin, k in cases:
        g = GraphArrays.from_dicts(seqs, edges, lin)
        for one in (False, True):
            for M in (0, 1, 4):
                f = DenseKmerFinder(g, k, only_save_one_node_per_kmer=one, max_variant_nodes=M)
                f.find()
                assert_same_records(finder_cols(f), oracle.find(g, k, None, one, M))
                d = f.find_flat_on_device()
                f.synchronize()
                assert d.n == len(f.get_found_kmers_and_nodes()[0])


def test_many_empty_nodes_in_a_row():
    # chains of deletions: windows cross many empty nodes
    def chain(n_empty):
        seqs, edges, lin = {0: "ACGTACGT"}, {}, [0]
        prev = 0
        for i in range(n_empty):
            a, b = 2 * i + 1, 2 * i + 2          # a: 1-bp ref allele, b: empty alt
            seqs[a], seqs[b] = "C", ""
            edges[prev] = [a, b]
            # join node
            prev_join = 1000 + i
            seqs[prev_join] = "G"
            edges[a] = [prev_join]
            edges[b] = [prev_join]
            lin += [a, prev_join]
            prev = prev_join
        seqs[5000] = "TTTTTTTTTT"
        edges[prev] = [5000]
        lin.append(5000)
        return GraphArrays.from_dicts(seqs, edges, lin)
    g = chain(6)
    f = DenseKmerFinder(g, 8, max_variant_nodes=100)
    f.find()
    assert_same_records(finder_cols(f), oracle.find(g, 8, None, False, 100))
    # 40 sites: windows over more nodes than the product kernels' stacks hold (GKI_MAX_WINDOW_NODES) -- round 2 refused
    # these; since round 3 the slow path answers (tests/test_deep_windows.py has the reference-generated cases and the
    # limit that remains).  The variant limit keeps the number of paths polynomial; without one this graph has 2^30
    # windows per end position, in the reference as here.
    g = chain(40)
    for one in (True, False):
        f = DenseKmerFinder(g, 31, max_variant_nodes=2, only_save_one_node_per_kmer=one)
        f.find()
        assert_same_records(finder_cols(f), oracle.find(g, 31, None, one, 2))


def test_environment_switches_do_not_change_find(monkeypatch):
    # round 1 shipped a diagnostic (GKI_DBG_SKIP_EXPAND) that dropped the boundary records when set
    for name, val in (("GKI_DBG_SKIP_EXPAND", "1"), ("GKI_BND_BLOCKS", "1"), ("GKI_SW", "128"), ("GKI_OVERLAP_EMIT", "1")):
        monkeypatch.setenv(name, val)
    g = synthetic_snp_graph(60000, 700, k=31, seed=77)
    for one in (True, False):
        f = DenseKmerFinder(g, 31, only_save_one_node_per_kmer=one, max_variant_nodes=5)
        f.find()
        assert_same_records(finder_cols(f), oracle.find(g, 31, None, one, 5))
