"""CPU-side checks: the C-ABI library loads and exports every symbol include/gki.h declares; host logic of the
drop-in classes (no GPU compute calls)."""
import ctypes as C
import os
import re
import numpy as np
import pytest

from conftest import ROOT
from graph_kmer_index_amd import _lib, GraphArrays, FlatKmers, CriticalGraphPaths
from graph_kmer_index_amd import letter_sequence_to_numeric, sequence_to_kmer_hash, kmer_hash_to_sequence, NpList
from graph_kmer_index_amd.kmer_hashing import power_array, reverse_power_array, kmer_hashes_to_bases, kmer_to_hash_fast
from graph_kmer_index_amd.kmer_finder import search_roots, lossy_table, update_hash, DenseKmerFinder
from graph_kmer_index_amd.graph import synthetic_snp_graph, synthetic_linear_graph
from golden_cases import CRITICAL_KATS, REFERENCE_TEST_GRAPHS
from oracle import oracle


def header_functions():
    text = open(os.path.join(ROOT, "include", "gki.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gki_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(_lib.LIB_PATH)
    names = header_functions()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(_lib.SYMBOLS) == names          # the ctypes binding covers the whole header
    _lib.load()


def test_no_cpu_fallback_without_gpu():
    if _lib.device_count() > 0:
        pytest.skip("GPU present")
    g = GraphArrays.from_dicts({1: "ACTG", 2: "A", 3: "G", 4: "CCCC"}, {1: [2, 3], 2: [4], 3: [4]}, [1, 2, 4])
    with pytest.raises(_lib.GkiError):
        DenseKmerFinder(g, 3).find()
    from graph_kmer_index_amd.kmer_hashing import kmer_hashes_to_reverse_complement_hash
    with pytest.raises(_lib.GkiError):
        kmer_hashes_to_reverse_complement_hash(np.array([5], dtype=np.uint64), 3)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "graph_kmer_index_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in text and "from oracle" not in text and "libgki_oracle" not in text, fn


@pytest.mark.parametrize("name", sorted(CRITICAL_KATS))
def test_critical_paths_known_answers(name):
    # host-side walk inside libgki_hip.so (no GPU needed); tests/test_critical_graph_paths.py of the reference
    (seqs, edges, lin), k, nodes, offsets = CRITICAL_KATS[name]
    cp = CriticalGraphPaths.from_graph(GraphArrays.from_dicts(seqs, edges, lin), k)
    assert cp.nodes.tolist() == nodes and cp.offsets.tolist() == offsets
    assert cp.nodes.dtype == np.uint32 and cp.offsets.dtype == np.uint16
    assert len(cp) == len(nodes) and list(cp) == list(zip(nodes, offsets))
    for n, o in zip(nodes, offsets):
        assert cp.is_critical(n, o) and not cp.is_critical(n, o + 1)


def test_critical_paths_synthetic_vs_oracle():
    for g in (synthetic_snp_graph(50000, 700, k=31, seed=1), synthetic_linear_graph(100000, 7000, seed=2)):
        cp = CriticalGraphPaths.from_graph(g, 31)
        on, oo = oracle.critical_paths(g, 31)
        assert np.array_equal(cp.nodes, on) and np.array_equal(cp.offsets, oo)
    with pytest.raises(Exception):      # reference crash E2: exactly k bases of chain before a node
        CriticalGraphPaths.from_graph(GraphArrays.from_dicts({0: "ACG", 1: "TTTTTT"}, {0: [1]}, [0, 1]), 3)


def test_hash_helpers_known_answers():
    # tests/test_kmer_hashing.py:11,27,30-35,69-75
    assert sequence_to_kmer_hash("ACTG") == 0 * 1 + 1 * 4 + 3 * 16 + 2 * 64
    assert sequence_to_kmer_hash("T" * 31) == 4611686018427387903
    for s in ["atg", "Acacatacgactacg", "CAtgAACAtttggtAATCTACAtgAACAttt", "G"]:
        assert kmer_hash_to_sequence(sequence_to_kmer_hash(s), len(s)) == s.lower()
        assert sequence_to_kmer_hash(s) == int(np.sum(reverse_power_array(len(s)) * letter_sequence_to_numeric(s)))
        assert sequence_to_kmer_hash(s) == oracle.sequence_to_kmer_hash(s)
    hashes = np.array([sequence_to_kmer_hash(s) for s in ["ACTG", "TGGC"]])
    assert kmer_hashes_to_bases(hashes, 4).tolist() == [[0, 1, 3, 2], [3, 2, 2, 1]]
    assert power_array(3).tolist() == [16, 4, 1] and reverse_power_array(3).tolist() == [1, 4, 16]
    assert letter_sequence_to_numeric("acgtnmACGTNM").tolist() == [0, 1, 2, 3, 0, 0, 0, 1, 2, 3, 0, 0]
    assert letter_sequence_to_numeric("acgt").dtype == np.uint64
    assert kmer_to_hash_fast(np.array([1, 2], dtype=np.uint64), 2) == 9
    for k in (3, 31):
        h = 0
        seq = np.random.default_rng(k).integers(0, 4, size=80)
        for i, b in enumerate(seq):
            h = update_hash(b, h, seq[i - k] if i >= k else 0, k, only_add=False if i >= k else i)
            assert h == oracle.update_hash(b, 0 if i == 0 else prev, seq[i - k] if i >= k else 0, k,
                                           only_add=False if i >= k else i)
            prev = h


def test_nplist_semantics():
    # tests/test_nplist.py of the reference
    l = NpList(dtype=np.int32)
    for i in range(250):
        l.append(i)
    assert len(l) == 250 and l[-1] == 249 and l.get_nparray().dtype == np.int32
    l.extend(np.arange(1000))
    assert len(l) == 1250 and l[250] == 0
    c = l.copy()
    assert c == l
    l.set_n_elements(10)
    assert len(l) == 10 and l.get_nparray().tolist() == list(range(10))


def test_flat_kmers_container(tmp_path):
    f1 = FlatKmers(np.array([5, 6, 5], dtype=np.int64), np.array([1, 2, 3], dtype=np.int32), np.array([7, 8, 9]),
                   np.array([0.5, 1.0, 0.25]))
    f2 = FlatKmers(np.array([6], dtype=np.int64), np.array([4], dtype=np.int32), np.array([1]), np.array([1.0]))
    m = FlatKmers.from_multiple_flat_kmers([f1, f2])
    assert m._hashes.dtype == np.uint64 and m._nodes.dtype == np.uint32
    assert m._ref_offsets.dtype == np.uint64 and m._allele_frequencies.dtype == np.float32
    assert m._hashes.tolist() == [5, 6, 5, 6]
    ns = m.get_new_without_singletons()          # 2nd+ occurrences, original order (flat_kmers.py:98-125)
    assert ns._hashes.tolist() == [5, 6] and ns._nodes.tolist() == [3, 4]
    assert np.array_equal(ns._hashes, m._hashes[oracle.without_singletons(m._hashes)])
    path = str(tmp_path / "flat")
    m.to_file(path)
    assert set(np.load(path + ".npz").keys()) == {"hashes", "nodes", "ref_offsets", "allele_frequencies"}
    back = FlatKmers.from_file(path)
    assert np.array_equal(back._hashes, m._hashes) and np.array_equal(back._allele_frequencies, m._allele_frequencies)
    with pytest.raises(AssertionError):
        FlatKmers(np.zeros(2), np.zeros(3))


def test_graph_arrays_accessors_and_adapter():
    seqs, edges, lin, k, kw = REFERENCE_TEST_GRAPHS["multiple_critical_points"]
    g = GraphArrays.from_dicts(seqs, edges, lin)
    assert g.get_node_size(4) == 3 and g.get_edges(4) == [5, 6] and g.get_first_node() == 1
    assert g.get_numeric_node_sequence(4).tolist() == [0, 1, 3]
    assert g.is_linear_ref_node_or_linear_ref_dummy_node(3) is False      # sibling 2 is linear
    assert sorted(g.get_reverse_edges_hashtable()[4]) == [2, 3]
    g2 = GraphArrays.from_obgraph(_AccessorOnly(g))
    for name in ("node_size", "seq", "edge_start", "edges", "rev_start", "rev_edges", "is_ref", "allele_freq"):
        assert np.array_equal(getattr(g, name), getattr(g2, name)), name


class _AccessorOnly:
    """Exposes only the obgraph accessor methods, to exercise GraphArrays.from_obgraph."""

    def __init__(self, g):
        self._g = g
        self.nodes = g.node_size
        self.chromosome_start_nodes = g.chromosome_start_nodes
        self.node_to_ref_offset = g.node_to_ref_offset

    def __getattr__(self, name):
        if name in ("max_node_id", "get_edges", "get_numeric_node_sequence", "get_reverse_edges_hashtable",
                    "is_linear_ref_node_or_linear_ref_dummy_node", "get_node_allele_frequencies", "get_first_node"):
            return getattr(self._g, name)
        raise AttributeError(name)


def test_supported_graph_checks():
    # a multi-node alternative allele (node 3 has no linear-ref predecessor) is supported: it only takes the kernels
    # off the fast path (classify_nodes reports `general`)
    from graph_kmer_index_amd.kmer_finder import classify_nodes
    g = GraphArrays.from_dicts({0: "ACGTACGT", 1: "A", 2: "C", 3: "G", 4: "TTTTTTTT"},
                               {0: [1, 2], 2: [3], 1: [4], 3: [4]}, [0, 1, 4])
    flags, general = classify_nodes(g, 4, 4)
    assert general and flags[3] & 16 and flags[2] & 8 and all(flags[n] & 4 for n in (0, 1, 4))
    assert flags.dtype == np.uint16 and (flags[4] >> 8) == 2            # up to two variant nodes (2, 3) right before node 4
    assert not classify_nodes(synthetic_snp_graph(20000, 300, k=31, seed=3), 31, 4)[1]
    # a second chromosome that starts with a node shorter than k gets no search root of its own (the reference prepends
    # its extra start point for the graph's first node only, kmer_finder.py:208-211): the node is never entered (DEAD)
    # and the run takes the general kernels; tests/test_two_chromosomes.py holds the reference's records for such graphs
    two = GraphArrays.from_dicts({0: "ACGTACGT", 1: "AC", 2: "GGGG"}, {1: [2]}, [0, 1, 2], chromosome_start_nodes=[0, 1])
    assert search_roots(two, 4) == [0, 0] and search_roots(two, 2) == [0, 0, 1]
    flags, general = classify_nodes(two, 4, 4, critical_nodes=[0, 2])
    assert general and flags[1] & 128 and not flags[2] & 128 and not flags[0] & 128
    # lossy restart table: critical (N, c) with 0 < c < k-1
    g = GraphArrays.from_dicts({0: "A", 1: "CTTT", 2: "TAAGGGG", 3: "AA", 4: ""}, {0: [1], 1: [2, 4], 2: [3], 4: [3]},
                               [0, 1, 2, 3])
    assert lossy_table(g, 3, [1], [1]) is None or lossy_table(g, 3, [1], [1])[1] == 1      # c=1 = k-2
    t = lossy_table(g, 5, [1], [1])
    assert t[1] == 1 and t[0] == 0xFFFF
    big = GraphArrays.from_dicts({0: "AC", 1: "A" * 40}, {0: [1]}, [0, 1])
    with pytest.raises(ValueError):
        lossy_table(big, 9, [1], [6])        # c=6 >= 3, node longer than 2k+3: reference undefined


def test_synthetic_generators_are_prefix_stable_and_wellformed():
    a = synthetic_snp_graph(100000, 900, k=31, seed=1234)
    assert a.seq.max() <= 3 and a.node_size.min() >= 1
    assert np.all(a.edges > np.repeat(np.arange(a.n_nodes), np.diff(a.edge_start)))       # topological ids
    alt = np.nonzero(a.is_ref == 0)[0]
    assert np.all(a.seq[a.seq_start[alt]] != a.seq[a.seq_start[alt - 1]])                # alt differs from ref allele
    assert np.allclose(a.allele_freq[alt] + a.allele_freq[alt - 1], 1.0)
    lin = synthetic_linear_graph(100000, 25000, seed=1234)
    assert lin.n_nodes == 4 and int(lin.seq_start[-1]) == 100000
    from graph_kmer_index_amd.graph import random_codes
    assert np.array_equal(random_codes(1000, 5), random_codes(5000, 5)[:1000])


def test_topological_rank_host_function():
    # gki_topological_rank (host side of libgki_hip.so): ranks respect every edge; a cycle is refused
    from graph_kmer_index_amd import _lib
    import ctypes as C
    from graphgen import overlapping_bubble_graph
    lib = _lib.load()
    rng = np.random.default_rng(4)
    for _ in range(20):
        seqs, edges, lin, af = overlapping_bubble_graph(rng, n_var=int(rng.integers(3, 9)))
        g = GraphArrays.from_dicts(seqs, edges, lin, af)
        rank = np.full(g.n_nodes, -1, dtype=np.int32)
        assert lib.gki_topological_rank(g.n_nodes, _lib.hptr(g.edge_start), _lib.hptr(g.edges), _lib.hptr(rank)) == 0
        assert sorted(rank.tolist()) == list(range(g.n_nodes))
        src = np.repeat(np.arange(g.n_nodes), np.diff(g.edge_start))
        assert np.all(rank[src] < rank[g.edges])
    edge_start = np.array([0, 1, 2], dtype=np.int64)
    cyc = np.array([1, 0], dtype=np.int32)
    out = np.zeros(2, dtype=np.int32)
    assert lib.gki_topological_rank(2, _lib.hptr(edge_start), _lib.hptr(cyc), _lib.hptr(out)) != 0
    assert b"cycle" in C.c_char_p(lib.gki_last_error()).value


def test_product_library_has_no_tuning_or_debug_switches():
    """The A/B knobs and the expansion-skipping diagnostic exist only in `make tuning` builds (-DGKI_TUNING)."""
    from graph_kmer_index_amd import _lib
    with open(_lib.LIB_PATH, "rb") as fh:
        blob = fh.read()
    for name in (b"GKI_DBG", b"GKI_SW", b"GKI_NE_CAP", b"GKI_RUN_BLOCKS", b"GKI_BND_BLOCKS", b"GKI_INT_BLOCKS",
                 b"GKI_OVERLAP_EMIT", b"GKI_BOUNDARY_FIRST", b"g_dbg_skip_expand"):
        assert name not in blob, name


def test_early_stop_bench_start_positions_follow_the_reference_pattern():
    # bench.py's `early_stop_search` record and tools/bench_forward.py share this generator: per SNP site the linear-ref
    # positions 2, 6, ... 26 bases before the variant (unique_variant_kmers.py:119-140 at k=31), on the segment in front
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_forward import start_positions
    from graph_kmer_index_amd.graph import synthetic_snp_graph
    g = synthetic_snp_graph(300000, 400, k=31, seed=7)
    nodes, offs = start_positions(g, 31)
    assert nodes.dtype == np.int32 and offs.dtype == np.int32 and len(nodes) == len(offs)
    n_sites = int(((g.is_ref == 0) & (g.node_size == 1)).sum())
    assert 5 * n_sites < len(nodes) <= 7 * n_sites
    assert np.all(g.is_ref[nodes] == 1) and np.all(offs >= 0) and np.all(offs <= g.node_size[nodes])
    assert sorted(set((g.node_size[nodes] - offs).tolist())) == [2, 6, 10, 14, 18, 22, 26]
    assert np.all(np.diff(nodes.astype(np.int64)) >= 0)          # site order, like a loop over the variants
    # every start's segment is followed by a bubble: its successors are a ref allele and the alt allele
    succ_n = g.edge_start[nodes.astype(np.int64) + 1] - g.edge_start[nodes.astype(np.int64)]
    assert np.all(succ_n == 2)


def test_scalar_getters_host_form_against_the_oracle(tmp_path):
    """CollisionFreeKmerIndex.get / get_frequency / __contains__ answer one k-mer from the object's own NumPy arrays
    (collision_free_kmer_index.py:303-315, :336-352) -- no device involved: checked here against the oracle's get on an
    oracle-built index, hits, misses, the max_hits rule, int64 and uint64 columns, and after a file round trip."""
    from graph_kmer_index_amd import CollisionFreeKmerIndex
    from graph_kmer_index_amd.kmer_hashing import kmer_hash_to_reverse_complement_hash
    rng = np.random.default_rng(12)
    pool = rng.integers(0, 4 ** 31, size=3000, dtype=np.int64)
    kmers = pool[rng.integers(0, len(pool), size=20000)]
    kmers[:1500] = pool[0]                                        # one k-mer with 1500 records
    nodes = rng.integers(0, 900, size=len(kmers)).astype(np.uint32)
    refs = rng.integers(0, 40, size=len(kmers)).astype(np.uint64)
    af = rng.random(len(kmers)).astype(np.float32)
    ref = oracle.index_build(kmers.astype(np.uint64), nodes, refs, af, modulo=4001)
    queries = [int(x) for x in pool[:200]] + [int(x) for x in rng.integers(0, 4 ** 31, size=80)]

    def oracle_freq(q):
        f = 0
        for x in (q, int(kmer_hash_to_reverse_complement_hash(q, 31))):
            r = oracle.index_get(ref, x, 10 ** 15)
            f += 0 if r[0] is None else int(r[2][0])
        return f

    for kmer_dtype in (np.uint64, np.int64):
        index = CollisionFreeKmerIndex(ref["_hashes_to_index"], ref["_n_kmers"], ref["_nodes"], ref["_ref_offsets"],
                                       ref["_kmers"].astype(kmer_dtype), 4001, ref["_frequencies"], ref["_allele_frequencies"])
        if kmer_dtype is np.int64:
            index.to_file(str(tmp_path / "idx"))
            index = CollisionFreeKmerIndex.from_file(str(tmp_path / "idx"))
        n_none = 0
        for mh in (1, 10, 10 ** 15):
            for q in queries:
                got, want = index.get(q, max_hits=mh), oracle.index_get(ref, q, mh)
                if want[0] is None:
                    n_none += 1
                    assert got == (None, None, None, None)
                    continue
                for a, b in zip(got, want):
                    assert a.dtype == b.dtype and np.array_equal(a, b)
        assert n_none > 100
        assert [index.get_frequency(q) for q in queries] == [oracle_freq(q) for q in queries]
        assert [q in index for q in queries] == [oracle.index_get(ref, q, 10 ** 15)[0] is not None for q in queries]
        assert list(index.get_nodes(queries[3])) == list(oracle.index_get(ref, queries[3])[0])
    assert index._device is None                                  # nothing above touched the device
    assert index.get(2 ** 64 + 5) == (None, None, None, None) and index.get(-3) == (None, None, None, None)


_GRAPH_COLS = ("node_size", "seq", "edge_start", "edges", "rev_start", "rev_edges", "is_ref", "allele_freq", "exists")


def test_from_obgraph_whole_arrays_equal_the_accessor_walk():
    """GraphArrays.from_obgraph takes whole arrays when the object offers them (ragged edges / sequences, vectorised
    flags) and otherwise walks the accessor methods node by node: both give the same arrays on random graphs (SNPs,
    indels with empty nodes, chains), the obgraph stand-in goes the fast way, and arrays that disagree with the accessors
    are not trusted."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "standins"))
    from obgraph_like import ObgraphLike
    from obgraph.graph import Graph as StandinGraph
    import graphgen
    from graph_kmer_index_amd import graph as graph_mod
    rng = np.random.default_rng(77)
    n_graphs = 0
    for i in range(220):
        gen = (graphgen.random_bubble_graph, graphgen.nested_bubble_graph, graphgen.deep_nested_graph)[i % 3]
        made = gen(rng)
        seqs, edges, lin = made[0], made[1], made[2]
        g = GraphArrays.from_dicts(seqs, edges, lin)
        fast = GraphArrays._from_obgraph_arrays(ObgraphLike(g))
        assert fast is not None and GraphArrays._first_disagreement(ObgraphLike(g), fast, 50) is None
        slow = GraphArrays._from_obgraph_accessors(ObgraphLike(g, with_arrays=False))
        assert GraphArrays._from_obgraph_arrays(ObgraphLike(g, with_arrays=False)) is None
        for name in _GRAPH_COLS:
            assert np.array_equal(getattr(fast, name), getattr(slow, name)), (i, name)
            assert np.array_equal(getattr(fast, name), getattr(g, name)), (i, name)
        assert (fast.first_node, fast._chromosome_start_nodes) == (slow.first_node, slow._chromosome_start_nodes)
        if i < 60:                                   # the reference's own entry: obgraph (stand-in) objects
            sg = StandinGraph.from_dicts(seqs, edges, lin)
            via_fast = GraphArrays._from_obgraph_arrays(sg)
            assert via_fast is not None
            via_slow = GraphArrays._from_obgraph_accessors(sg)
            for name in _GRAPH_COLS:
                assert np.array_equal(getattr(via_fast, name), getattr(via_slow, name)), (i, name)
        n_graphs += 1
    assert n_graphs >= 200
    # whole arrays that contradict the accessors: found by the sample, the accessor walk answers
    g = synthetic_snp_graph(30000, 300, k=31, seed=4)
    liar = ObgraphLike(g)
    twisted = g.edges.copy()
    twisted[::7] = np.roll(twisted, 1)[::7]
    liar.edges = type(liar.edges)(g.edge_start, twisted)
    assert GraphArrays._first_disagreement(liar, GraphArrays._from_obgraph_arrays(liar), 2000) is not None
    got = GraphArrays.from_obgraph(liar)
    for name in _GRAPH_COLS:
        assert np.array_equal(getattr(got, name), getattr(g, name)), name
    # rows not laid out back to back (a ragged view after row selection): gathered
    rs, flat = graph_mod._ragged_rows(_Shuffled(g.edge_start, g.edges), g.n_nodes)
    assert np.array_equal(rs, g.edge_start) and np.array_equal(flat, g.edges)


class _Shuffled:
    """A ragged array whose rows lie in reverse order in its flat data."""

    def __init__(self, row_start, flat):
        lengths = np.diff(row_start)
        order = np.arange(len(lengths))[::-1]
        pieces = [flat[row_start[r]:row_start[r + 1]] for r in order]
        self._data = np.concatenate(pieces) if pieces else flat[:0]
        starts = np.zeros(len(lengths), dtype=np.int64)
        starts[order] = np.concatenate([[0], np.cumsum(lengths[order])[:-1]])

        class S:
            pass
        self.shape = S()
        self.shape.starts, self.shape.lengths = starts, lengths

    def ravel(self):
        return self._data
