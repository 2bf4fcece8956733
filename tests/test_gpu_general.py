"""-m gpu: DenseKmerFinder.find() on graphs OUTSIDE the simple class -- nodes with no linear-ref predecessor
(variants inside alternative alleles, multi-node alleles), `only_follow_nodes`, and the reference's
`assert len(next_nodes) == 1` (kmer_finder.py:402) -- against the oracle and the reference-generated fixtures."""
import json
import os
import numpy as np
import pytest

from gpu_util import finder_cols, assert_same_records
from graph_kmer_index_amd import DenseKmerFinder, GraphArrays, CriticalGraphPaths, _lib
from graph_kmer_index_amd.kmer_finder import classify_nodes
from graphgen import nested_bubble_graph, deep_nested_graph, random_bubble_graph, overlapping_bubble_graph
from oracle import oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def run_case(g, k, M, one, follow=None, chunk=None, store=None):
    """GPU find() vs oracle.find() incl. the assertion; returns 'ok' / 'assert' / 'skip'."""
    try:
        cn, co = oracle.critical_paths(g, k)
    except oracle.OracleError:
        return "skip"
    kw = {}
    if chunk is not None:
        kw = dict(start_at_critical_path_number=chunk[0], stop_at_critical_path_number=chunk[1])
    try:
        exp, flags = oracle.find(g, k, (cn, co), one, M, only_follow_nodes=follow, only_store_nodes=store,
                                 return_flags=True, **kw)
        err = None
    except oracle.OracleError as e:
        exp, flags, err = None, 0, e.code
    if err not in (None, 3) or flags & oracle.ORC_FLAG_UNDEFINED_BULK:
        return "skip"
    f = DenseKmerFinder(g, k, critical_graph_paths=CriticalGraphPaths(cn, co), only_save_one_node_per_kmer=one,
                        max_variant_nodes=M, only_follow_nodes=follow, only_store_nodes=store, **kw)
    if err == 3:
        with pytest.raises(AssertionError):
            f.find()
        return "assert"
    f.find()
    assert_same_records(finder_cols(f), exp)
    return "ok"


GENERATORS = {
    "nested": lambda r: nested_bubble_graph(r, n_var=int(r.integers(2, 6)), p_nest=0.7),
    "deep": lambda r: deep_nested_graph(r, n_var=int(r.integers(1, 4)), max_depth=int(r.integers(1, 4))),
    "deep_long": lambda r: deep_nested_graph(r, n_var=int(r.integers(1, 4)), max_depth=2, min_ref=3, max_ref=40,
                                             max_allele=12),
    "bubble": lambda r: random_bubble_graph(r),
    "overlap": lambda r: overlapping_bubble_graph(r, n_var=int(r.integers(3, 7))),
}


@pytest.mark.parametrize("gen,n,seed,kmax", [("nested", 60, 31, 9), ("deep", 60, 32, 9), ("deep_long", 50, 33, 31),
                                             ("nested", 30, 34, 31)])
def test_nested_graphs_against_oracle(gen, n, seed, kmax):
    rng = np.random.default_rng(seed)
    seen = {"ok": 0, "assert": 0, "skip": 0}
    general = 0
    for _ in range(n):
        seqs, edges, lin, af = GENERATORS[gen](rng)
        g = GraphArrays.from_dicts(seqs, edges, lin, af)
        for _ in range(4):
            k = int(rng.integers(3, kmax + 1))
            M = int(rng.choice([0, 1, 2, 3, 4, 5, 100]))
            one = bool(rng.integers(0, 2))
            general += classify_nodes(g, k, M)[1]
            seen[run_case(g, k, M, one)] += 1
    assert seen["ok"] > n and seen["assert"] > 0 and general > n, (seen, general)


@pytest.mark.parametrize("gen,n,seed", [("bubble", 40, 41), ("nested", 40, 42), ("deep", 40, 43), ("overlap", 30, 44)])
def test_only_follow_nodes_in_find(gen, n, seed):
    """kmer_finder.py:386-388 in find(): forced successors bypass the limit and hide their siblings."""
    rng = np.random.default_rng(seed)
    seen = {"ok": 0, "assert": 0, "skip": 0}
    for _ in range(n):
        seqs, edges, lin, af = GENERATORS[gen](rng)
        g = GraphArrays.from_dicts(seqs, edges, lin, af)
        cand = [x for x in seqs if not g.is_ref[x] or rng.random() < 0.15]
        if not cand:
            continue
        for _ in range(4):
            k = int(rng.integers(3, 10))
            M = int(rng.choice([0, 1, 2, 3, 100]))
            follow = set(int(x) for x in rng.choice(cand, size=int(rng.integers(1, len(cand) // 2 + 2)), replace=False)
                         if True)
            seen[run_case(g, k, M, bool(rng.integers(0, 2)), follow=follow)] += 1
    assert seen["ok"] > n, seen


def test_chunked_runs_and_store_filter_on_nested_graphs():
    rng = np.random.default_rng(51)
    ok = 0
    for _ in range(40):
        seqs, edges, lin, af = deep_nested_graph(rng, n_var=int(rng.integers(3, 7)), max_depth=2, min_ref=2, max_ref=12)
        g = GraphArrays.from_dicts(seqs, edges, lin, af)
        k = int(rng.integers(3, 8))
        try:
            n_crit = len(oracle.critical_paths(g, k)[0])
        except oracle.OracleError:
            continue
        a = int(rng.integers(0, n_crit + 1))
        b = int(rng.integers(a, n_crit + 2))
        ok += run_case(g, k, 100, bool(rng.integers(0, 2)), chunk=(a, b)) == "ok"
        store = set(int(x) for x in rng.choice(list(seqs), size=max(1, len(seqs) // 3), replace=False))
        ok += run_case(g, k, 100, False, store=store) == "ok"
    assert ok > 40


def test_every_critical_point_starts_a_search_of_its_own():
    """A forced allele plus a tight limit can cut the graph in two for the search that comes from upstream; the next
    critical point still starts its own search (kmer_finder.py:190-232), so a chunk that begins there has records and
    the linear-successor assertion only fires in the chunk that reaches the offending node (found by the soak)."""
    rng = np.random.default_rng(53)
    seen = {"ok": 0, "assert": 0, "skip": 0}
    for _ in range(120):
        seqs, edges, lin, af = nested_bubble_graph(rng, n_var=int(rng.integers(4, 12)), min_ref=1, max_ref=24, p_nest=0.7)
        g = GraphArrays.from_dicts(seqs, edges, lin, af)
        k = int(rng.integers(6, 20))
        try:
            n_crit = len(oracle.critical_paths(g, k)[0])
        except oracle.OracleError:
            continue
        variant = np.nonzero(g.is_ref == 0)[0]
        follow = set(int(x) for x in rng.choice(variant, size=max(1, len(variant) // 4), replace=False))
        a = int(rng.integers(0, n_crit + 1))
        b = int(rng.integers(a, n_crit + 2))
        for chunk in (None, (a, b)):
            seen[run_case(g, k, int(rng.choice([1, 2, 3])), bool(rng.integers(0, 2)), follow=follow, chunk=chunk)] += 1
    assert seen["ok"] > 60 and seen["assert"] > 10, seen


def test_flat_layouts_on_nested_graphs():
    """find_flat_on_device (both layouts) = find() on general graphs too."""
    rng = np.random.default_rng(52)
    done = 0
    for _ in range(30):
        seqs, edges, lin, af = deep_nested_graph(rng, n_var=int(rng.integers(2, 6)), max_depth=2, min_ref=2, max_ref=30,
                                                 max_allele=8)
        g = GraphArrays.from_dicts(seqs, edges, lin, af)
        k = int(rng.integers(3, 20))
        one = bool(rng.integers(0, 2))
        try:
            cn, co = oracle.critical_paths(g, k)
            exp, flags = oracle.find(g, k, (cn, co), one, 100, return_flags=True)
        except oracle.OracleError:
            continue
        if flags & oracle.ORC_FLAG_UNDEFINED_BULK:
            continue
        pos = g.position_id_base()[exp["start_nodes"]] + exp["start_offsets"]
        want = sorted(zip(exp["kmers"].tolist(), exp["nodes"].tolist(), pos.tolist(),
                          exp["allele_frequencies"].astype(np.float32).tolist()))
        for split in (True, False):
            f = DenseKmerFinder(g, k, critical_graph_paths=CriticalGraphPaths(cn, co), only_save_one_node_per_kmer=one,
                                max_variant_nodes=100)
            d = f.find_flat_on_device(split_layout=split)
            f.synchronize()
            fl = d.to_flat_kmers()
            got = sorted(zip(fl._hashes.astype(np.int64).tolist(), fl._nodes.astype(np.int64).tolist(),
                             fl._ref_offsets.astype(np.int64).tolist(), fl._allele_frequencies.tolist()))
            assert got == want
            d.free()
        done += 1
    assert done > 15


def test_reference_fixtures_for_nested_graphs():
    """Records the Python reference produced for nested graphs (tests/golden/make_golden.py -> finder_nested.json);
    cases where it raised its AssertionError are recorded as such."""
    with open(os.path.join(GOLD, "finder_nested.json")) as fh:
        cases = json.load(fh)
    assert len(cases) >= 40
    n_assert = 0
    for case in cases:
        seqs = {int(a): b for a, b in case["seqs"].items()}
        edges = {int(a): b for a, b in case["edges"].items()}
        g = GraphArrays.from_dicts(seqs, edges, case["linear"])
        follow = None if case.get("follow") is None else set(case["follow"])
        f = DenseKmerFinder(g, case["k"], only_save_one_node_per_kmer=case["one"], max_variant_nodes=case["M"],
                            only_follow_nodes=follow)
        if case["raises"]:
            with pytest.raises(AssertionError):
                f.find()
            n_assert += 1
            continue
        f.find()
        exp = dict(kmers=np.array(case["kmers"], np.int64), nodes=np.array(case["nodes"], np.int32),
                   start_nodes=np.array(case["start_nodes"], np.int32),
                   start_offsets=np.array(case["start_offsets"], np.int16),
                   allele_frequencies=np.array(case["allele_frequencies"], np.float64))
        assert_same_records(finder_cols(f), exp)
    assert 0 < n_assert < len(cases)


def test_struct_size_guard():
    """A binding built against another gki_find_params layout is refused, not read out of bounds."""
    import ctypes as C
    g = GraphArrays.from_dicts({0: "ACGTACGT"}, {}, [0])
    f = DenseKmerFinder(g, 3)
    p = _lib.FindParams(3, 4, 0, 0, 0, 0, g.n_nodes, 0)
    p.struct_size = 72
    n = C.c_int64(0)
    assert _lib.load().gki_finder_count(f._finder_handle(), C.byref(p), C.byref(n)) == 2
    assert b"struct_size" in _lib.load().gki_last_error()


def test_integration_stub_runs():
    """Executes the binding INTEGRATION.md section B shows to a maintainer of the reference, verbatim, against the built
    library: README toy graph (BASELINE configs[0]) through graph_to_device + find, compared with the records the
    reference produced for it (tests/golden/finder_toy.json `readme_c1`)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "INTEGRATION.md")) as fh:
        text = fh.read()
    blocks = re.findall(r"```python\n(# graph_kmer_index/_gki\.py.*?)```", text, flags=re.S)
    assert len(blocks) == 1
    os.environ["GKI_LIB"] = _lib.LIB_PATH
    ns = {}
    exec(compile(blocks[0], "INTEGRATION.md#B", "exec"), ns)
    with open(os.path.join(GOLD, "finder_toy.json")) as fh:
        case = [c for c in json.load(fh) if c["name"] == "readme_c1"][0]
    g = GraphArrays.from_dicts({int(a): b for a, b in case["seqs"].items()}, {int(a): b for a, b in case["edges"].items()},
                               case["linear"])
    h = ns["graph_to_device"](g.node_size, g.seq, g.edge_start, g.edges, g.rev_start, g.rev_edges, g.is_ref, g.allele_freq)
    kmers, start_nodes, start_offsets, nodes, af = ns["find"](h, case["k"], 4, False, g.n_nodes)
    got = dict(kmers=kmers, nodes=nodes, start_nodes=start_nodes, start_offsets=start_offsets, allele_frequencies=af)
    exp = dict(kmers=np.array(case["kmers"], np.int64), nodes=np.array(case["nodes"], np.int32),
               start_nodes=np.array(case["start_nodes"], np.int32), start_offsets=np.array(case["start_offsets"], np.int16),
               allele_frequencies=np.array(case["allele_frequencies"], np.float64))
    assert len(kmers) == 26
    assert_same_records(got, exp)
    # the stub refuses a library with another struct layout
    assert "gki_find_params_size" in blocks[0]


@pytest.mark.parametrize("one", [True, False])
def test_only_store_nodes_on_device(one):
    """kmer_finder.py:153 applied by the kernels (count and emit passes), incl. the bulk path that ignores the filter
    (:370-374): find() and find_flat_on_device() against oracle.find(only_store_nodes=...)."""
    from graph_kmer_index_amd.graph import synthetic_snp_graph, synthetic_indel_graph
    rng = np.random.default_rng(61 + one)
    cases = [synthetic_snp_graph(40000, 500, k=31, seed=5), synthetic_indel_graph(30000, 400, k=31, seed=6)]
    for _ in range(12):
        seqs, edges, lin, af = deep_nested_graph(rng, n_var=int(rng.integers(2, 6)), max_depth=2, min_ref=2, max_ref=90,
                                                 max_allele=6)
        cases.append(GraphArrays.from_dicts(seqs, edges, lin, af))
    done = 0
    for i, g in enumerate(cases):
        k = 31 if i < 2 else int(rng.integers(3, 16))
        ids = np.nonzero(g.exists)[0]
        if i < 2:
            store = set(int(x) for x in np.nonzero(g.is_ref == 0)[0][::2])         # what UniqueVariantKmersFinder passes
        else:
            store = set(int(x) for x in rng.choice(ids, size=max(1, len(ids) // 3), replace=False))
        try:
            cn, co = oracle.critical_paths(g, k)
            exp, flags = oracle.find(g, k, (cn, co), one, 100, only_store_nodes=store, return_flags=True)
        except oracle.OracleError:
            continue
        if flags & oracle.ORC_FLAG_UNDEFINED_BULK:
            continue
        f = DenseKmerFinder(g, k, critical_graph_paths=CriticalGraphPaths(cn, co), only_save_one_node_per_kmer=one,
                            max_variant_nodes=100, only_store_nodes=store)
        f.find()
        assert_same_records(finder_cols(f), exp)
        pos = g.position_id_base()[exp["start_nodes"]] + exp["start_offsets"]
        want = sorted(zip(exp["kmers"].tolist(), exp["nodes"].tolist(), pos.tolist()))
        for split in (True, False):
            d = f.find_flat_on_device(split_layout=split)
            f.synchronize()
            fl = d.to_flat_kmers()
            assert sorted(zip(fl._hashes.astype(np.int64).tolist(), fl._nodes.astype(np.int64).tolist(),
                              fl._ref_offsets.astype(np.int64).tolist())) == want
            d.free()
        done += 1
    assert done >= 8


def test_early_stop_search_on_nested_graphs_incl_assertion():
    """find_only_kmers_starting_at_position on graphs with nested variants: the reference's records in its order, and its
    AssertionError (kmer_finder.py:402) where a path at the limit ends a node without exactly one linear-ref successor."""
    rng = np.random.default_rng(71)
    n_ok = n_assert = 0
    for _ in range(60):
        seqs, edges, lin, af = deep_nested_graph(rng, n_var=int(rng.integers(1, 4)), max_depth=int(rng.integers(1, 3)))
        g = GraphArrays.from_dicts(seqs, edges, lin, af)
        k = int(rng.integers(3, 9))
        M = int(rng.choice([0, 1, 2, 4]))
        one = bool(rng.integers(0, 2))
        for node in rng.choice(list(seqs), size=4, replace=False):
            node = int(node)
            off = int(rng.integers(0, max(1, len(seqs[node]))))
            f = DenseKmerFinder(g, k, only_save_one_node_per_kmer=one, max_variant_nodes=M)
            try:
                exp = oracle.find_from_position(g, k, node, off, one, M)
            except oracle.OracleError as e:
                assert e.code == 3
                with pytest.raises(AssertionError):
                    f.find_only_kmers_starting_at_position(node, off)
                n_assert += 1
                continue
            f.find_only_kmers_starting_at_position(node, off)
            assert_same_records(finder_cols(f), exp, exact_order=True)
            n_ok += 1
    assert n_ok > 60 and n_assert > 5


def test_assertion_after_a_lossy_restart():
    """A single-edge chain right behind a nested bubble gives a critical point (N, c) with 0 < c < k-1: the search
    restarts there without history (SURVEY.md 8a' E1), so at the nodes that follow it holds a window shorter than k --
    and the assertion of kmer_finder.py:402 is decided on that shorter window (found by the soak)."""
    rng = np.random.default_rng(54)
    seen = {"ok": 0, "assert": 0, "skip": 0}
    for _ in range(150):
        seqs, edges, lin, af = nested_bubble_graph(rng, n_var=int(rng.integers(2, 7)), min_ref=1, max_ref=int(rng.integers(2, 14)),
                                                   p_nest=0.6, p_chain=0.5)
        g = GraphArrays.from_dicts(seqs, edges, lin, af)
        seen[run_case(g, int(rng.integers(3, 24)), int(rng.choice([1, 2, 3, 4])), bool(rng.integers(0, 2)))] += 1
    assert seen["ok"] > 30 and seen["assert"] > 30, seen


def test_host_and_device_forms_of_the_per_run_tables_agree():
    """gki_find_params takes the four per-run tables (lossy restarts, topological ranks, node flags, only_store_nodes) as
    host arrays, uploaded by every count, or as device pointers (what the Python finder hands in since round 4, uploaded
    once): chunked runs over nested graphs with a store filter and shuffled node ids give the same counts and the same
    records either way, and a finder's second count uploads nothing new."""
    import ctypes as C
    lib = _lib.load()

    def emit_flat(f, n):
        cols = [_lib.DeviceArray(max(n, 1), dt) for dt in (np.uint64, np.uint32, np.uint64, np.float32)]
        _lib.check(lib.gki_finder_emit_flat(f._finder_handle(), *[c.ptr for c in cols]))
        _lib.check(lib.gki_finder_synchronize(f._finder_handle()))
        return [c.to_host(n) for c in cols]

    rng = np.random.default_rng(404)
    n_cases = 0
    for i in range(40):
        seqs, edges, lin, af = nested_bubble_graph(rng, n_var=int(rng.integers(3, 9)), min_ref=1, max_ref=12, p_nest=0.6, p_chain=0.3)
        if i % 2:                                            # node ids that do not grow along the edges: the rank table
            ids = sorted(seqs)
            perm = dict(zip(ids, rng.permutation(ids).tolist()))
            seqs = {perm[a]: b for a, b in seqs.items()}
            edges = {perm[a]: [perm[x] for x in b] for a, b in edges.items()}
            lin = [perm[a] for a in lin]
        g = GraphArrays.from_dicts(seqs, edges, lin)
        k = int(rng.integers(3, 8))
        try:
            cp = CriticalGraphPaths.from_graph(g, k)
        except Exception:
            continue
        if len(cp.nodes) < 3:
            continue
        store = set(rng.choice(sorted(seqs), size=max(1, len(seqs) // 2), replace=False).tolist()) if i % 3 == 0 else None
        a, b = 1, len(cp.nodes) - 1
        try:
            f = DenseKmerFinder(g, k, critical_graph_paths=cp, max_variant_nodes=int(rng.integers(1, 5)), only_store_nodes=store,
                                start_at_critical_path_number=a, stop_at_critical_path_number=b)
            p = f._params()
        except (ValueError, AssertionError, RecursionError):
            continue
        if p is None:
            continue
        lossy, rank, flags, st = p._keep[:4]
        n_dev, n_host = C.c_int64(0), C.c_int64(0)
        rc_dev = _lib.load().gki_finder_count(f._finder_handle(), C.byref(p), C.byref(n_dev))
        cols_dev = emit_flat(f, n_dev.value) if rc_dev == 0 else None
        general = bool(p.d_node_flags)
        q = _lib.FindParams(p.k, p.max_variant_nodes, p.one_node_per_kmer, p.layout, p.node_begin, p.off_begin, p.node_end, p.off_end,
                            _lib.hptr(lossy), _lib.hptr(rank), _lib.hptr(flags) if general else None, _lib.hptr(st))
        rc_host = _lib.load().gki_finder_count(f._finder_handle(), C.byref(q), C.byref(n_host))
        assert rc_dev == rc_host and n_dev.value == n_host.value, (i, rc_dev, rc_host)
        if rc_host == 0:
            cols_host = emit_flat(f, n_host.value)
            for x, y in zip(cols_dev, cols_host):
                assert np.array_equal(x, y)
            n_cases += 1
        tables = dict(f.__dict__.get("_resident_tables", {}))
        f._params_cache = None
        f._params()                                           # the same chunk again: the same device copies
        assert {k_: id(v[1]) for k_, v in f.__dict__.get("_resident_tables", {}).items()} == {k_: id(v[1]) for k_, v in tables.items()}
    assert n_cases >= 10
