"""-m gpu: hashing kernels, CollisionFreeKmerIndex build and batched probe against golden vectors and oracle."""
import os
import numpy as np
import pytest

from graph_kmer_index_amd import (ReverseKmerIndex, CollisionFreeKmerIndex, FlatKmers, ReadKmers, _lib, sequence_to_kmer_hash,
                                  kmer_hash_to_sequence)
from graph_kmer_index_amd.kmer_hashing import (kmer_hashes_to_reverse_complement_hash, kmer_hashes_to_complement_hashes,
                                               kmer_hash_to_reverse_complement_hash, power_array)
from graph_kmer_index_amd.read_kmers import hash_reads
from oracle import oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_reverse_complement_and_complement_golden():
    h = np.load(os.path.join(GOLD, "hashing.npz"))
    for k in (3, 9, 16, 31):
        x = h["rc_in_k%d" % k]
        rc = kmer_hashes_to_reverse_complement_hash(x, k)
        assert rc.dtype == np.uint64
        assert np.array_equal(rc, h["rc_out_k%d" % k])
        assert np.array_equal(kmer_hashes_to_complement_hashes(x, k), h["comp_out_k%d" % k])
        assert np.array_equal(kmer_hashes_to_reverse_complement_hash(rc, k), x)      # involution
    # tests/test_kmer_hashing.py:38-54 known sequences
    for seq, rc_seq in (("AcATaCAG", "ctgtatgt"), ("ACT", "agt"), ("G" * 31, "c" * 31)):
        hh = sequence_to_kmer_hash(seq)
        assert kmer_hash_to_sequence(int(kmer_hash_to_reverse_complement_hash(hh, len(seq))), len(seq)) == rc_seq


def test_reverse_complement_large_random_vs_oracle():
    rng = np.random.default_rng(1)
    for k in (1, 2, 15, 30, 31):
        x = rng.integers(0, 4 ** k, size=300001, dtype=np.uint64)
        assert np.array_equal(kmer_hashes_to_reverse_complement_hash(x, k), oracle.reverse_complement(x, k))
        assert np.array_equal(kmer_hashes_to_complement_hashes(x, k), oracle.complement(x, k))
    assert len(kmer_hashes_to_reverse_complement_hash(np.zeros(0, np.uint64), 31)) == 0


def test_read_kmers_golden_and_strands():
    h = np.load(os.path.join(GOLD, "hashing.npz"))
    reads = [str(r) for r in h["reads"]]
    for k in (5, 31):
        got, start = hash_reads(reads, k, 0)
        assert np.array_equal(got, h["read_kmers_k%d" % k])
        assert start[-1] == len(got)
        one = ReadKmers.get_kmers_from_read_dynamic(reads[0], power_array(k))
        assert np.array_equal(one, got[start[0]:start[1]])
    comp = str.maketrans("ACGTacgt", "TGCAtgca")
    rng = np.random.default_rng(2)
    reads = ["".join("ACGTNacgtn"[i] for i in rng.integers(0, 10, size=int(n))) for n in rng.integers(1, 400, size=300)]
    reads += ["", "A" * 31, "ACGT" * 50]
    for k in (4, 31):
        got, start = hash_reads(reads, k, 1)
        exp = [oracle.read_kmers(r.translate(comp)[::-1], k) for r in reads]
        assert np.array_equal(got, np.concatenate(exp))
        assert np.array_equal(np.diff(start), [len(e) for e in exp])
        fwd, _ = hash_reads(reads, k, 0)
        assert np.array_equal(fwd, np.concatenate([oracle.read_kmers(r, k) for r in reads]))


def test_hash_sequence_vs_oracle():
    import ctypes as C
    rng = np.random.default_rng(3)
    for n, k in ((1000, 31), (100003, 31), (64, 5), (31, 31), (30, 31)):
        codes = rng.integers(0, 4, size=n).astype(np.uint8)
        d_in = _lib.DeviceArray.from_host(codes)
        d_out = _lib.DeviceArray(max(n - k + 1, 1), np.uint64)
        _lib.check(_lib.load().gki_hash_sequence(d_in.ptr, n, k, d_out.ptr))
        assert np.array_equal(d_out.to_host(max(n - k + 1, 0)), oracle.hash_sequence(codes, k))


def _bucket_multisets(idx):
    out = {}
    h2i, nk = np.asarray(idx["_hashes_to_index"]), np.asarray(idx["_n_kmers"])
    for b in np.nonzero(nk)[0]:
        s, n = int(h2i[b]), int(nk[b])
        out[int(b)] = sorted(zip(np.asarray(idx["_kmers"][s:s + n]).tolist(), np.asarray(idx["_nodes"][s:s + n]).tolist(),
                                 np.asarray(idx["_ref_offsets"][s:s + n]).tolist(),
                                 np.asarray(idx["_frequencies"][s:s + n]).tolist(),
                                 np.asarray(idx["_allele_frequencies"][s:s + n], dtype=np.float64).tolist()))
    return out


def _attrs(index):
    return {name: getattr(index, name) for name in ("_hashes_to_index", "_n_kmers", "_nodes", "_ref_offsets", "_kmers",
                                                    "_frequencies", "_allele_frequencies")}


@pytest.mark.parametrize("tag,kw", [("kat", {}), ("rand", {}), ("rand_skipfreq", {"skip_frequencies": True}),
                                    ("rand_nosingle", {"skip_singletons": True})])
def test_index_build_and_get_golden(tag, kw, tmp_path):
    z = np.load(os.path.join(GOLD, "index.npz"))
    modulo = int(z[tag + "_modulo"])
    flat = FlatKmers(z[tag + "_in_hashes"], z[tag + "_in_nodes"], z[tag + "_in_ref_offsets"], z[tag + "_in_af"])
    idx = CollisionFreeKmerIndex.from_flat_kmers(flat, modulo=modulo, **kw)
    ref = {name: z[tag + name] for name in ("_hashes_to_index", "_n_kmers", "_nodes", "_ref_offsets", "_kmers",
                                            "_frequencies", "_allele_frequencies")}
    assert idx._hashes_to_index.dtype == np.int32 and idx._n_kmers.dtype == np.uint32
    assert idx._frequencies.dtype == np.uint16
    assert np.array_equal(idx._hashes_to_index, ref["_hashes_to_index"])
    assert np.array_equal(idx._n_kmers, ref["_n_kmers"])
    for name in ("_nodes", "_ref_offsets", "_kmers", "_allele_frequencies"):
        assert getattr(idx, name).dtype == ref[name].dtype, name
    assert _bucket_multisets(_attrs(idx)) == _bucket_multisets(ref)
    # element-wise against the oracle (both stable)
    o = oracle.index_build(z[tag + "_in_hashes"], z[tag + "_in_nodes"], z[tag + "_in_ref_offsets"], z[tag + "_in_af"],
                           modulo=modulo, **kw)
    for name in ref:
        assert np.array_equal(getattr(idx, name), o[name]), name
    # npz round trip with the reference's keys (collision_free_kmer_index.py:393-420)
    path = str(tmp_path / "idx")
    idx.to_file(path)
    back = CollisionFreeKmerIndex.from_file(path)
    assert set(np.load(path + ".npz").keys()) == {"hashes_to_index", "n_kmers", "nodes", "ref_offsets", "kmers", "modulo",
                                                  "frequencies", "allele_frequencies"}
    for index in (idx, back):
        for mh in (10, 1):
            pos = 0
            for q, ne in zip(z[tag + "_queries"], z[tag + "_get%d_n" % mh]):
                r = index.get(int(q), max_hits=mh)
                if ne < 0:
                    assert r == (None, None, None, None)
                    continue
                exp = sorted(zip(z[tag + "_get%d_nodes" % mh][pos:pos + ne].tolist(),
                                 z[tag + "_get%d_ref_offsets" % mh][pos:pos + ne].tolist(),
                                 z[tag + "_get%d_frequencies" % mh][pos:pos + ne].tolist(),
                                 z[tag + "_get%d_af" % mh][pos:pos + ne].tolist()))
                got = sorted(zip(r[0].tolist(), r[1].tolist(), r[2].tolist(), np.asarray(r[3], np.float64).tolist()))
                assert got == exp
                pos += ne


def test_reference_known_answer_fixture():
    # tests/test_collision_free_kmer_index.py:6-23 of the reference
    flat = FlatKmers(np.array([1, 1, 2, 2, 4, 5, 3], dtype=np.uint64), np.array([5, 6, 7, 8, 10, 11, 100]),
                     np.array([1, 1, 2, 3, 10, 11, 100]))
    index = CollisionFreeKmerIndex.from_flat_kmers(flat, modulo=4)
    assert list(index.get(1)[0]) == [5, 6]
    assert list(index.get(1)[1]) == [1, 1]
    assert list(index.get(5)[0]) == [11]
    assert index.get(7) == (None, None, None, None)
    assert 3 in index and 9 not in index
    n, r, q, f = index.get_nodes_and_ref_offsets_from_multiple_kmers(np.array([1, 5]))
    assert n.tolist() == [5, 6, 11] and r.tolist() == [1, 1, 11] and q.tolist() == [0.0, 0.0, 1.0]
    assert f.tolist() == [1, 1, 1]
    assert index.get_nodes_from_multiple_kmers(np.array([9, 9])).tolist() == []
    assert index.has_kmers(np.array([1, 2, 3, 10, 10, 12, 100, 101, 102, 5], dtype=np.uint64)).tolist() == \
        [True, True, True, False, False, False, False, False, False, True]     # :30-34
    # tests/test_collision_free_kmer_index.py:30-34 of the reference calls the process-pool form
    assert index.has_kmers_parallel(np.array([1, 2, 3, 10, 10, 12, 100, 101, 102, 5], dtype=np.uint64), 4).tolist() == \
        [True, True, True, False, False, False, False, False, False, True]


@pytest.mark.parametrize("n,modulo,n_distinct", [(200000, 452930477, 150000), (300000, 65537, 5000), (50000, 7, 40),
                                                  (70000, 1000003, 3)])
def test_index_build_random_vs_oracle(n, modulo, n_distinct):
    """Includes buckets far larger than the per-lane path (large-bucket sort) and tiny moduli."""
    rng = np.random.default_rng(n)
    pool = rng.integers(0, 4 ** 31, size=n_distinct, dtype=np.uint64)
    kmers = pool[rng.integers(0, n_distinct, size=n)]
    nodes = rng.integers(0, 1 << 24, size=n).astype(np.uint32)
    refs = (kmers % np.uint64(50)) + rng.integers(0, 30, size=n).astype(np.uint64)
    af = rng.uniform(0, 1, size=n).astype(np.float32)
    idx = CollisionFreeKmerIndex.from_flat_kmers(FlatKmers(kmers, nodes, refs, af), modulo=modulo)
    o = oracle.index_build(kmers, nodes, refs, af, modulo=modulo)
    for name in ("_hashes_to_index", "_n_kmers", "_kmers", "_nodes", "_ref_offsets", "_allele_frequencies", "_frequencies"):
        assert np.array_equal(getattr(idx, name), o[name]), name
    # batched probe == loop of oracle gets (hits in bucket order)
    queries = np.concatenate([pool[:2000], rng.integers(0, 4 ** 31, size=500, dtype=np.uint64)])
    for mh in (10, 3, 10 ** 9):
        got_nodes, got_refs, got_q, got_f = idx.get_nodes_and_ref_offsets_from_multiple_kmers(queries, max_hits=mh)
        en, er, eq, ef = [], [], [], []
        for i, q in enumerate(queries[:700]):
            r = oracle.index_get(o, int(q), max_hits=mh)
            if r[0] is None:
                continue
            en.append(r[0]); er.append(r[1]); eq.append(np.full(len(r[0]), i)); ef.append(r[2])
        m = sum(len(x) for x in en)
        sel = got_q < 700
        assert sel.sum() == m
        if m:
            assert np.array_equal(got_nodes[sel], np.concatenate(en)) and np.array_equal(got_refs[sel], np.concatenate(er))
            assert np.array_equal(got_q[sel], np.concatenate(eq)) and np.array_equal(got_f[sel], np.concatenate(ef))


def test_index_from_finder_output_float64_af():
    from graph_kmer_index_amd import DenseKmerFinder
    from graph_kmer_index_amd.graph import synthetic_snp_graph
    g = synthetic_snp_graph(60000, 800, k=31, seed=4)
    f = DenseKmerFinder(g, 31)
    f.find()
    fl = f.get_flat_kmers(v="1")             # int64 hashes, int32 nodes, int64 ref_offsets, float64 af
    idx = CollisionFreeKmerIndex.from_flat_kmers(fl, modulo=200003)
    o = oracle.index_build(fl._hashes, fl._nodes, fl._ref_offsets, fl._allele_frequencies, modulo=200003)
    for name in ("_hashes_to_index", "_n_kmers", "_kmers", "_nodes", "_ref_offsets", "_allele_frequencies", "_frequencies"):
        assert getattr(idx, name).dtype == o[name].dtype, name
        assert np.array_equal(getattr(idx, name), o[name]), name
    assert idx.get_frequency(int(fl._hashes[0])) >= 1


def test_set_frequencies_using_other_index_and_complement():
    rng = np.random.default_rng(8)
    k = 31
    pool = rng.integers(0, 4 ** k, size=300, dtype=np.uint64)
    rc = oracle.reverse_complement(pool, k)
    other_kmers = np.concatenate([pool[:200], rc[100:250], pool[:50]])
    other = CollisionFreeKmerIndex.from_flat_kmers(
        FlatKmers(other_kmers, np.arange(len(other_kmers), dtype=np.uint32),
                  rng.integers(0, 5, size=len(other_kmers)).astype(np.uint64), np.ones(len(other_kmers), np.float32)),
        modulo=1009)
    mine_kmers = np.concatenate([pool, pool[:30]])
    mine = CollisionFreeKmerIndex.from_flat_kmers(
        FlatKmers(mine_kmers, np.arange(len(mine_kmers), dtype=np.uint32), np.zeros(len(mine_kmers), np.uint64),
                  np.ones(len(mine_kmers), np.float32)), modulo=257)
    mine.set_frequencies_using_other_index(other, multiplier=2, min_frequency=1)
    for i in range(0, len(mine._kmers), 7):
        km = int(mine._kmers[i])
        f = 0
        for q in (km, int(oracle.reverse_complement(np.array([km], np.uint64), k)[0])):
            r = other.get(q, max_hits=10 ** 15)
            if r[0] is not None:
                f += int(r[2][0])
        assert int(mine._frequencies[i]) == max(1, 2 * f)
        assert other.get_frequency(km) == f
    comp = mine.convert_kmers_to_complement(k=k)
    assert sorted(np.asarray(comp._kmers).tolist()) == sorted(oracle.complement(mine_kmers, k).tolist())
    assert comp._modulo == mine._modulo


def test_map_kmers_fused_node_counts():
    rng = np.random.default_rng(12)
    n = 50000
    pool = rng.integers(0, 4 ** 31, size=4000, dtype=np.uint64)
    kmers = pool[rng.integers(0, len(pool), size=n)]
    nodes = rng.integers(0, 3000, size=n).astype(np.uint32)
    idx = CollisionFreeKmerIndex.from_flat_kmers(FlatKmers(kmers, nodes, np.arange(n, dtype=np.uint64),
                                                           np.ones(n, np.float32)), modulo=100003)
    queries = np.concatenate([pool[:1500], pool[:700], rng.integers(0, 4 ** 31, size=900, dtype=np.uint64)])
    got = idx.map_kmers(queries, 3000)
    # expectation from the batched probe (itself checked against the oracle above)
    hit_nodes = idx.get_nodes_from_multiple_kmers(queries, max_hits=2 ** 62)
    exp = np.bincount(hit_nodes.astype(np.int64), minlength=3000)
    assert got.dtype == np.uint32 and np.array_equal(got, exp)
    assert idx.has_kmers(queries).sum() == 2200


# ------------------------------------------------------------------ ReverseKmerIndex (reverse_kmer_index.py:47-83)
def _reverse_index_numpy(nodes, kmers, refs):
    """the reference's from_flat_kmers with a stable argsort"""
    order = np.argsort(nodes, kind="stable")
    snodes = nodes[order].astype(np.int64)
    first = np.flatnonzero(np.ediff1d(snodes, to_begin=1))
    uniq = snodes[first]
    index = np.zeros(int(nodes.max()) + 1, np.uint32)
    counts = np.zeros(int(nodes.max()) + 1, np.uint16)
    index[uniq] = first
    counts[uniq] = np.ediff1d(first, to_end=len(nodes) - first[-1]).astype(np.uint16)
    return index, counts, kmers[order], refs[order]


def test_reverse_index_reference_known_answer(tmp_path):
    # tests/test_reverse_kmer_index.py:6-22 of the reference
    flat = FlatKmers(np.array([10, 3, 11, 4]), np.array([5, 3, 5, 8]), np.array([1, 2, 3, 4]))
    r = ReverseKmerIndex.from_flat_kmers(flat)
    assert list(r.get_node_kmers(5)) == [10, 11]
    assert list(r.get_node_kmers(3)) == [3]
    assert list(r.get_node_kmers(8)) == [4]
    assert list(r.get_node_kmers(4)) == [] and r.get_node_kmers_and_ref_positions(0) == [[], []]
    km, rp = r.get_node_kmers_and_ref_positions(5)
    assert list(km) == [10, 11] and list(rp) == [1, 3]
    assert r.nodes_to_index_positions.dtype == np.uint32 and r.nodes_to_n_hashes.dtype == np.uint16
    assert r.hashes.dtype == flat._hashes.dtype
    r.to_file(str(tmp_path / "rev"))
    r2 = ReverseKmerIndex.from_file(str(tmp_path / "rev"))
    for name in ReverseKmerIndex.properties:
        assert np.array_equal(getattr(r, name), getattr(r2, name))
    with pytest.raises(IndexError):
        r.get_node_kmers_and_ref_positions(9)


@pytest.mark.parametrize("form", ["rows", "pairs"])
@pytest.mark.parametrize("n,n_nodes,seed", [(1, 1, 0), (5000, 70, 1), (300000, 100000, 2), (200000, 3, 3), (3000, 5000000, 4),
                                            (2000000, 1500, 5)])
def test_reverse_index_random_vs_numpy(n, n_nodes, seed, form, monkeypatch):
    # form: the row-carrying build with key = node id (the default), and the pair-sorting form that stays behind it
    # (GKI_REVERSE_FORM is read by the library per call)
    monkeypatch.setenv("GKI_REVERSE_FORM", form)
    rng = np.random.default_rng(seed)
    nodes = rng.integers(0, n_nodes, size=n).astype(np.uint32)
    kmers = rng.integers(0, 4 ** 31, size=n, dtype=np.uint64)
    refs = rng.integers(0, 2 ** 40, size=n, dtype=np.uint64)
    r = ReverseKmerIndex.from_flat_kmers(FlatKmers(kmers, nodes, refs))
    index, counts, skm, srf = _reverse_index_numpy(nodes, kmers, refs)        # (200000, 3): counts wrap at 2^16
    assert np.array_equal(r.nodes_to_index_positions, index)
    assert np.array_equal(r.nodes_to_n_hashes, counts)
    assert np.array_equal(r.hashes, skm) and np.array_equal(r.ref_positions, srf)


@pytest.mark.parametrize("form", ["rows", "pairs"])
def test_reverse_index_refuses_a_node_id_beyond_n_nodes(form, monkeypatch):
    # include/gki.h: GKI_ERR_BAD_ARG in both forms, and nothing written outside the caller's n_nodes-sized outputs
    monkeypatch.setenv("GKI_REVERSE_FORM", form)
    rng = np.random.default_rng(9)
    n, n_nodes = 50000, 1000
    nodes = rng.integers(0, n_nodes, size=n).astype(np.uint32)
    nodes[12345] = n_nodes + 7
    d_nodes = _lib.DeviceArray.from_host(nodes)
    d_kmers = _lib.DeviceArray.from_host(rng.integers(0, 4 ** 31, size=n, dtype=np.uint64))
    d_refs = _lib.DeviceArray.from_host(rng.integers(0, 2 ** 40, size=n, dtype=np.uint64))
    guard = 4096                                              # elements behind the directory that must stay as they are
    index_pos = _lib.DeviceArray.from_host(np.full(n_nodes + guard, 0xABCDABCD, np.uint32))
    n_hashes = _lib.DeviceArray.from_host(np.full(n_nodes + guard, 0xABCD, np.uint16))
    out_kmers, out_refs = _lib.DeviceArray(n, np.uint64), _lib.DeviceArray(n, np.uint64)
    with pytest.raises(_lib.GkiError) as e:
        _lib.check(_lib.load().gki_reverse_index_build(d_nodes.ptr, d_kmers.ptr, d_refs.ptr, n, n_nodes, index_pos.ptr, n_hashes.ptr,
                                                       out_kmers.ptr, out_refs.ptr))
    assert e.value.code == 2
    assert (index_pos.to_host()[n_nodes:] == 0xABCDABCD).all() and (n_hashes.to_host()[n_nodes:] == 0xABCD).all()


def test_reverse_index_from_device_flat_kmers():
    from graph_kmer_index_amd import DeviceFlatKmers
    rng = np.random.default_rng(5)
    n = 40000
    flat = FlatKmers(rng.integers(0, 4 ** 31, size=n, dtype=np.uint64), rng.integers(0, 999, size=n).astype(np.uint32),
                     rng.integers(0, 10 ** 9, size=n, dtype=np.uint64), np.ones(n, np.float32))
    a = ReverseKmerIndex.from_flat_kmers(DeviceFlatKmers.from_flat_kmers(flat))
    b = ReverseKmerIndex.from_flat_kmers(flat)
    for name in ReverseKmerIndex.properties:
        assert np.array_equal(getattr(a, name), getattr(b, name))


# ------------------------------------------------------------------ probe table + fused read mapping
def _oracle_node_counts(index, queries, n_nodes, max_hits):
    exp = np.zeros(n_nodes, np.int64)
    hits = 0
    for q in queries:
        nodes = oracle.index_get(index, int(q), max_hits)[0]
        if nodes is not None:
            np.add.at(exp, np.asarray(nodes, np.int64), 1)
            hits += len(nodes)
    return exp, hits


@pytest.mark.parametrize("modulo,n,pool_size,skip_freq", [(100003, 60000, 5000, False), (3, 250000, 40, False),
                                                         (257, 30000, 900, True), (452930477, 20000, 20000, False)])
def test_probe_table_count_nodes_vs_oracle(modulo, n, pool_size, skip_freq):
    # (3, 250000): three buckets of > 65535 records (saturated directory count, all-ones fingerprint set)
    rng = np.random.default_rng(modulo % 1000)
    pool = rng.integers(0, 4 ** 31, size=pool_size, dtype=np.int64)
    kmers = pool[rng.integers(0, pool_size, size=n)]
    nodes = rng.integers(0, 2000, size=n).astype(np.uint32)
    refs = rng.integers(0, 50, size=n).astype(np.uint64)
    af = np.ones(n, np.float32)
    idx = CollisionFreeKmerIndex.from_flat_kmers(FlatKmers(kmers, nodes, refs, af), modulo=modulo, skip_frequencies=skip_freq)
    orc = oracle.index_build(kmers, nodes, refs, af, modulo=modulo, skip_frequencies=skip_freq)
    queries = np.concatenate([pool[:300], pool[:100], rng.integers(0, 4 ** 31, size=300, dtype=np.int64)]).astype(np.uint64)
    dev = idx._device_index()
    for max_hits in (2 ** 62, 10, 2):
        exp, exp_hits = _oracle_node_counts(orc, queries[:120] if n > 100000 else queries, 2000, max_hits)
        qs = queries[:120] if n > 100000 else queries
        got, hits = dev.count_nodes(qs, 2000, max_hits=max_hits, return_hits=True)
        ref_layout = dev.count_nodes(qs, 2000, max_hits=max_hits, use_probe_table=False)
        assert hits == exp_hits
        assert np.array_equal(got.to_host(2000), exp)
        assert np.array_equal(ref_layout.to_host(2000), exp)
    # nodes beyond n_counts are dropped, counts accumulate across calls
    small = dev.count_nodes(queries, 100, max_hits=2 ** 62)
    dev.count_nodes(queries, 100, max_hits=2 ** 62, counts=small)
    full, _ = _oracle_node_counts(orc, queries[:120], 2000, 2 ** 62) if n > 100000 else _oracle_node_counts(orc, queries, 2000, 2 ** 62)
    if n <= 100000:
        assert np.array_equal(small.to_host(100), 2 * full[:100])


@pytest.mark.parametrize("k", [5, 31])
def test_map_reads_fused_vs_oracle(k):
    rng = np.random.default_rng(40 + k)
    genome = "".join("ACGT"[i] for i in rng.integers(0, 4, size=6000))
    codes = oracle.letter_sequence_to_numeric(genome)
    hashes = oracle.hash_sequence(codes, k).astype(np.int64)
    n = len(hashes)
    nodes = (np.arange(n) // 50).astype(np.uint32)
    n_nodes = int(nodes.max()) + 1
    refs = np.arange(n, dtype=np.uint64)
    af = np.ones(n, np.float32)
    modulo = 10007
    idx = CollisionFreeKmerIndex.from_flat_kmers(FlatKmers(hashes, nodes, refs, af), modulo=modulo)
    orc = oracle.index_build(hashes, nodes, refs, af, modulo=modulo)
    comp = str.maketrans("ACGTacgt", "TGCAtgca")
    reads = []
    for _ in range(150):
        a = int(rng.integers(0, len(genome) - 200))
        r = list(genome[a:a + int(rng.integers(1, 200))])
        for p in rng.integers(0, len(r), size=rng.integers(0, 3)):
            r[p] = "NnacgtRY"[int(rng.integers(0, 8))]
        s = "".join(r)
        reads.append(s.translate(comp)[::-1] if rng.random() < 0.5 else s)
    reads += ["", "A" * (k - 1), genome[100:100 + k], genome[:130].lower()]
    for max_hits in (2 ** 62, 3):
        fwd = np.concatenate([oracle.read_kmers(r, k) for r in reads])
        rev = np.concatenate([oracle.read_kmers(r.translate(comp)[::-1], k) for r in reads])
        e_f, h_f = _oracle_node_counts(orc, fwd, n_nodes, max_hits)
        e_r, h_r = _oracle_node_counts(orc, rev, n_nodes, max_hits)
        assert h_f > 0 and h_r > 0
        assert np.array_equal(idx.map_reads(reads, k, n_nodes, max_hits=max_hits, include_reverse_complement=False), e_f)
        assert np.array_equal(idx.map_reads(reads, k, n_nodes, max_hits=max_hits), e_f + e_r)
        enc = "".join(reads).encode()
        start = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.int64)
        c, n_kmers, n_hits = idx._device_index().count_nodes_from_reads(np.frombuffer(enc, np.uint8), start, k, n_nodes,
                                                                       strands=2, max_hits=max_hits)
        assert np.array_equal(c.to_host(n_nodes), e_r) and n_kmers == len(rev) and n_hits == h_r


def test_contains_and_compaction_primitives():
    from graph_kmer_index_amd import DeviceFlatKmers
    rng = np.random.default_rng(8)
    n = 300000
    kmers = rng.integers(0, 4 ** 31, size=n, dtype=np.uint64)
    flat = FlatKmers(kmers, rng.integers(0, 99, size=n).astype(np.uint32), np.arange(n, dtype=np.uint64),
                     rng.random(n).astype(np.float32))
    idx = CollisionFreeKmerIndex.from_flat_kmers(FlatKmers(kmers[::3], np.zeros(n // 3, np.uint32), np.zeros(n // 3, np.uint64),
                                                           np.ones(n // 3, np.float32)), modulo=50021)
    member = np.isin(kmers, kmers[::3])
    assert np.array_equal(idx.has_kmers(kmers), member)
    assert not idx.has_kmers(rng.integers(0, 4 ** 31, size=1000, dtype=np.uint64)).any()
    d = DeviceFlatKmers.from_flat_kmers(flat)
    flags = idx._device_index().contains(d.hashes)
    assert flags.checksum(n)[0] == int(member.sum())
    kept = d.compacted(flags).to_flat_kmers()
    for name in ("_hashes", "_nodes", "_ref_offsets", "_allele_frequencies"):
        assert np.array_equal(getattr(kept, name), getattr(flat, name)[member]), name


def test_counter_kmer_index_node_counts():
    # collision_free_kmer_index.py:14-40 restated with NumPy: counter[kmer] = occurrences among the counted k-mers,
    # node counts = bincount(nodes, weights=counter[kmers])
    from graph_kmer_index_amd import CounterKmerIndex
    rng = np.random.default_rng(17)
    n = 40000
    pool = rng.integers(0, 4 ** 31, size=6000, dtype=np.int64)
    kmers = pool[rng.integers(0, len(pool), size=n)]
    nodes = rng.integers(0, 700, size=n).astype(np.uint32)
    idx = CollisionFreeKmerIndex.from_flat_kmers(FlatKmers(kmers, nodes, np.zeros(n, np.uint64), np.ones(n, np.float32)),
                                                 modulo=30011)
    counter = CounterKmerIndex.from_kmer_index(idx)
    batches = [np.concatenate([pool[rng.integers(0, len(pool), size=5000)], rng.integers(0, 4 ** 31, size=800, dtype=np.int64)])
               for _ in range(3)]

    def expect(seen):
        uniq, cnt = np.unique(seen, return_counts=True)
        per_kmer = dict(zip(uniq.tolist(), cnt.tolist()))
        w = np.array([per_kmer.get(int(km), 0) for km in idx._kmers.astype(np.int64)], dtype=np.float64)
        return np.bincount(idx._nodes.astype(np.int64), w, minlength=900)

    counter.count_kmers(batches[0])
    counter.count_kmers(batches[1])
    got = counter.get_node_counts(min_nodes=900)
    assert got.dtype == np.float64 and len(got) == 900
    assert np.array_equal(got, expect(np.concatenate(batches[:2])))
    counter.count_kmers(batches[2], update_counter=False)           # resets first
    assert np.array_equal(counter.get_node_counts(900), expect(batches[2]))
    bare = CounterKmerIndex(idx._kmers, idx._nodes, None, modulo=30011)
    bare.count_kmers(batches[0])
    assert np.array_equal(bare.get_node_counts(900), expect(batches[0]))


def test_lookup_positions_probe_table_equals_reference_layout():
    rng = np.random.default_rng(23)
    n = 80000
    pool = rng.integers(0, 4 ** 31, size=9000, dtype=np.uint64)
    kmers = pool[rng.integers(0, len(pool), size=n)]
    idx = CollisionFreeKmerIndex.from_flat_kmers(FlatKmers(kmers, rng.integers(0, 500, size=n).astype(np.uint32),
                                                           rng.integers(0, 40, size=n).astype(np.uint64), np.ones(n, np.float32)),
                                                 modulo=12007)
    queries = np.concatenate([pool[:3000], rng.integers(0, 4 ** 31, size=2000, dtype=np.uint64), pool[:50]])
    dev = idx._device_index()
    for max_hits in (2 ** 62, 10, 3, 1):
        a = dev.lookup_positions(queries, max_hits, use_probe_table=True)
        b = dev.lookup_positions(queries, max_hits, use_probe_table=False)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        assert len(a[1]) > 0


def test_without_singletons_on_device_and_skip_singletons_build():
    from graph_kmer_index_amd import DeviceFlatKmers
    rng = np.random.default_rng(29)
    for n, distinct in [(1, 1), (50000, 20000), (80000, 300), (30000, 30000)]:
        pool = rng.integers(0, 4 ** 31, size=distinct, dtype=np.uint64)
        flat = FlatKmers(pool[rng.integers(0, distinct, size=n)], rng.integers(0, 999, size=n).astype(np.uint32),
                         np.arange(n, dtype=np.uint64), rng.random(n).astype(np.float32))
        want = flat.get_new_without_singletons()
        keep = oracle.without_singletons(flat._hashes)
        assert np.array_equal(want._hashes, flat._hashes[keep])                 # host method == oracle
        d = DeviceFlatKmers.from_flat_kmers(flat)
        got = d.get_new_without_singletons().to_flat_kmers()
        for name in ("_hashes", "_nodes", "_ref_offsets", "_allele_frequencies"):
            assert np.array_equal(getattr(got, name), getattr(want, name)), (n, distinct, name)
        if n > 1:
            a = CollisionFreeKmerIndex.from_flat_kmers(d, modulo=10007, skip_singletons=True)
            b = oracle.index_build(flat._hashes, flat._nodes, flat._ref_offsets, flat._allele_frequencies, modulo=10007,
                                   skip_singletons=True)
            assert np.array_equal(a._kmers, b["_kmers"]) and np.array_equal(a._frequencies, b["_frequencies"])
            assert np.array_equal(a._hashes_to_index, b["_hashes_to_index"])


def test_device_flat_kmers_reverse_complement_and_concatenation():
    from graph_kmer_index_amd import DeviceFlatKmers
    rng = np.random.default_rng(30)
    n = 20000
    flat = FlatKmers(rng.integers(0, 4 ** 31, size=n, dtype=np.uint64), rng.integers(0, 99, size=n).astype(np.uint32),
                     np.arange(n, dtype=np.uint64), rng.random(n).astype(np.float32))
    d = DeviceFlatKmers.from_flat_kmers(flat)
    both = DeviceFlatKmers.from_multiple_flat_kmers([d, d.get_reverse_complement_flat_kmers(31)]).to_flat_kmers()
    want = FlatKmers.from_multiple_flat_kmers([flat, flat.get_reverse_complement_flat_kmers(31)])
    assert np.array_equal(want._hashes[n:], oracle.reverse_complement(flat._hashes, 31))
    for name in ("_hashes", "_nodes", "_ref_offsets", "_allele_frequencies"):
        assert np.array_equal(getattr(both, name), getattr(want, name)), name


def test_scalar_getters_one_launch_and_batched_frequency_helpers():
    """get / get_frequency through gki_index_get_small (one launch, pinned staging) and the batched
    FlatKmers.sum_of_kmer_frequencies / maximum_kmer_frequency, against the oracle's get."""
    import time
    from graph_kmer_index_amd import CollisionFreeKmerIndex, FlatKmers
    rng = np.random.default_rng(12)
    pool = rng.integers(0, 4 ** 31, size=3000, dtype=np.int64)
    kmers = pool[rng.integers(0, len(pool), size=20000)]
    kmers[:1500] = pool[0]                                        # one k-mer with more than 1024 records
    nodes = rng.integers(0, 900, size=len(kmers)).astype(np.uint32)
    refs = rng.integers(0, 40, size=len(kmers)).astype(np.uint64)
    af = rng.random(len(kmers)).astype(np.float32)
    flat = FlatKmers(kmers, nodes, refs, af)
    index = CollisionFreeKmerIndex.from_flat_kmers(flat, modulo=4001)
    ref = oracle.index_build(kmers.astype(np.uint64), nodes, refs, af, modulo=4001)
    queries = [int(x) for x in pool[:200]] + [int(x) for x in rng.integers(0, 4 ** 31, size=50)]
    for mh in (10, 10 ** 15):
        for q in queries:
            got, want = index.get(q, max_hits=mh), oracle.index_get(ref, q, mh)
            if want[0] is None:
                assert got == (None, None, None, None)
                continue
            for a, b in zip(got, want):
                assert np.array_equal(a, b)
    from graph_kmer_index_amd.kmer_hashing import kmer_hash_to_reverse_complement_hash

    def oracle_freq(q):
        f = 0
        for x in (q, int(kmer_hash_to_reverse_complement_hash(q, 31))):
            r = oracle.index_get(ref, x, 10 ** 15)
            f += 0 if r[0] is None else int(r[2][0])
        return f
    want_f = [oracle_freq(q) for q in queries]
    assert [index.get_frequency(q) for q in queries] == want_f
    assert index.get_frequencies(np.array(queries, dtype=np.int64)).tolist() == want_f
    sub = FlatKmers(np.array(queries, dtype=np.int64), np.zeros(len(queries), np.uint32))
    assert sub.sum_of_kmer_frequencies(index) == sum(max(1, f) for f in want_f)
    assert sub.maximum_kmer_frequency(index) == max(want_f)
    t = time.perf_counter()
    for q in queries * 8:
        index.get(q)
    rate = len(queries) * 8 / (time.perf_counter() - t)
    print("scalar get(): %.0f calls/s" % rate)
    assert rate > 2000
