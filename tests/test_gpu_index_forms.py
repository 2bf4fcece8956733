"""The two forms of the CollisionFreeKmerIndex build (collision_free_kmer_index.py:423-467, set_frequencies :267-293):
the row-carrying form (csrc/gki_index_rows.hip: partition passes + in-LDS finish) and the pair-sorting form
(csrc/gki_index.hip) must agree element by element with the oracle's stable build and with each other, on every shape
the row-carrying form treats differently: no / one / two / three partition passes, groups finished in LDS, groups too
large for LDS (streamed by one workgroup), buckets past the per-lane frequency path, a bucket range, a wanted
permutation, and the hand-over to the pair-sorting form for an index that is one giant bucket."""
import numpy as np
import pytest

from graph_kmer_index_amd import _lib
from graph_kmer_index_amd.flat_kmers import FlatKmers, DeviceFlatKmers
from graph_kmer_index_amd.collision_free_kmer_index import DeviceIndex, bucket_range
from oracle import oracle

pytestmark = pytest.mark.gpu

COLS = (("_hashes_to_index", "hashes_to_index"), ("_n_kmers", "n_kmers"), ("_kmers", "kmers"), ("_nodes", "nodes"),
        ("_ref_offsets", "ref_offsets"), ("_allele_frequencies", "allele_frequencies"), ("_frequencies", "frequencies"))


def _records(n, n_distinct, seed, heavy=0):
    """n records over a pool of n_distinct k-mers; `heavy` of them repeat one k-mer (a bucket of that size)."""
    rng = np.random.default_rng(seed)
    pool = rng.integers(0, 4 ** 31, size=n_distinct, dtype=np.uint64)
    kmers = pool[rng.integers(0, n_distinct, size=n)]
    if heavy:
        kmers[rng.choice(n, size=heavy, replace=False)] = pool[0]
    nodes = rng.integers(0, 1 << 24, size=n).astype(np.uint32)
    refs = (kmers % np.uint64(50)) + rng.integers(0, 30, size=n).astype(np.uint64)
    af = rng.uniform(0, 1, size=n).astype(np.float32)
    return kmers, nodes, refs, af


def _check(dev, o, n, perm=None):
    for name, attr in COLS:
        got = getattr(dev, attr).to_host(len(o[name]) if name in ("_hashes_to_index", "_n_kmers") else n)
        assert np.array_equal(got, o[name]), name
    if perm is not None:
        assert np.array_equal(dev.permutation.to_host(n), perm)


@pytest.mark.parametrize("n,modulo,n_distinct,heavy", [
    (7, 4, 5, 0),                       # one group, no real partition pass
    (5000, 1009, 3000, 0),              # 10-bit key: all of it inside LDS
    (300000, 65537, 5000, 0),           # one partition pass; buckets of ~60 records (large-bucket frequencies)
    (200000, 452930477, 150000, 0),     # default modulo, sparse: two passes of 9 bits, nearly empty groups
    (400000, 1 << 21, 300000, 0),       # two passes
    (250000, 4294967291, 200000, 0),    # 32-bit key: three passes
    (50000, 7, 40, 0),                  # seven buckets of ~7000 records: groups streamed by one workgroup each
    (600000, 1000003, 400000, 30000),   # a 30 000-record bucket inside an ordinary index: one large group
    (70000, 1000003, 3, 0),             # three giant buckets
    (3000, 2, 1, 0),                    # a single k-mer
    (800000, 100003, 300000, 0),        # 8 records per bucket, k-mers repeated 2-3 times: the finish ranks by ballots (dense
                                        # slices of a whole-genome index), frequencies through the several-rows-per-k-mer path
    (700000, 100003, 5000000, 0),       # 7 per bucket, nearly every k-mer once (the bench's full_index slices in small)
    (300000, 100003, 100000, 0),        # 3 per bucket
    (1000000, 49999, 700000, 0),        # 20 per bucket: most buckets near the per-lane frequency limit, some beyond it
])
@pytest.mark.parametrize("skip_frequencies", [False, True])
def test_both_forms_equal_the_oracle(n, modulo, n_distinct, heavy, skip_frequencies):
    kmers, nodes, refs, af = _records(n, n_distinct, seed=n + modulo % 1000, heavy=heavy)
    o = oracle.index_build(kmers, nodes, refs, af, modulo=modulo, skip_frequencies=skip_frequencies)
    d = DeviceFlatKmers.from_flat_kmers(FlatKmers(kmers, nodes, refs, af))
    order = np.argsort(kmers % np.uint64(modulo), kind="stable").astype(np.uint32)
    for pairs in (False, True):
        for want_perm in (False, True):
            dev = DeviceIndex.build(d, modulo, skip_frequencies, want_permutation=want_perm, pairs_form=pairs)
            _check(dev, o, n, order if want_perm else None)
            dev.free()
    d.free()


def test_bucket_range_slices_and_out_of_range_record():
    n, modulo, world = 300000, 999983, 3
    kmers, nodes, refs, af = _records(n, 200000, seed=5)
    full = oracle.index_build(kmers, nodes, refs, af, modulo=modulo)
    buckets = kmers % np.uint64(modulo)
    for r in range(world):
        lo, hi = bucket_range(modulo, world, r)
        sel = (buckets >= lo) & (buckets < hi)
        d = DeviceFlatKmers.from_flat_kmers(FlatKmers(kmers[sel], nodes[sel], refs[sel], af[sel]))
        for pairs in (False, True):
            dev = DeviceIndex.build(d, modulo, bucket_begin=lo, n_buckets=hi - lo, pairs_form=pairs)
            first = int(np.searchsorted(np.sort(buckets, kind="stable"), lo))
            m = int(sel.sum())
            assert np.array_equal(dev.kmers.to_host(m), full["_kmers"][first:first + m])
            assert np.array_equal(dev.frequencies.to_host(m), full["_frequencies"][first:first + m])
            nk = dev.n_kmers.to_host()
            assert np.array_equal(nk, full["_n_kmers"][lo:hi])
            h2i = dev.hashes_to_index.to_host()
            assert np.array_equal(h2i[nk > 0], full["_hashes_to_index"][lo:hi][nk > 0] - first)
            assert not h2i[nk == 0].any()
            dev.free()
        d.free()
    # a record outside the range is refused by both forms
    lo, hi = bucket_range(modulo, world, 1)
    d = DeviceFlatKmers.from_flat_kmers(FlatKmers(kmers[:1000], nodes[:1000], refs[:1000], af[:1000]))
    for pairs in (False, True):
        with pytest.raises(_lib.GkiError):
            DeviceIndex.build(d, modulo, bucket_begin=lo, n_buckets=hi - lo, pairs_form=pairs)
    d.free()


def test_giant_bucket_hands_over_to_the_pair_sorting_form():
    # one k-mer 4.3 million times: a group past what one workgroup should stream; gki_index_build still answers, through
    # the pair-sorting form, and equals the oracle.  Frequencies are skipped: the oracle's count of distinct ref offsets
    # is quadratic in the bucket (as the reference's loop is), large buckets with frequencies are covered at 30 000 above
    n = (1 << 22) + 100000
    rng = np.random.default_rng(3)
    kmers = np.full(n, 123456789123, dtype=np.uint64)
    kmers[::1000] = rng.integers(0, 4 ** 31, size=len(kmers[::1000]), dtype=np.uint64)
    nodes = rng.integers(0, 1 << 20, size=n).astype(np.uint32)
    refs = rng.integers(0, 40000, size=n).astype(np.uint64)
    af = np.ones(n, np.float32)
    o = oracle.index_build(kmers, nodes, refs, af, modulo=1000003, skip_frequencies=True)
    d = DeviceFlatKmers.from_flat_kmers(FlatKmers(kmers, nodes, refs, af))
    dev = DeviceIndex.build(d, 1000003, skip_frequencies=True)
    _check(dev, o, n)
    dev.free()
    d.free()


def test_a_million_records_default_modulo_device_resident():
    # the shape of the bench's index_build record at 1/300 of its size: dense enough that groups hold ~1400 rows
    n, modulo = 1 << 20, 1530013
    kmers, nodes, refs, af = _records(n, 900000, seed=11)
    o = oracle.index_build(kmers, nodes, refs, af, modulo=modulo)
    d = DeviceFlatKmers.from_flat_kmers(FlatKmers(kmers, nodes, refs, af))
    dev = DeviceIndex.build(d, modulo)
    _check(dev, o, n)
    dev.free()
    d.free()


def test_wave_prefix_sum_on_the_dpp_path_equals_the_shuffle_form():
    # csrc/gki_common.h gki_wave_incl_sum: used by the finder's expansion and by both index builds
    import ctypes as C
    bad = C.c_int64(-1)
    _lib.check(_lib.load().gki_selftest_wave_scan(C.byref(bad)))
    assert bad.value == 0


def _group_ids(buckets, lo, hi, g):
    """Group of every bucket of the part [lo, hi) under the rule of gki_partition_by_bucket_range_grouped."""
    kb = int(hi - lo - 1).bit_length()
    return (buckets - np.uint64(lo)) >> np.uint64(max(0, kb - g))


@pytest.mark.parametrize("n,modulo,n_parts,g,n_distinct", [
    (600000, 100003, 8, 7, 250000),     # 6 per bucket (ballot ranking, 4096-row finish groups), k-mers repeated: 1024 digits
    (500000, 452930477, 8, 7, 400000),  # default modulo, sparse: one partition pass inside the groups, nearly empty finish groups
    (300000, 65537, 3, 4, 200000),      # parts that are no power of two, few groups
    (200000, 1009, 4, 6, 900),          # more groups than buckets per part: groups of single buckets, 200 records per bucket
    (50000, 999983, 1, 10, 40000),      # one part cut into 1024 groups
    (7, 4, 2, 2, 5),                    # seven records
])
def test_grouped_partition_and_grouped_build_equal_the_oracle(n, modulo, n_parts, g, n_distinct):
    """The single-GPU whole-genome build: the partition pass groups every part's records by the top bits of their key
    (gki_partition_by_bucket_range_grouped) and the slice builds start from that grouping with one pass less
    (gki_index_build_range_grouped).  The partition must be the stable partition by (part, group); every slice must equal
    the oracle's stable build cut at the slice's bucket range, element by element."""
    from graph_kmer_index_amd.collision_free_kmer_index import (partition_by_bucket_range, partition_rows_by_bucket_range,
                                                                PartitionedDeviceIndex)
    kmers, nodes, refs, af = _records(n, n_distinct, seed=n + g)
    full = oracle.index_build(kmers, nodes, refs, af, modulo=modulo)
    buckets = kmers % np.uint64(modulo)
    d = DeviceFlatKmers.from_flat_kmers(FlatKmers(kmers, nodes, refs, af))
    part, start = partition_by_bucket_range(d, modulo, n_parts, group_bits=g, max_rows_per_pass=150000)
    begins = np.array([bucket_range(modulo, n_parts, p)[0] for p in range(n_parts)], dtype=np.uint64)
    owner = np.searchsorted(begins, buckets, side="right") - 1
    digit = np.zeros(n, dtype=np.int64)
    for p in range(n_parts):
        lo, hi = bucket_range(modulo, n_parts, p)
        sel = owner == p
        digit[sel] = (p << g) | _group_ids(buckets[sel], lo, hi, g).astype(np.int64)
    order = np.argsort(digit, kind="stable")
    got = part.to_flat_kmers()
    for name, col in (("_hashes", kmers), ("_nodes", nodes), ("_ref_offsets", refs), ("_allele_frequencies", af)):
        assert np.array_equal(getattr(got, name), col[order]), name
    assert start == np.concatenate([[0], np.cumsum(np.bincount(digit, minlength=n_parts << g))]).tolist()
    # the same partition with the records left as rows: row = (k-mer, ref offset, node | allele frequency bits << 32), key =
    # the bucket's offset in its part
    rows, start_r = partition_rows_by_bucket_range(d, modulo, n_parts, group_bits=g, max_rows_per_pass=150000)
    assert start_r == start
    r = rows.rows.to_host(3 * n).reshape(n, 3)
    assert np.array_equal(r[:, 0], kmers[order]) and np.array_equal(r[:, 1], refs[order])
    assert np.array_equal(r[:, 2], nodes[order].astype(np.uint64) | (af[order].view(np.uint32).astype(np.uint64) << np.uint64(32)))
    assert np.array_equal(rows.keys.to_host(n), (buckets - begins[owner])[order].astype(np.uint32))
    sorted_buckets = np.sort(buckets, kind="stable")
    for p in range(n_parts):
        lo, hi = bucket_range(modulo, n_parts, p)
        for skip, src in ((False, part), (True, part), (False, rows), (True, rows)):
            dev = PartitionedDeviceIndex.build_slice(src, start, modulo, n_parts, p, g, skip_frequencies=skip)
            first = int(np.searchsorted(sorted_buckets, lo))
            m = dev.n
            assert m == int((owner == p).sum())
            for name, attr in (("_kmers", "kmers"), ("_nodes", "nodes"), ("_ref_offsets", "ref_offsets"),
                               ("_allele_frequencies", "allele_frequencies")):
                assert np.array_equal(getattr(dev, attr).to_host(m), full[name][first:first + m]), (p, name)
            if not skip:
                assert np.array_equal(dev.frequencies.to_host(m), full["_frequencies"][first:first + m]), p
            nk = dev.n_kmers.to_host()
            assert np.array_equal(nk, full["_n_kmers"][lo:hi])
            h2i = dev.hashes_to_index.to_host()
            assert np.array_equal(h2i[nk > 0], full["_hashes_to_index"][lo:hi][nk > 0] - first)
            assert not h2i[nk == 0].any()
            dev.free()
    # rows without a grouping (group_bits = 0): the build from rows sorts everything itself
    if n_parts > 1:
        plain_rows, plain_start = partition_rows_by_bucket_range(d, modulo, n_parts)
        dev = PartitionedDeviceIndex.build_slice(plain_rows, plain_start, modulo, n_parts, 1, 0)
        lo, hi = bucket_range(modulo, n_parts, 1)
        first = int(np.searchsorted(sorted_buckets, lo))
        assert np.array_equal(dev.kmers.to_host(dev.n), full["_kmers"][first:first + dev.n])
        assert np.array_equal(dev.frequencies.to_host(dev.n), full["_frequencies"][first:first + dev.n])
        dev.free()
        plain_rows.free()
    rows.free()
    part.free()
    d.free()


def test_partitioned_index_grouped_and_plain_count_the_same_nodes():
    from graph_kmer_index_amd.collision_free_kmer_index import PartitionedDeviceIndex
    n, modulo = 400000, 200003
    kmers, nodes, refs, af = _records(n, 150000, seed=21)
    nodes = nodes % np.uint32(5000)
    d = DeviceFlatKmers.from_flat_kmers(FlatKmers(kmers, nodes, refs, af))
    queries = np.concatenate([kmers[::7], np.random.default_rng(3).integers(0, 4 ** 31, size=5000, dtype=np.uint64)])
    want = None
    for grouped in (True, False):
        idx = PartitionedDeviceIndex.build(d, modulo, n_parts=8, grouped=grouped)
        assert idx.n == n
        got = idx.count_nodes(queries, 5000, max_hits=2 ** 40).to_host()
        if want is None:
            want = got
        assert got.sum() > 0 and np.array_equal(got, want)
        idx.free()
    d.free()


def test_partitioned_index_with_a_giant_bucket_falls_back_to_columns():
    """A slice whose records sit in a handful of buckets (4.3 million copies of one k-mer) is outside the row-carrying
    build's domain; the build from columns hands it to the pair-sorting form, the build from rows cannot:
    PartitionedDeviceIndex.build starts over through columns and still answers."""
    from graph_kmer_index_amd.collision_free_kmer_index import PartitionedDeviceIndex
    n = (1 << 22) + 50000
    rng = np.random.default_rng(8)
    kmers = np.full(n, 987654321987, dtype=np.uint64)
    kmers[::500] = rng.integers(0, 4 ** 31, size=len(kmers[::500]), dtype=np.uint64)
    nodes = rng.integers(0, 3000, size=n).astype(np.uint32)
    refs = rng.integers(0, 40000, size=n).astype(np.uint64)
    d = DeviceFlatKmers.from_flat_kmers(FlatKmers(kmers, nodes, refs, np.ones(n, np.float32)))
    idx = PartitionedDeviceIndex.build(d, 1000003, n_parts=4, skip_frequencies=True)
    assert idx.n == n
    queries = np.unique(np.concatenate([kmers[::500][:2000], [np.uint64(987654321987)]]))
    got = idx.count_nodes(queries, 3000, max_hits=2 ** 40).to_host()
    want = np.zeros(3000, dtype=np.int64)
    sel = np.isin(kmers, queries)
    np.add.at(want, nodes[sel], 1)
    assert np.array_equal(got.astype(np.int64), want)
    idx.free()
    d.free()
