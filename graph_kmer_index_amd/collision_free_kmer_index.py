"""CollisionFreeKmerIndex with the reference's attributes, constructor and methods
(collision_free_kmer_index.py:163-490); `from_flat_kmers` and the batched getters run on MI355X.

The attribute arrays stay NumPy-readable (`to_file`/`from_file` use the reference's .npz keys, and
kmer_mapper reads `_hashes_to_index/_n_kmers/_nodes/_kmers/_modulo` directly, :210-216); a device
copy for batched lookups is created lazily and cached.
"""
import ctypes as C
import gc
import logging
import numpy as np

from . import _lib
from .flat_kmers import FlatKmers, DeviceFlatKmers
from .kmer_hashing import kmer_hash_to_reverse_complement_hash


class DeviceIndex:
    """The seven index arrays resident in HBM."""

    def __init__(self, n, modulo, hashes_to_index, n_kmers, kmers, nodes, ref_offsets, af, frequencies,
                 bucket_begin=0, n_buckets=None):
        self.n, self.modulo = int(n), int(modulo)
        # a bucket-range slice of a partitioned index holds the buckets [bucket_begin, bucket_begin + n_buckets) only
        self.bucket_begin, self.n_buckets = int(bucket_begin), int(modulo if n_buckets is None else n_buckets)
        self.permutation = None
        self._probe = None
        self.hashes_to_index, self.n_kmers, self.kmers, self.nodes = hashes_to_index, n_kmers, kmers, nodes
        self.ref_offsets, self.allele_frequencies, self.frequencies = ref_offsets, af, frequencies

    def view(self):
        return _lib.IndexView(self.hashes_to_index.ptr, self.n_kmers.ptr, self.kmers.ptr, self.nodes.ptr,
                              self.ref_offsets.ptr, self.frequencies.ptr, self.allele_frequencies.ptr,
                              self.modulo, self.n, self.bucket_begin, self.n_buckets)

    @classmethod
    def build(cls, dflat, modulo=452930477, skip_frequencies=False, want_permutation=False, bucket_begin=0, n_buckets=None,
              pairs_form=False, group_start=None):
        """gki_index_build(_range) on device-resident FlatKmers columns.  With a bucket range, `dflat` must hold only
        records of that range (one slice of `partition_by_bucket_range`).  pairs_form=True runs the pair-sorting form of
        the build (gki_index_build_pairs) instead of the row-carrying one; the results are identical.
        group_start (2^g + 1 row numbers): the records arrive grouped by the top g bits of their key, as
        `partition_by_bucket_range(..., group_bits=g)` leaves the records of a part (gki_index_build_range_grouped)."""
        _lib.require_device()
        n = dflat.n
        na = max(n, 1)
        nb = int(modulo if n_buckets is None else n_buckets)
        out = cls(n, modulo, _lib.DeviceArray(nb, np.int32), _lib.DeviceArray(nb, np.uint32),
                  _lib.DeviceArray(na, np.uint64), _lib.DeviceArray(na, np.uint32), _lib.DeviceArray(na, np.uint64),
                  _lib.DeviceArray(na, np.float32), _lib.DeviceArray(na, np.uint16), bucket_begin, nb)
        perm = _lib.DeviceArray(na, np.uint32) if want_permutation else None
        out.permutation = perm
        head = (dflat.hashes.ptr, dflat.nodes.ptr, dflat.ref_offsets.ptr, dflat.allele_frequencies.ptr, n, int(modulo),
                int(bucket_begin), nb, int(bool(skip_frequencies)))
        tail = (out.hashes_to_index.ptr, out.n_kmers.ptr, out.kmers.ptr, out.nodes.ptr, out.ref_offsets.ptr,
                out.allele_frequencies.ptr, out.frequencies.ptr, None if perm is None else perm.ptr)
        if group_start is not None and not pairs_form:
            group_bits = (len(group_start) - 1).bit_length() - 1
            assert len(group_start) == (1 << group_bits) + 1
            bounds = (C.c_int64 * len(group_start))(*[int(x) for x in group_start])
            _lib.check(_lib.load().gki_index_build_range_grouped(*head, group_bits, bounds, *tail))
        else:
            fn = _lib.load().gki_index_build_pairs if pairs_form else _lib.load().gki_index_build_range
            _lib.check(fn(*head, *tail))
        return out

    @classmethod
    def build_from_rows(cls, rows, keys, n, modulo, skip_frequencies=False, bucket_begin=0, n_buckets=None, group_start=None):
        """gki_index_build_range_from_rows: one part of `partition_rows_by_bucket_range` (views of its rows and keys)."""
        _lib.require_device()
        na = max(n, 1)
        nb = int(modulo if n_buckets is None else n_buckets)
        out = cls(n, modulo, _lib.DeviceArray(nb, np.int32), _lib.DeviceArray(nb, np.uint32),
                  _lib.DeviceArray(na, np.uint64), _lib.DeviceArray(na, np.uint32), _lib.DeviceArray(na, np.uint64),
                  _lib.DeviceArray(na, np.float32), _lib.DeviceArray(na, np.uint16), bucket_begin, nb)
        group_bits, bounds = 0, None
        if group_start is not None:
            group_bits = (len(group_start) - 1).bit_length() - 1
            assert len(group_start) == (1 << group_bits) + 1
            bounds = (C.c_int64 * len(group_start))(*[int(x) for x in group_start])
        _lib.check(_lib.load().gki_index_build_range_from_rows(
            rows.ptr, keys.ptr, int(n), int(modulo), int(bucket_begin), nb, int(bool(skip_frequencies)), group_bits, bounds,
            out.hashes_to_index.ptr, out.n_kmers.ptr, out.kmers.ptr, out.nodes.ptr, out.ref_offsets.ptr,
            out.allele_frequencies.ptr, out.frequencies.ptr))
        return out

    def lookup_positions(self, queries, max_hits=10, use_probe_table=True):
        """Batched CollisionFreeKmerIndex.get: (hit_start int64[q+1], position int64[hits] into the payload
        arrays, query index int64[hits]) as NumPy arrays.  use_probe_table=False probes the reference-layout arrays
        (gki_index_lookup_*) instead of the probe table (gki_probe_lookup_*)."""
        lib = _lib.load()
        q = np.ascontiguousarray(np.asarray(queries)).astype(np.uint64)
        nq = len(q)
        dq = _lib.DeviceArray.from_host(q if nq else np.zeros(1, np.uint64))
        hs = _lib.DeviceArray(nq + 1, np.int64)
        n_hits = C.c_int64(0)
        mh = int(min(max_hits, 2 ** 62))
        if use_probe_table:
            table = self.probe_table()
            _lib.check(lib.gki_probe_lookup_count(table, dq.ptr, nq, mh, hs.ptr, C.byref(n_hits)))
        else:
            view = self.view()
            _lib.check(lib.gki_index_lookup_count(C.byref(view), dq.ptr, nq, mh, hs.ptr, C.byref(n_hits)))
        m = n_hits.value
        pos, qi = _lib.DeviceArray(max(m, 1), np.int64), _lib.DeviceArray(max(m, 1), np.int64)
        if m:
            if use_probe_table:
                _lib.check(lib.gki_probe_lookup_emit(table, dq.ptr, nq, mh, hs.ptr, qi.ptr, pos.ptr))
            else:
                _lib.check(lib.gki_index_lookup_emit(C.byref(view), dq.ptr, nq, mh, hs.ptr, None, None, qi.ptr, None, None,
                                                     pos.ptr))
        out = (hs.to_host(), pos.to_host(m), qi.to_host(m))
        for b in (dq, hs, pos, qi):
            b.free()
        return out

    def get_small(self, queries, max_hits=10, capacity=1024):
        """CollisionFreeKmerIndex.get for up to 64 k-mers in one launch (gki_index_get_small): per query (number of
        hits, int64 positions of the first min(hits, capacity, 1024) of them in bucket order)."""
        q = np.ascontiguousarray(queries, dtype=np.uint64)
        n_hits = np.zeros(len(q), dtype=np.int64)
        pos = np.empty(max(len(q) * capacity, 1), dtype=np.int64)
        view = self.view()
        _lib.check(_lib.load().gki_index_get_small(C.byref(view), _lib.hptr(q), len(q), int(min(max_hits, 2 ** 62)),
                                                   _lib.hptr(n_hits), _lib.hptr(pos), capacity))
        return [(n, pos[i * capacity:i * capacity + min(n, capacity, 1024)].copy()) for i, n in enumerate(n_hits.tolist())]

    def probe_table(self):
        """The probe-table re-layout of this index in HBM (gki_probe_create), built on first use."""
        if self._probe is None:
            p = C.c_void_p()
            view = self.view()
            _lib.check(_lib.load().gki_probe_create(C.byref(view), C.byref(p)))
            self._probe = p
        return self._probe

    def contains(self, queries):
        """`kmer in index` for every query (gki_probe_contains).  queries: NumPy array or DeviceArray of uint64.
        Returns a DeviceArray uint8[q] of 0 / 1."""
        own = not isinstance(queries, _lib.DeviceArray)
        dq = _lib.DeviceArray.from_host(np.ascontiguousarray(np.asarray(queries)).astype(np.uint64)) if own else queries
        flags = _lib.DeviceArray(max(dq.n, 1), np.uint8)
        _lib.check(_lib.load().gki_probe_contains(self.probe_table(), dq.ptr, dq.n, flags.ptr))
        flags.n = dq.n
        if own:
            dq.free()
        return flags

    def count_nodes(self, queries, n_nodes, max_hits=10, counts=None, use_probe_table=True, return_hits=False):
        """Fused probe + node histogram.  queries: NumPy array or DeviceArray of uint64.  Returns a DeviceArray
        uint32[n_nodes] (accumulates into `counts` when given).  use_probe_table=False probes the reference-layout
        arrays directly (gki_index_count_nodes) instead of the probe table (gki_probe_count_nodes)."""
        lib = _lib.load()
        if counts is None:
            counts = _lib.DeviceArray(max(int(n_nodes), 1), np.uint32)
            counts.zero()
        own = not isinstance(queries, _lib.DeviceArray)
        dq = _lib.DeviceArray.from_host(np.ascontiguousarray(np.asarray(queries)).astype(np.uint64)) if own else queries
        mh = int(min(max_hits, 2 ** 62))
        n_hits = C.c_int64(-1)
        if use_probe_table:
            _lib.check(lib.gki_probe_count_nodes(self.probe_table(), dq.ptr, dq.n, mh, counts.ptr, int(n_nodes),
                                                 C.byref(n_hits)))
        else:
            view = self.view()
            _lib.check(lib.gki_index_count_nodes(C.byref(view), dq.ptr, dq.n, mh, counts.ptr, int(n_nodes)))
        if own:
            dq.free()
        return (counts, n_hits.value) if return_hits else counts

    def count_nodes_from_reads(self, letters, read_start, k, n_nodes, strands=3, max_hits=10, counts=None):
        """Read hashing fused with the probe (gki_probe_reads_count_nodes): letters = ASCII uint8 of all reads
        (NumPy or DeviceArray), read_start int64[n_reads+1]; strands bit 0 forward, bit 1 reverse complement.
        Returns (counts DeviceArray uint32[n_nodes], k-mers probed, hits)."""
        lib = _lib.load()
        if counts is None:
            counts = _lib.DeviceArray(max(int(n_nodes), 1), np.uint32)
            counts.zero()
        owned = []

        def dev(a, dtype):
            if isinstance(a, _lib.DeviceArray):
                return a
            a = np.ascontiguousarray(a, dtype=dtype)
            d = _lib.DeviceArray.from_host(a if a.size else np.zeros(1, dtype))
            owned.append(d)
            return d
        n_reads = len(read_start) - 1
        d_letters, d_start = dev(letters, np.uint8), dev(read_start, np.int64)
        n_kmers, n_hits = C.c_int64(0), C.c_int64(0)
        try:
            _lib.check(lib.gki_probe_reads_count_nodes(self.probe_table(), d_letters.ptr, d_start.ptr, n_reads, int(k),
                                                       int(strands), int(min(max_hits, 2 ** 62)), counts.ptr, int(n_nodes),
                                                       C.byref(n_kmers), C.byref(n_hits)))
        finally:
            for d in owned:
                d.free()
        return counts, n_kmers.value, n_hits.value

    def __del__(self):
        try:
            if self._probe is not None:
                _lib.load().gki_probe_destroy(self._probe)
                self._probe = None
        except Exception:
            pass

    def free(self):
        if self._probe is not None:
            _lib.load().gki_probe_destroy(self._probe)
            self._probe = None
        for a in (self.hashes_to_index, self.n_kmers, self.kmers, self.nodes, self.ref_offsets,
                  self.allele_frequencies, self.frequencies):
            a.free()


def bucket_range(modulo, n_parts, part):
    """Buckets [begin, end) owned by `part` (gki_partition_by_bucket_range)."""
    return modulo * part // n_parts, modulo * (part + 1) // n_parts


def partition_by_bucket_range(dflat, modulo, n_parts, out=None, max_rows_per_pass=0, group_bits=0):
    """Stable partition of device FlatKmers columns by owning part.  Returns (DeviceFlatKmers, part_start[n_parts+1]).
    `out`: columns to write into (at least dflat.n records, not overlapping dflat) instead of a fresh allocation.
    Any number of records (2^31 and more go through several passes, `max_rows_per_pass` at a time; 0: the default).
    group_bits = g > 0: the records of a part additionally leave grouped by the top g bits of their bucket's offset in
    the part (what `DeviceIndex.build(group_start=...)` takes one sort pass less for); the returned table then has
    (n_parts << g) + 1 entries, entry p << g | group = first row of that group of part p."""
    _lib.require_device()
    if out is None:
        out = DeviceFlatKmers.allocate(dflat.n)
    else:
        assert out.hashes.n >= dflat.n
        out.n = dflat.n
    start = (C.c_int64 * ((n_parts << group_bits) + 1))()
    _lib.check(_lib.load().gki_partition_by_bucket_range_grouped(
        dflat.hashes.ptr, dflat.nodes.ptr, dflat.ref_offsets.ptr, dflat.allele_frequencies.ptr, dflat.n, int(modulo),
        int(n_parts), int(group_bits), int(max_rows_per_pass), out.hashes.ptr, out.nodes.ptr, out.ref_offsets.ptr,
        out.allele_frequencies.ptr, start))
    return out, [int(x) for x in start]


class DeviceRows:
    """Records partitioned by bucket range and kept as 24-byte rows + 32-bit keys between the partition and the slice
    builds (gki_partition_rows_by_bucket_range -> gki_index_build_range_from_rows): the intermediate of the single-GPU
    whole-genome build, not a FlatKmers."""

    def __init__(self, n, rows=None, keys=None):
        self.n = int(n)
        self.rows = rows if rows is not None else _lib.DeviceArray(3 * max(self.n, 1), np.uint64)
        self.keys = keys if keys is not None else _lib.DeviceArray(max(self.n, 1), np.uint32)

    def free(self):
        self.rows.free()
        self.keys.free()


def partition_rows_by_bucket_range(dflat, modulo, n_parts, group_bits=0, out=None, max_rows_per_pass=0):
    """partition_by_bucket_range with the partitioned records left as rows (DeviceRows).  Returns (DeviceRows, start)."""
    _lib.require_device()
    if out is None:
        out = DeviceRows(dflat.n)
    assert out.keys.n >= dflat.n and out.rows.n >= 3 * dflat.n
    out.n = dflat.n
    start = (C.c_int64 * ((n_parts << group_bits) + 1))()
    _lib.check(_lib.load().gki_partition_rows_by_bucket_range(
        dflat.hashes.ptr, dflat.nodes.ptr, dflat.ref_offsets.ptr, dflat.allele_frequencies.ptr, dflat.n, int(modulo),
        int(n_parts), int(group_bits), int(max_rows_per_pass), out.rows.ptr, out.keys.ptr, start))
    return out, [int(x) for x in start]


class PartitionedDeviceIndex:
    """A CollisionFreeKmerIndex cut into bucket-range slices, each a DeviceIndex with its own directory slice
    (SURVEY.md 8f-1).  On one GPU the slices sit side by side (no 2^31 limit on the total); across GPUs each rank
    keeps one slice (`parallel.build_index_partitioned`).  A k-mer outside a slice's range misses there without a
    memory access, so counting over all slices equals counting on the monolithic index."""

    def __init__(self, modulo, parts):
        self.modulo, self.parts = int(modulo), list(parts)

    @property
    def n(self):
        return sum(p.n for p in self.parts)

    @classmethod
    def build(cls, dflat, modulo=452930477, n_parts=8, skip_frequencies=False, grouped=True, out=None):
        """Partition by bucket range, build every slice.  grouped (the default): the partition also groups every slice's
        records by the top bits of their key (as many as keep n_parts << bits within the partition pass's 1024 digits)
        and leaves them as rows + keys; the slice builds start from there with one sort pass less -- on the 3 Gbp graph
        187 ms against 195 through four columns and a plain partition (DESIGN.md 4.3 "Grouped build").  `out`: where the
        partitioned records go (DeviceRows when grouped, DeviceFlatKmers otherwise)."""
        g = max(0, 10 - max(0, (n_parts - 1).bit_length())) if grouped else 0
        if grouped:
            part, start = partition_rows_by_bucket_range(dflat, modulo, n_parts, group_bits=g, out=out)
        else:
            part, start = partition_by_bucket_range(dflat, modulo, n_parts, out=out)
        parts = []
        try:
            for p in range(n_parts):
                parts.append(cls.build_slice(part, start, modulo, n_parts, p, g, skip_frequencies))
        except _lib.GkiError as e:
            # a slice outside the row-carrying build's domain (a handful of buckets holding millions of records): the build
            # from columns can hand such a slice to the pair-sorting form, the build from rows cannot -- start over that way
            for sl in parts:
                sl.free()
            if out is None:
                part.free()
            if not grouped or e.code != 8:                # GKI_ERR_OUT_OF_DOMAIN
                raise
            return cls.build(dflat, modulo, n_parts, skip_frequencies, grouped=False)
        if out is None:
            part.free()
        return cls(modulo, parts)

    @staticmethod
    def build_slice(part, start, modulo, n_parts, p, group_bits, skip_frequencies=False):
        """The DeviceIndex of slice p from the partitioned records: `part` a DeviceFlatKmers (partition_by_bucket_range)
        or a DeviceRows (partition_rows_by_bucket_range), `start` the table that call returned."""
        lo, hi = bucket_range(modulo, n_parts, p)
        a, b = start[p << group_bits], start[(p + 1) << group_bits]
        groups = [x - a for x in start[p << group_bits:((p + 1) << group_bits) + 1]] if group_bits else None
        if isinstance(part, DeviceRows):
            return DeviceIndex.build_from_rows(part.rows.view(3 * a, 3 * (b - a)), part.keys.view(a, b - a), b - a, modulo,
                                               skip_frequencies, lo, hi - lo, groups)
        sl = DeviceFlatKmers(b - a, part.hashes.view(a, b - a), part.nodes.view(a, b - a),
                             part.ref_offsets.view(a, b - a), part.allele_frequencies.view(a, b - a))
        return DeviceIndex.build(sl, modulo, skip_frequencies, bucket_begin=lo, n_buckets=hi - lo, group_start=groups)

    def count_nodes(self, queries, n_nodes, max_hits=10, counts=None):
        own = not isinstance(queries, _lib.DeviceArray)
        dq = _lib.DeviceArray.from_host(np.ascontiguousarray(np.asarray(queries)).astype(np.uint64)) if own else queries
        for p in self.parts:
            counts = p.count_nodes(dq, n_nodes, max_hits, counts)
        if own:
            dq.free()
        return counts

    def count_nodes_from_reads(self, letters, read_start, k, n_nodes, strands=3, max_hits=10, counts=None):
        dl = letters if isinstance(letters, _lib.DeviceArray) else _lib.DeviceArray.from_host(np.ascontiguousarray(letters, np.uint8))
        ds = read_start if isinstance(read_start, _lib.DeviceArray) else \
            _lib.DeviceArray.from_host(np.ascontiguousarray(read_start, np.int64))
        n_kmers = hits = 0
        for p in self.parts:
            counts, n_kmers, h = p.count_nodes_from_reads(dl, ds, k, n_nodes, strands, max_hits, counts)
            hits += h
        return counts, n_kmers, hits

    def free(self):
        for p in self.parts:
            p.free()


class CounterKmerIndex:
    """collision_free_kmer_index.py:14-40 (what kmer_mapper feeds KAGE): count how often every index k-mer occurs among
    the k-mers handed to `count_kmers`, then `get_node_counts` = for every node the summed counts of its records'
    k-mers.  The reference keeps a per-k-mer npstructures.Counter (uint16 values) and reduces it with np.bincount; here
    the probe adds straight into a per-node histogram on the device (gki_probe_count_nodes), which is the same sum as
    long as no k-mer is seen more than 65 535 times (npstructures' behaviour beyond that is not pinned offline)."""

    def __init__(self, kmers, nodes, counter=None, modulo=452930477):
        self.kmers = kmers
        self.nodes = nodes
        self.counter = counter
        self._modulo = int(modulo)
        self._index = None
        self._counts = None

    @classmethod
    def from_kmer_index(cls, kmer_index):
        out = cls(np.asarray(kmer_index._kmers).astype(np.int64), kmer_index._nodes, None, kmer_index._modulo)
        out._index = kmer_index._device_index()
        return out

    def _n_nodes(self):
        return int(np.max(self.nodes)) + 1 if len(self.nodes) else 1

    def reset(self):
        if self._counts is not None:
            self._counts.zero()

    def count_kmers(self, kmers, update_counter=True):
        if not update_counter:
            self.reset()
        if self._index is None:                       # constructed from bare arrays: build the table once
            z = np.zeros(len(self.kmers), np.uint64)
            self._index = DeviceIndex.build(DeviceFlatKmers.from_flat_kmers(
                FlatKmers(np.asarray(self.kmers).astype(np.uint64), np.asarray(self.nodes).astype(np.uint32), z,
                          z.astype(np.float32))), self._modulo, skip_frequencies=True)
        self._counts = self._index.count_nodes(np.asarray(kmers).astype(np.int64).view(np.uint64), self._n_nodes(),
                                               max_hits=2 ** 62, counts=self._counts)

    def get_node_counts(self, min_nodes=0):
        """float64 like np.bincount with weights (:39-40)."""
        n = self._n_nodes()
        got = self._counts.to_host(n).astype(np.float64) if self._counts is not None else np.zeros(n)
        if min_nodes > n:
            got = np.concatenate([got, np.zeros(min_nodes - n)])
        return got


_NO_HITS = np.zeros(0, dtype=np.int64)
_ONE_HIT = np.zeros(1, dtype=np.int64)


class CollisionFreeKmerIndex:
    properties = {"_hashes_to_index", "_n_kmers", "_nodes", "_ref_offsets", "_kmers", "_modulo", "_frequencies",
                  "_allele_frequencies"}

    def __init__(self, _hashes_to_index=None, _n_kmers=None, _nodes=None, _ref_offsets=None, _kmers=None,
                 _modulo=452930477, _frequencies=None, _allele_frequencies=None):
        self._hashes_to_index = _hashes_to_index
        self._n_kmers = _n_kmers
        self._nodes = _nodes
        self._ref_offsets = _ref_offsets
        self._kmers = _kmers
        self._modulo = int(_modulo)
        self._frequencies = 0 if _frequencies is None else _frequencies
        self._allele_frequencies = _allele_frequencies
        self._device = None

    # ------------------------------------------------------------------ build
    @classmethod
    def from_flat_kmers(cls, flat_kmers, modulo=452930477, skip_frequencies=False, skip_singletons=False):
        """collision_free_kmer_index.py:423-467.  Records of a bucket keep their input order (the
        reference's np.argsort leaves that order unspecified)."""
        if skip_singletons:
            flat_kmers = flat_kmers.get_new_without_singletons()
        host_cols = None
        if not isinstance(flat_kmers, DeviceFlatKmers):
            host_cols = [np.asarray(flat_kmers._hashes), np.asarray(flat_kmers._nodes),
                         np.asarray(flat_kmers._ref_offsets), np.asarray(flat_kmers._allele_frequencies)]
            dflat = DeviceFlatKmers.from_flat_kmers(flat_kmers)
        else:
            dflat = flat_kmers
        dev = DeviceIndex.build(dflat, modulo, skip_frequencies, want_permutation=host_cols is not None)
        n = dflat.n
        freq = dev.frequencies.to_host(n)
        if skip_singletons:
            freq = freq + np.uint16(1)                                                     # :463-465
        dev_cols = [dev.kmers, dev.nodes, dev.ref_offsets, dev.allele_frequencies]
        if host_cols is None:
            cols = [c.to_host(n) for c in dev_cols]
        else:
            # the reference permutes the caller's arrays whatever their dtype (:436-440): columns whose dtype is the
            # device's (up to signedness) come back from HBM, others are permuted here with the device's `sorting`
            dflat.free()
            perm = None
            cols = []
            for h, d in zip(host_cols, dev_cols):
                if h.dtype.kind in "iu" and d.dtype.kind in "iu" and h.dtype.itemsize == d.dtype.itemsize \
                        or h.dtype == d.dtype:
                    cols.append(d.to_host(n).view(h.dtype))
                else:
                    if perm is None:
                        perm = dev.permutation.to_host(n).astype(np.int64)
                    cols.append(h[perm])
            dev.permutation.free()
            dev.permutation = None
        obj = cls(dev.hashes_to_index.to_host(), dev.n_kmers.to_host(), cols[1], cols[2], cols[0], modulo, freq, cols[3])
        if skip_singletons:
            dev.free()            # device frequencies differ from the host ones by the +1
        else:
            obj._device = dev
        return obj

    def _device_index(self):
        if self._device is None:
            _lib.require_device()
            n = len(self._kmers)
            freq = self._frequencies
            if not isinstance(freq, np.ndarray) or len(freq) != n:
                freq = np.zeros(n, dtype=np.uint16)
            af = self._allele_frequencies if self._allele_frequencies is not None else np.zeros(n, np.float32)
            h = _lib.DeviceArray.from_host
            self._device = DeviceIndex(
                n, self._modulo, h(np.asarray(self._hashes_to_index).astype(np.int32)),
                h(np.asarray(self._n_kmers).astype(np.uint32)), h(np.asarray(self._kmers).astype(np.uint64)),
                h(np.asarray(self._nodes).astype(np.uint32)), h(np.asarray(self._ref_offsets).astype(np.uint64)),
                h(np.asarray(af).astype(np.float32)), h(np.asarray(freq).astype(np.uint16)))
        return self._device

    def _invalidate_device(self):
        if self._device is not None:
            self._device.free()
            self._device = None

    # ------------------------------------------------------------------ housekeeping (reference :191-244)
    def clear(self):
        self._hashes_to_index = self._n_kmers = self._nodes = self._kmers = self._modulo = None
        self._invalidate_device()
        gc.collect()

    def copy(self):
        return CollisionFreeKmerIndex(self._hashes_to_index.copy(), self._n_kmers.copy(), self._nodes.copy(),
                                      self._ref_offsets.copy(), self._kmers.copy(), self._modulo,
                                      self._frequencies.copy(), self._allele_frequencies.copy())

    def get_kmers(self):
        return self._kmers

    def set_allele_frequencies(self, frequencies):
        pass

    def max_node_id(self):
        return np.max(self._nodes)

    def convert_to_int32(self):
        self._hashes_to_index = self._hashes_to_index.astype(np.int32)
        self._nodes = self._nodes.astype(np.int32)
        self._n_kmers = self._n_kmers.astype(np.int32)
        self._modulo = np.uint64(self._modulo)

    def remove_ref_offsets(self):
        self._ref_offsets = np.array([0])
        self._invalidate_device()

    def remove_frequencies(self):
        self._frequencies = np.array([0])
        self._invalidate_device()

    def has_kmers(self, kmers):
        """kmer_mapper.in_graph_index equivalent (:214-216): membership of every query."""
        flags = self._device_index().contains(kmers)
        out = flags.to_host(len(kmers)).astype(bool)
        flags.free()
        return out

    def has_kmers_parallel(self, kmers, n_threads=1):
        """collision_free_kmer_index.py:222-232 spreads `has_kmers` over a shared-memory process pool; one batched probe
        on the device does the whole array, so `n_threads` is accepted and ignored."""
        return self.has_kmers(kmers)

    def map_kmers(self, kmers, n_nodes):
        """kmer_mapper.map_kmers_to_graph_index equivalent (:210-212): node hit counts, probe and histogram fused
        on the device."""
        counts = self._device_index().count_nodes(kmers, n_nodes, max_hits=2 ** 62)
        out = counts.to_host(n_nodes)
        counts.free()
        return out

    def map_reads(self, reads, k, n_nodes, max_hits=2 ** 62, include_reverse_complement=True):
        """ReadKmers (read_kmers.py:21-26) + map_kmers in one device pass: node hit counts of all k-mers of `reads`
        (list of str/bytes, or (uint8 letters, int64 read_start)) and, by default, of their reverse complements."""
        if isinstance(reads, tuple):
            letters, read_start = reads
        else:
            enc = [r.encode("ascii") if isinstance(r, str) else bytes(r) for r in reads]
            read_start = np.zeros(len(enc) + 1, dtype=np.int64)
            np.cumsum([len(e) for e in enc], out=read_start[1:])
            letters = np.frombuffer(b"".join(enc), dtype=np.uint8)
        counts, _, _ = self._device_index().count_nodes_from_reads(letters, read_start, k, n_nodes,
                                                                   3 if include_reverse_complement else 1, max_hits)
        out = counts.to_host(n_nodes)
        counts.free()
        return out

    # ------------------------------------------------------------------ probes
    def _hit_positions(self, kmer):
        """Positions of `kmer`'s records in the payload arrays, bucket order (collision_free_kmer_index.py:304-309), from
        this object's own host arrays: one k-mer is a latency-bound accessor -- a device round trip (launch +
        synchronisation, ~20 us) costs four times the bucket's few reads, so the scalar getters stay on the host and the
        device serves the batched ones (and `DeviceIndex.get_small` the indexes that only exist in HBM)."""
        kmer = int(kmer)
        b = kmer % self._modulo
        n = int(self._n_kmers[b])                                            # :306
        if n == 0:
            return _NO_HITS
        s = int(self._hashes_to_index[b])                                    # :305
        try:
            if n == 1:
                return _ONE_HIT + s if self._kmers[s] == kmer else _NO_HITS
            return np.flatnonzero(self._kmers[s:s + n] == kmer) + s          # :309
        except OverflowError:                                                # a query no element of the column can equal
            return _NO_HITS

    def get(self, kmer, max_hits=10):
        """collision_free_kmer_index.py:303-315."""
        pos = self._hit_positions(kmer)
        if len(pos) == 0:
            return None, None, None, None
        frequencies = self._frequencies[pos]
        if frequencies[0] > max_hits:                                        # :312
            return None, None, None, None
        return self._nodes[pos], self._ref_offsets[pos], frequencies, self._allele_frequencies[pos]

    def __contains__(self, item):
        return self.get(int(item), 100000000000)[0] is not None

    def get_nodes(self, kmer, max_hits=10):
        return self.get(kmer, max_hits)[0]

    def get_grouped_nodes(self, kmer, max_hits=10):
        hits = self.get(kmer, max_hits)
        if hits[0] is None:
            return None
        sorting = np.argsort(hits[1])
        ref_offsets, nodes = hits[1][sorting], hits[0][sorting]
        _, idx = np.unique(ref_offsets, return_index=True)
        idx = list(idx) + [len(ref_offsets)]
        return [nodes[a:b] for a, b in zip(idx[:-1], idx[1:])]

    def get_frequency(self, kmer, include_reverse_complement=True, k=31):
        """:336-352 -- the first hit's frequency of the k-mer, plus that of its reverse complement."""
        queries = [int(kmer)]
        if include_reverse_complement:
            queries.append(int(kmer_hash_to_reverse_complement_hash(kmer, k)))
        f = 0
        for q in queries:
            pos = self._hit_positions(q)
            if len(pos):                                 # the first hit's frequency (:342, :349)
                f += int(self._frequencies[pos[0]])
        return f

    def get_frequencies(self, kmers, include_reverse_complement=True, k=31):
        """get_frequency for many k-mers: int64 array, two batched probes (k-mers, reverse complements)."""
        kmers = np.ascontiguousarray(np.asarray(kmers)).astype(np.uint64)

        def first_hit_frequency(queries):
            hs, pos, _ = self._device_index().lookup_positions(queries, max_hits=1000000000000000)
            f = np.zeros(len(queries), dtype=np.int64)
            has = np.diff(hs) > 0
            f[has] = np.asarray(self._frequencies)[pos[hs[:-1][has]]]
            return f
        out = first_hit_frequency(kmers)
        if include_reverse_complement:
            from .kmer_hashing import kmer_hashes_to_reverse_complement_hash
            out = out + first_hit_frequency(kmer_hashes_to_reverse_complement_hash(kmers, k))
        return out

    def get_nodes_and_ref_offsets_from_multiple_kmers(self, kmers, max_hits=10):
        """:354-376 -- one batched device probe instead of a Python loop of get()."""
        _, pos, query = self._device_index().lookup_positions(kmers, max_hits)
        if len(pos) == 0:
            return np.array([]), np.array([]), np.array([]), np.array([])
        return self._nodes[pos], self._ref_offsets[pos], query.astype(np.float64), self._frequencies[pos]

    def get_nodes_from_multiple_kmers(self, kmers, max_hits=10):
        """:378-391."""
        _, pos, _ = self._device_index().lookup_positions(kmers, max_hits)
        if len(pos) == 0:
            return np.array([])
        return self._nodes[pos]

    # ------------------------------------------------------------------ I/O (:393-420)
    def to_file(self, file_name):
        np.savez(file_name, hashes_to_index=self._hashes_to_index, n_kmers=self._n_kmers, nodes=self._nodes,
                 ref_offsets=self._ref_offsets, kmers=self._kmers, modulo=self._modulo, frequencies=self._frequencies,
                 allele_frequencies=self._allele_frequencies)

    @classmethod
    def from_file(cls, file_name):
        try:
            data = np.load(file_name + ".npz")
        except FileNotFoundError:
            data = np.load(file_name)
        af = data["allele_frequencies"] if "allele_frequencies" in data else np.zeros(len(data["ref_offsets"]))
        return cls(data["hashes_to_index"], data["n_kmers"], data["nodes"], data["ref_offsets"], data["kmers"],
                   data["modulo"], data["frequencies"], af)

    def set_frequencies_using_other_index(self, other, multiplier=1, min_frequency=1):
        """:246-265 -- every record gets max(min_frequency, other.get_frequency(kmer) * multiplier), where
        get_frequency adds the hit of the reverse complement (k=31, :336-352).  Two batched probes of `other`."""
        kmers = np.asarray(self._kmers)
        uniq, inverse = np.unique(kmers, return_inverse=True)
        freq = other.get_frequencies(uniq, True, 31)
        value = np.maximum(min_frequency, freq * multiplier)
        self._frequencies = np.asarray(self._frequencies).copy()
        self._frequencies[:] = value[inverse].astype(self._frequencies.dtype)
        self._invalidate_device()

    def convert_kmers_to_complement(self, k=31, skip_frequencies=True):
        """:470-490 -- rebuild the index on the complement (not reversed) hashes."""
        from .kmer_hashing import kmer_hashes_to_complement_hashes
        new_kmers = kmer_hashes_to_complement_hashes(np.asarray(self._kmers), k)
        return CollisionFreeKmerIndex.from_flat_kmers(
            FlatKmers(new_kmers, self._nodes, self._ref_offsets, self._allele_frequencies), modulo=self._modulo,
            skip_frequencies=skip_frequencies)

    def set_frequencies(self, skip=False):
        """:267-293 -- rebuilds the frequency column on device from the current payload."""
        n = len(self._kmers)
        self._frequencies = np.zeros(n, dtype=np.uint16)
        if skip:
            return
        flat = FlatKmers(np.asarray(self._kmers), np.asarray(self._nodes), np.asarray(self._ref_offsets),
                         np.asarray(self._allele_frequencies))
        rebuilt = CollisionFreeKmerIndex.from_flat_kmers(flat, self._modulo)
        # a stable re-sort of already bucket-sorted records is the identity permutation
        self._frequencies = rebuilt._frequencies
        logging.info("Frequencies set")
