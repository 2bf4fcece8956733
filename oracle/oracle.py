"""ctypes front-end of the CPU oracle (oracle/gki_oracle.c).

TEST INFRASTRUCTURE, NOT PRODUCT CODE: importable only from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg.  graph_kmer_index_amd never imports this module.

The functions mirror the reference entry points they restate (file:line in gki_oracle.c).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgki_oracle.so")

ORC_FLAG_UNDEFINED_BULK = 1
ERRORS = {1: "out of memory", 2: "recursion limit (graph too complex between critical points)",
          3: "not exactly one linear-ref successor", 4: "negative critical offset (reference E2 crash)",
          5: "bad argument"}


class OracleError(Exception):
    def __init__(self, code):
        super().__init__("oracle error %d: %s" % (code, ERRORS.get(code, "?")))
        self.code = code


def build(force=False):
    src = os.path.join(_HERE, "gki_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libgki_oracle.so"])
    return _LIB_PATH


class _Graph(C.Structure):
    _fields_ = [("n_nodes", C.c_int64), ("node_size", C.c_void_p), ("seq_start", C.c_void_p),
                ("seq", C.c_void_p), ("edge_start", C.c_void_p), ("edges", C.c_void_p),
                ("rev_start", C.c_void_p), ("rev_edges", C.c_void_p), ("is_ref", C.c_void_p),
                ("allele_freq", C.c_void_p), ("first_node", C.c_int32), ("n_chrom", C.c_int32),
                ("chrom_start", C.c_void_p)]


class _Records(C.Structure):
    _fields_ = [("n", C.c_int64), ("cap", C.c_int64), ("kmers", C.c_void_p), ("nodes", C.c_void_p),
                ("start_nodes", C.c_void_p), ("start_offsets", C.c_void_p), ("af", C.c_void_p),
                ("window_id", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_kmer_to_hash.restype = C.c_uint64
        _lib.orc_update_hash.restype = C.c_uint64
        _lib.orc_update_hash.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int]
        _lib.orc_hash_sequence_valid.restype = C.c_int64
        _lib.orc_critical_paths.restype = C.c_int64
        _lib.orc_index_get.restype = C.c_int64
        _lib.orc_without_singletons.restype = C.c_int64
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _graph_struct(g):
    chrom = np.ascontiguousarray(list(g.chromosome_start_nodes.values()), dtype=np.int32)
    s = _Graph(g.n_nodes, _p(g.node_size), _p(g.seq_start), _p(g.seq), _p(g.edge_start), _p(g.edges),
               _p(g.rev_start), _p(g.rev_edges), _p(g.is_ref), _p(g.allele_freq), g.first_node,
               len(chrom), _p(chrom))
    s._keep = (chrom, g)
    return s


# ------------------------------------------------------------------ hashing (A1, A2, A8, A10)
def letter_sequence_to_numeric(sequence):
    b = sequence.encode("ascii") if isinstance(sequence, str) else bytes(sequence)
    out = np.zeros(len(b), dtype=np.uint8)
    lib().orc_letters_to_numeric(b, C.c_int64(len(b)), _p(out))
    return out


def kmer_to_hash(codes):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    return int(lib().orc_kmer_to_hash(_p(codes), C.c_int(len(codes))))


def sequence_to_kmer_hash(sequence):
    return kmer_to_hash(letter_sequence_to_numeric(sequence))


def kmer_hash_to_sequence(h, k):
    return "".join("acgt"[(int(h) >> (2 * i)) & 3] for i in range(k))


def update_hash(base, current_hash, first_base, k, only_add=False):
    oa = -1 if isinstance(only_add, bool) else int(only_add)
    return int(lib().orc_update_hash(int(base), int(current_hash), int(first_base), k, oa))


def hash_sequence(codes, k):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    out = np.zeros(max(0, len(codes) - k + 1), dtype=np.uint64)
    lib().orc_hash_sequence_valid(_p(codes), C.c_int64(len(codes)), C.c_int(k), _p(out))
    return out


def read_kmers(read, k):
    """ReadKmers.get_kmers_from_read_dynamic (read_kmers.py:67-70)."""
    return hash_sequence(letter_sequence_to_numeric(read), k)


def reverse_complement(hashes, k):
    hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
    out = np.zeros_like(hashes)
    lib().orc_reverse_complement(_p(hashes), C.c_int64(len(hashes)), C.c_int(k), _p(out))
    return out


def complement(hashes, k):
    hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
    out = np.zeros_like(hashes)
    lib().orc_complement(_p(hashes), C.c_int64(len(hashes)), C.c_int(k), _p(out))
    return out


# ------------------------------------------------------------------ critical paths (A6)
def critical_paths(g, k):
    gs = _graph_struct(g)
    nodes = np.zeros(g.n_nodes, dtype=np.uint32)
    offsets = np.zeros(g.n_nodes, dtype=np.uint16)
    n = lib().orc_critical_paths(C.byref(gs), C.c_int(k), _p(nodes), _p(offsets))
    if n < 0:
        raise OracleError(-n)
    return nodes[:n].copy(), offsets[:n].copy()


# ------------------------------------------------------------------ finder (A3-A5)
def _take(rec, with_window_id=False):
    n = rec.n

    def arr(ptr, dt):
        if n == 0:
            return np.zeros(0, dtype=dt)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_byte)), shape=(n * np.dtype(dt).itemsize,)) \
            .view(dt).copy()

    out = dict(kmers=arr(rec.kmers, np.int64), nodes=arr(rec.nodes, np.int32),
               start_nodes=arr(rec.start_nodes, np.int32), start_offsets=arr(rec.start_offsets, np.int16),
               allele_frequencies=arr(rec.af, np.float64))
    if with_window_id:
        out["window_id"] = arr(rec.window_id, np.int64)
    lib().orc_free_records(C.byref(rec))
    return out


def _node_mask(g, nodes):
    if nodes is None:
        return None
    m = np.zeros(g.n_nodes, dtype=np.uint8)
    m[np.asarray(sorted(nodes), dtype=np.int64)] = 1
    return m


def find(g, k, critical=None, only_save_one_node_per_kmer=False, max_variant_nodes=4,
         start_at_critical_path_number=None, stop_at_critical_path_number=None, whitelist=None,
         only_store_nodes=None, only_follow_nodes=None, with_window_id=False, return_flags=False):
    """DenseKmerFinder(...).find() + get_flat_kmers(v="2") columns (kmer_finder.py:179-244,109-126)."""
    if critical is None:
        critical = critical_paths(g, k)
    cn = np.ascontiguousarray(critical[0], dtype=np.uint32)
    co = np.ascontiguousarray(critical[1], dtype=np.uint16)
    wl = None if whitelist is None else np.ascontiguousarray(sorted(whitelist), dtype=np.int64)
    osn, ofn = _node_mask(g, only_store_nodes), _node_mask(g, only_follow_nodes)
    gs = _graph_struct(g)
    rec, flags = _Records(), C.c_int32(0)
    err = lib().orc_find(C.byref(gs), C.c_int(k), _p(cn), _p(co), C.c_int64(len(cn)),
                         C.c_int(bool(only_save_one_node_per_kmer)), C.c_int(max_variant_nodes),
                         C.c_int64(-1 if start_at_critical_path_number is None else start_at_critical_path_number),
                         C.c_int64(-1 if stop_at_critical_path_number is None else stop_at_critical_path_number),
                         _p(wl), C.c_int64(-1 if wl is None else len(wl)), _p(osn), _p(ofn),
                         C.byref(rec), C.byref(flags))
    if err:
        raise OracleError(err)
    out = _take(rec, with_window_id)
    return (out, flags.value) if return_flags else out


def find_from_position(g, k, node, offset, only_save_one_node_per_kmer=False, max_variant_nodes=4,
                       whitelist=None, only_store_nodes=None, only_follow_nodes=None):
    """DenseKmerFinder.find_only_kmers_starting_at_position (kmer_finder.py:170-177)."""
    wl = None if whitelist is None else np.ascontiguousarray(sorted(whitelist), dtype=np.int64)
    osn, ofn = _node_mask(g, only_store_nodes), _node_mask(g, only_follow_nodes)
    gs = _graph_struct(g)
    rec, flags = _Records(), C.c_int32(0)
    err = lib().orc_find_from_position(C.byref(gs), C.c_int(k), C.c_int32(node), C.c_int64(offset),
                                       C.c_int(bool(only_save_one_node_per_kmer)), C.c_int(max_variant_nodes),
                                       _p(wl), C.c_int64(-1 if wl is None else len(wl)), _p(osn), _p(ofn),
                                       C.byref(rec), C.byref(flags))
    if err:
        raise OracleError(err)
    return _take(rec)


# ------------------------------------------------------------------ index (A9)
def without_singletons(hashes):
    hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
    keep = np.zeros(len(hashes), dtype=np.uint8)
    n = lib().orc_without_singletons(_p(hashes), C.c_int64(len(hashes)), _p(keep))
    if n < 0:
        raise OracleError(-n)
    return keep.astype(bool)


def index_build(kmers, nodes, ref_offsets, allele_frequencies, modulo=452930477,
                skip_frequencies=False, skip_singletons=False):
    """CollisionFreeKmerIndex.from_flat_kmers (collision_free_kmer_index.py:423-467); returns the
    attribute arrays as a dict keyed like the reference's `properties` (:164-173)."""
    kmers = np.asarray(kmers)
    nodes = np.asarray(nodes)
    ref_offsets = np.asarray(ref_offsets)
    allele_frequencies = np.asarray(allele_frequencies)
    if skip_singletons:
        keep = without_singletons(kmers)
        kmers, nodes, ref_offsets, allele_frequencies = (kmers[keep], nodes[keep], ref_offsets[keep],
                                                         allele_frequencies[keep])
    n = len(kmers)
    ku = np.ascontiguousarray(kmers, dtype=np.uint64)
    ru = np.ascontiguousarray(ref_offsets).astype(np.int64).view(np.uint64)
    sorting = np.zeros(n, dtype=np.int64)
    h2i = np.zeros(modulo, dtype=np.int32)
    nk = np.zeros(modulo, dtype=np.uint32)
    freq = np.zeros(n, dtype=np.uint16)
    err = lib().orc_index_build(_p(ku), _p(ru), C.c_int64(n), C.c_uint64(modulo), C.c_int(bool(skip_frequencies)),
                                _p(sorting), _p(h2i), _p(nk), _p(freq))
    if err:
        raise OracleError(err)
    if skip_singletons:
        freq = freq + np.uint16(1)
    return dict(_hashes_to_index=h2i, _n_kmers=nk, _nodes=nodes[sorting], _ref_offsets=ref_offsets[sorting],
                _kmers=kmers[sorting], _modulo=int(modulo), _frequencies=freq,
                _allele_frequencies=allele_frequencies[sorting])


def index_get(index, kmer, max_hits=10):
    """CollisionFreeKmerIndex.get (collision_free_kmer_index.py:303-315)."""
    h = int(kmer) % index["_modulo"]
    nb = int(index["_n_kmers"][h])
    hit = np.zeros(max(nb, 1), dtype=np.int64)
    ks = np.ascontiguousarray(index["_kmers"]).astype(np.int64, copy=False).view(np.uint64)
    fr = np.ascontiguousarray(index["_frequencies"], dtype=np.uint16)
    c = lib().orc_index_get(_p(index["_hashes_to_index"]), _p(index["_n_kmers"]), _p(ks), _p(fr),
                            C.c_uint64(index["_modulo"]), C.c_uint64(int(kmer)), C.c_int64(min(max_hits, 2**62)),
                            _p(hit))
    if c < 0:
        return None, None, None, None
    hit = hit[:c]
    return (index["_nodes"][hit], index["_ref_offsets"][hit], index["_frequencies"][hit],
            index["_allele_frequencies"][hit])


# ------------------------------------------------------------------ batch loops (bench.py's CPU baselines, test checker)
def map_reads(index, letters, read_start, k, n_nodes, strands=3, max_hits=10):
    """Node counts of all k-mers of the reads (both strands by default): a loop of read_kmers + index_get in C.
    Returns (counts uint32[n_nodes], n_kmers, n_hits)."""
    letters = np.ascontiguousarray(letters, dtype=np.uint8)
    read_start = np.ascontiguousarray(read_start, dtype=np.int64)
    counts = np.zeros(n_nodes, dtype=np.uint32)
    nk, nh = C.c_int64(0), C.c_int64(0)
    ks = np.ascontiguousarray(index["_kmers"]).astype(np.int64, copy=False).view(np.uint64)
    fr = index["_frequencies"]
    fr = None if not isinstance(fr, np.ndarray) or len(fr) != len(ks) else np.ascontiguousarray(fr, dtype=np.uint16)
    nodes = np.ascontiguousarray(index["_nodes"]).astype(np.uint32, copy=False)
    err = lib().orc_map_reads(_p(letters), _p(read_start), C.c_int64(len(read_start) - 1), C.c_int(k), C.c_int(strands),
                              _p(index["_hashes_to_index"]), _p(index["_n_kmers"]), _p(ks), _p(nodes), _p(fr),
                              C.c_uint64(index["_modulo"]), C.c_int64(min(max_hits, 2 ** 62)), _p(counts),
                              C.c_int64(n_nodes), C.byref(nk), C.byref(nh))
    if err:
        raise OracleError(-err)
    return counts, nk.value, nh.value


def find_from_positions(g, k, nodes, offsets, only_save_one_node_per_kmer=False, max_variant_nodes=4):
    """Number of records of a loop of find_only_kmers_starting_at_position over the start positions."""
    nodes = np.ascontiguousarray(nodes, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    gs = _graph_struct(g)
    n = C.c_int64(0)
    err = lib().orc_find_from_positions(C.byref(gs), C.c_int(k), _p(nodes), _p(offsets), C.c_int64(len(nodes)),
                                        C.c_int(bool(only_save_one_node_per_kmer)), C.c_int(max_variant_nodes), C.byref(n))
    if err:
        raise OracleError(err)
    return n.value
