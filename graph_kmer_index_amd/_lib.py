"""ctypes binding of libgki_hip.so (C ABI in include/gki.h).

There is no CPU fallback: if the shared library is missing or no MI355X is visible, the
operations raise.  `python -c "import __graft_entry__ as g; g.build()"` (or
`make -C graph_kmer_index_amd/csrc`) builds the library in-tree.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GKI_LIB selects another build of the same C ABI (the A/B scripts of tools/exp compare builds this way instead of
# copying a variant over the product library)
LIB_PATH = os.environ.get("GKI_LIB") or os.path.join(_HERE, "libgki_hip.so")


class GkiError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libgki_hip error %d: %s" % (code, message))
        self.code = code


class NotOneLinearRefSuccessor(GkiError, AssertionError):
    """GKI_ERR_NOT_ONE_REF_SUCC: where the reference's `assert len(next_nodes) == 1` fails (kmer_finder.py:402)."""


class WindowTooDeep(GkiError, RecursionError):
    """GKI_ERR_WINDOW_TOO_DEEP: a k-window over more nodes than the kernels' slow path holds, or more paths into one end
    node than it will enumerate.  The reference ends such a search in a RecursionError (its search recurses per node under
    sys.setrecursionlimit(20000), kmer_finder.py:7, and re-raises :234-241), so a caller that catches that catches this."""


class FindParams(C.Structure):
    """gki_find_params (include/gki.h); struct_size is filled in by the constructor."""
    _fields_ = [("struct_size", C.c_uint32), ("k", C.c_int32), ("max_variant_nodes", C.c_int32),
                ("one_node_per_kmer", C.c_int32), ("layout", C.c_int32), ("node_begin", C.c_int64),
                ("off_begin", C.c_int64), ("node_end", C.c_int64), ("off_end", C.c_int64),
                ("h_lossy_crit", C.c_void_p), ("h_node_rank", C.c_void_p), ("h_node_flags", C.c_void_p),
                ("h_store_nodes", C.c_void_p),
                ("d_lossy_crit", C.c_void_p), ("d_node_rank", C.c_void_p), ("d_node_flags", C.c_void_p),
                ("d_store_nodes", C.c_void_p), ("rank_begin", C.c_int32), ("rank_end", C.c_int32)]

    def __init__(self, k, max_variant_nodes, one_node_per_kmer, layout, node_begin, off_begin, node_end, off_end,
                 h_lossy_crit=None, h_node_rank=None, h_node_flags=None, h_store_nodes=None,
                 d_lossy_crit=None, d_node_rank=None, d_node_flags=None, d_store_nodes=None, rank_begin=0, rank_end=0):
        super().__init__(C.sizeof(FindParams), k, max_variant_nodes, one_node_per_kmer, layout, node_begin, off_begin,
                         node_end, off_end, h_lossy_crit, h_node_rank, h_node_flags, h_store_nodes,
                         d_lossy_crit, d_node_rank, d_node_flags, d_store_nodes, rank_begin, rank_end)


class IndexView(C.Structure):
    _fields_ = [("d_hashes_to_index", C.c_void_p), ("d_n_kmers", C.c_void_p), ("d_kmers", C.c_void_p),
                ("d_nodes", C.c_void_p), ("d_ref_offsets", C.c_void_p), ("d_frequencies", C.c_void_p),
                ("d_af32", C.c_void_p), ("modulo", C.c_uint64), ("n", C.c_int64),
                ("bucket_begin", C.c_uint64), ("n_buckets", C.c_uint64)]


# every symbol include/gki.h declares: name -> (restype, argtypes)
_P, _I64, _I32, _U64 = C.c_void_p, C.c_int64, C.c_int, C.c_uint64
SYMBOLS = {
    "gki_last_error": (C.c_char_p, []),
    "gki_device_count": (_I32, [C.POINTER(C.c_int)]),
    "gki_set_device": (_I32, [_I32]),
    "gki_malloc": (_I32, [C.POINTER(_P), _I64]),
    "gki_free": (_I32, [_P]),
    "gki_trim": (_I32, []),
    "gki_pool_stats": (_I32, [C.POINTER(_I64), C.POINTER(_I64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_I64)]),
    "gki_memcpy_h2d": (_I32, [_P, _P, _I64]),
    "gki_memcpy_d2h": (_I32, [_P, _P, _I64]),
    "gki_memcpy_d2d": (_I32, [_P, _P, _I64]),
    "gki_memset": (_I32, [_P, _I32, _I64]),
    "gki_device_synchronize": (_I32, []),
    "gki_mem_info": (_I32, [C.POINTER(_I64), C.POINTER(_I64)]),
    "gki_flag_repeated_kmers": (_I32, [_P, _I64, _P]),
    "gki_compact_flat": (_I32, [_P, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _I64, C.POINTER(_I64)]),
    "gki_column_checksum": (_I32, [_P, _I64, _I32, C.POINTER(_U64), C.POINTER(_U64)]),
    "gki_hash_sequence": (_I32, [_P, _I64, _I32, _P]),
    "gki_hash_reads": (_I32, [_P, _P, _I64, _I32, _I32, _P, _P, _I64, C.POINTER(_I64)]),
    "gki_reverse_complement": (_I32, [_P, _I64, _I32, _P]),
    "gki_complement": (_I32, [_P, _I64, _I32, _P]),
    "gki_graph_create": (_I32, [C.POINTER(_P), _I64, _P, _P, _I64, _P, _P, _P, _P, _I64, _P, _P, _P]),
    "gki_graph_create_dseq": (_I32, [C.POINTER(_P), _I64, _P, _P, _I64, _P, _P, _P, _P, _I64, _P, _P, _P]),
    "gki_graph_prepare": (_I32, [_P]),
    "gki_graph_destroy": (_I32, [_P]),
    "gki_graph_n_bases": (_I64, [_P]),
    "gki_topological_rank": (_I32, [_I64, _P, _P, _P]),
    "gki_classify_nodes": (_I32, [_I64, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P, C.POINTER(C.c_int32)]),
    "gki_graph_classify_nodes": (_I32, [_P, _P, _P, _I32, _I32, _I32, _P, _I32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gki_find_params_size": (_I64, []),
    "gki_critical_paths": (_I32, [_I64, _P, _P, _P, _P, _P, _P, _I32, _I32, _P, _P, C.POINTER(_I64)]),
    "gki_graph_critical_paths": (_I32, [_P, _P, _I32, _I32, _P, _P, C.POINTER(_I64)]),
    "gki_finder_create": (_I32, [_P, C.POINTER(_P)]),
    "gki_finder_destroy": (_I32, [_P]),
    "gki_finder_count": (_I32, [_P, C.POINTER(FindParams), C.POINTER(_I64)]),
    "gki_finder_emit_flat": (_I32, [_P, _P, _P, _P, _P]),
    "gki_finder_emit_v2": (_I32, [_P, _P, _P, _P, _P, _P]),
    "gki_finder_synchronize": (_I32, [_P]),
    "gki_finder_kernel_ms": (_I32, [_P, _I32, C.POINTER(C.c_float)]),
    "gki_finder_interior_records": (_I64, [_P]),
    "gki_forward_count": (_I32, [_P, _I32, _I32, _I32, _P, _P, _P, _I64, _P, C.POINTER(_I64)]),
    "gki_forward_emit": (_I32, [_P, _I32, _I32, _I32, _P, _P, _P, _I64, _P, _P, _P, _P, _P, _P]),
    "gki_index_build": (_I32, [_P, _P, _P, _P, _I64, _U64, _I32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gki_partition_by_bucket_range": (_I32, [_P, _P, _P, _P, _I64, _U64, _I32, _P, _P, _P, _P, C.POINTER(_I64)]),
    "gki_partition_by_bucket_range_chunked": (_I32, [_P, _P, _P, _P, _I64, _U64, _I32, _I64, _P, _P, _P, _P, C.POINTER(_I64)]),
    "gki_partition_by_bucket_range_grouped": (_I32, [_P, _P, _P, _P, _I64, _U64, _I32, _I32, _I64, _P, _P, _P, _P, C.POINTER(_I64)]),
    "gki_index_build_range": (_I32, [_P, _P, _P, _P, _I64, _U64, _U64, _U64, _I32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gki_index_build_range_grouped": (_I32, [_P, _P, _P, _P, _I64, _U64, _U64, _U64, _I32, _I32, C.POINTER(_I64),
                                             _P, _P, _P, _P, _P, _P, _P, _P]),
    "gki_partition_rows_by_bucket_range": (_I32, [_P, _P, _P, _P, _I64, _U64, _I32, _I32, _I64, _P, _P, C.POINTER(_I64)]),
    "gki_index_build_range_from_rows": (_I32, [_P, _P, _I64, _U64, _U64, _U64, _I32, _I32, C.POINTER(_I64),
                                               _P, _P, _P, _P, _P, _P, _P]),
    "gki_index_build_pairs": (_I32, [_P, _P, _P, _P, _I64, _U64, _U64, _U64, _I32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gki_reverse_index_build": (_I32, [_P, _P, _P, _I64, _I64, _P, _P, _P, _P]),
    "gki_index_lookup_count": (_I32, [C.POINTER(IndexView), _P, _I64, _I64, _P, C.POINTER(_I64)]),
    "gki_index_count_nodes": (_I32, [C.POINTER(IndexView), _P, _I64, _I64, _P, _I64]),
    "gki_index_lookup_emit": (_I32, [C.POINTER(IndexView), _P, _I64, _I64, _P, _P, _P, _P, _P, _P, _P]),
    "gki_index_get_small": (_I32, [C.POINTER(IndexView), _P, _I32, _I64, _P, _P, _I64]),
    "gki_probe_create": (_I32, [C.POINTER(IndexView), C.POINTER(_P)]),
    "gki_probe_destroy": (_I32, [_P]),
    "gki_probe_lookup_count": (_I32, [_P, _P, _I64, _I64, _P, C.POINTER(_I64)]),
    "gki_probe_lookup_emit": (_I32, [_P, _P, _I64, _I64, _P, _P, _P]),
    "gki_probe_contains": (_I32, [_P, _P, _I64, _P]),
    "gki_probe_count_nodes": (_I32, [_P, _P, _I64, _I64, _P, _I64, C.POINTER(_I64)]),
    "gki_probe_reads_count_nodes": (_I32, [_P, _P, _P, _I64, _I32, _I32, _I64, _P, _I64, C.POINTER(_I64), C.POINTER(_I64)]),
    "gki_measure_random_loads": (_I32, [_I64, _I64, C.POINTER(C.c_double)]),
    "gki_measure_store_bw": (_I32, [_P, _P, _P, _P, _I64, C.POINTER(C.c_double)]),
    "gki_selftest_wave_scan": (_I32, [C.POINTER(_I64)]),
    "gki_simulate_reads": (_I32, [_P, _I64, _I64, _I32, _U64, C.c_double, C.c_double, _I64, _P]),
    "gki_comm_get_unique_id": (_I32, [_P]),
    "gki_comm_create": (_I32, [C.POINTER(_P), _I32, _I32, _P]),
    "gki_comm_destroy": (_I32, [_P]),
    "gki_comm_info": (_I32, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gki_device_bus_id": (_I32, [C.c_char_p, _I32]),
    "gki_ipc_export": (_I32, [_P, _P, C.POINTER(_I64)]),
    "gki_ipc_open": (_I32, [_P, C.POINTER(_P)]),
    "gki_ipc_close": (_I32, [_P]),
    "gki_comm_alltoall_flat": (_I32, [_P, C.POINTER(_I64), _P, _P, _P, _P, C.POINTER(_I64), _P, _P, _P, _P]),
    "gki_comm_allreduce_u32": (_I32, [_P, _P, _I64]),
    "gki_comm_allgather_flat": (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
}

_lib = None


def load():
    """Load libgki_hip.so and bind every exported symbol.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libgki_hip.so is not built (%s). Run `make -C graph_kmer_index_amd/csrc`; "
                          "graph_kmer_index_amd has no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    missing = [name for name in SYMBOLS if not hasattr(lib, name)]
    if missing:
        raise ImportError("libgki_hip.so is stale: missing %s (rebuild with make -C graph_kmer_index_amd/csrc)" % missing)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.gki_find_params_size() != C.sizeof(FindParams):
        raise ImportError("libgki_hip.so is stale: gki_find_params is %d bytes there, %d in this binding"
                          % (lib.gki_find_params_size(), C.sizeof(FindParams)))
    _lib = lib
    return lib


def check(code):
    if code != 0:
        cls = NotOneLinearRefSuccessor if code == 7 else WindowTooDeep if code == 4 else GkiError
        raise cls(code, load().gki_last_error().decode("utf-8", "replace"))


def device_count():
    n = C.c_int(0)
    rc = load().gki_device_count(C.byref(n))
    return 0 if rc != 0 else n.value


def require_device():
    if device_count() < 1:
        raise GkiError(3, "no HIP device visible: graph_kmer_index_amd needs an MI355X (no CPU fallback)")


def pool_stats():
    """(device mallocs, device frees, ms in hipMalloc, ms in hipFree, bytes parked) of this process so far."""
    a, b, c, d, e = _I64(0), _I64(0), C.c_double(0), C.c_double(0), _I64(0)
    check(load().gki_pool_stats(C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(e)))
    return a.value, b.value, c.value, d.value, e.value


def hptr(a):
    """Host pointer of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


class DeviceArray:
    """A typed 1-D buffer in HBM owned by this object (freed on garbage collection)."""

    def __init__(self, n, dtype):
        self.dtype = np.dtype(dtype)
        self.n = int(n)
        p = C.c_void_p()
        check(load().gki_malloc(C.byref(p), self.n * self.dtype.itemsize))
        self.ptr = p

    @classmethod
    def from_host(cls, a):
        a = np.ascontiguousarray(a)
        d = cls(a.size, a.dtype)
        check(load().gki_memcpy_h2d(d.ptr, hptr(a), a.nbytes))
        return d

    def to_host(self, n=None):
        n = self.n if n is None else int(n)
        out = np.empty(n, dtype=self.dtype)
        check(load().gki_memcpy_d2h(hptr(out), self.ptr, out.nbytes))
        return out

    def zero(self):
        check(load().gki_memset(self.ptr, 0, self.n * self.dtype.itemsize))

    @property
    def nbytes(self):
        return self.n * self.dtype.itemsize

    def free(self):
        if self.ptr is not None and self.ptr.value:
            load().gki_free(self.ptr)
            self.ptr = None

    def __len__(self):
        return self.n

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def checksum(self, n=None):
        """(sum mod 2^64, xor) of the first n elements, computed on the device."""
        s, x = _U64(0), _U64(0)
        check(load().gki_column_checksum(self.ptr, self.n if n is None else int(n), self.dtype.itemsize, C.byref(s), C.byref(x)))
        return s.value, x.value

    def view(self, offset, n):
        """Non-owning window [offset, offset + n) of this buffer (keeps the parent alive)."""
        if offset < 0 or n < 0 or offset + n > self.n:
            raise ValueError("view out of range")
        return DeviceView(self, int(offset), int(n))


class DeviceView(DeviceArray):
    def __init__(self, parent, offset, n):
        self.parent, self.dtype, self.n = parent, parent.dtype, n
        self.ptr = C.c_void_p(parent.ptr.value + offset * parent.dtype.itemsize)

    def free(self):
        self.ptr = None
