"""DenseKmerFinder with the reference's constructor and methods (kmer_finder.py:37-244), running
on MI355X through libgki_hip.so.

What `find()` computes is the reference's record multiset: for every base
position and every backward k-base window with at most `max_variant_nodes` non-linear-ref nodes,
one record per distinct window node.  Record ORDER differs from the reference's depth-first order
on branching graphs (it is by end position); on linear graphs it is identical.

On graphs with nodes that have no linear-ref predecessor (a variant inside an alternative allele, multi-node
alleles) and with `only_follow_nodes`, the variant limit also depends on the history before the window; the
kernels then enumerate those histories (`classify_nodes`, include/gki.h GKI_NODE_*).  Where the reference's
`assert len(next_nodes) == 1` (kmer_finder.py:402) fails, find() raises `_lib.NotOneLinearRefSuccessor`
(an AssertionError).

Refused:
  * a critical point (N, c) with 3 <= c < k-1 on a node longer than 2k+3: the reference itself
    emits meaningless hashes there (SURVEY.md 8a' E1 + bulk path)        -> ValueError
  * a k-window over more than GKI_MAX_DEEP_WINDOW_NODES - 2 (12 286) nodes, or more paths into one end node than the
    walk will enumerate -> `_lib.WindowTooDeep` (a RecursionError, which is how the reference ends there)
"""
import ctypes as C
import logging
import numpy as np

from . import _lib
from .critical_graph_paths import CriticalGraphPaths
from .device_graph import DeviceGraph
from .flat_kmers import FlatKmers, FlatKmers2, DeviceFlatKmers
from .graph import GraphArrays


def update_hash(current_base, current_hash, first_base, k, only_add=False):
    """kmer_finder.py:15-34 (kept for API parity; the kernels read hashes straight out of the 2-bit
    sequence instead of rolling them)."""
    current_hash, current_base, first_base = int(current_hash), int(current_base), int(first_base)
    if not isinstance(only_add, bool):
        return current_hash + 4 ** only_add * current_base
    return (current_hash - first_base) // 4 + current_base * 4 ** (k - 1)


def search_roots(g, k):
    """Nodes at which the reference starts a search with no history, besides the critical nodes: the graph's first node
    (kmer_finder.py:208-211 prepends (first_node, 0) when it is not longer than k; longer, it is critical itself).  The
    start node of a LATER chromosome gets no such extra start point: if it is shorter than k it is not critical either
    (critical_graph_paths.py:76-82), no search ever enters it, and every window that touches the nodes before that
    chromosome's first critical point is never emitted -- the classification marks those nodes DEAD (no alive
    predecessor, not a root) and the general kernels drop their windows."""
    return [int(g.first_node)] + [int(s) for s in g.chromosome_start_nodes.values() if g.node_size[s] >= k]


def classify_nodes(g, k, max_variant_nodes, only_follow_nodes=None, critical_nodes=None, on_device=None, always_flags=True):
    """(uint16[n_nodes] GKI_NODE_* flags | history bound << 8, general) -- gki_classify_nodes (include/gki.h): which nodes the order-free
    form of the variant limit (kmer_finder.py:383-417) can stop at when it looks for a history (on the device by
    relaxation sweeps and history rounds, gki_graph_classify_nodes; the host pass takes over without a device or beyond that
    call's stack / budget).  `general` False
    means "at most max_variant_nodes variant nodes in the window" is the whole rule for this graph and the kernels run
    without the flags.  Host pass in topological order; kept on the graph object per (k, limit, follow set)."""
    follow = None
    if only_follow_nodes is not None:
        follow = np.zeros(g.n_nodes, dtype=np.uint8)
        ids = np.fromiter((int(x) for x in only_follow_nodes), dtype=np.int64)
        follow[ids[(ids >= 0) & (ids < g.n_nodes)]] = 1
    # the same clamp as gki_finder_count / gki_forward_* (250): classification and kernels decide on one limit
    crit = np.zeros(0, dtype=np.int32) if critical_nodes is None else np.ascontiguousarray(critical_nodes, dtype=np.int32)
    key = (int(k), min(int(max_variant_nodes), 250), None if follow is None else follow.tobytes(), crit.tobytes(), bool(always_flags), on_device)
    cache = g.__dict__.setdefault("_node_classes", {})
    if key not in cache:
        if len(cache) > 8:
            cache.clear()
        # search roots: chromosome starts and every critical node (each critical point starts a search with no history,
        # kmer_finder.py:190-232)
        roots = np.ascontiguousarray(np.concatenate([np.asarray(search_roots(g, k), dtype=np.int32), crit]))
        flags = np.zeros(g.n_nodes, dtype=np.uint16)
        general, needs_host = C.c_int32(0), C.c_int32(1)
        M = min(int(max_variant_nodes), 250)
        if on_device is None:
            on_device = _lib.device_count() > 0
        if on_device:                                # relaxation sweeps over the resident graph (csrc/gki_classify.hip)
            _lib.check(_lib.load().gki_graph_classify_nodes(
                DeviceGraph.of(g).handle, _lib.hptr(follow), _lib.hptr(roots), len(roots), int(k), M, _lib.hptr(flags),
                int(bool(always_flags)), C.byref(general), C.byref(needs_host)))
        if needs_host.value:                         # no device, or histories beyond the device call's stack / budget: the host pass
            _lib.check(_lib.load().gki_classify_nodes(
                g.n_nodes, _lib.hptr(g.node_size), _lib.hptr(g.edge_start), _lib.hptr(g.edges), _lib.hptr(g.rev_start),
                _lib.hptr(g.rev_edges), _lib.hptr(g.is_ref), _lib.hptr(follow), _lib.hptr(roots), len(roots), int(k),
                M, _lib.hptr(flags), C.byref(general)))
        cache[key] = (flags, bool(general.value))
    return cache[key]


def _resident(owner, table):
    """The device copy of a host table, uploaded once per (owner, table object): `owner` (a graph or a finder) keeps the
    last few, each together with the host array it mirrors (so that the identity stays valid).  A copy that falls out of
    the cache is NOT freed here: run parameters that still point at it hold a reference (`FindParams._keep`), and the
    buffer goes back to the pool with the last of them."""
    _lib.require_device()
    cache = owner.__dict__.setdefault("_resident_tables", {})
    hit = cache.get(id(table))
    if hit is None or hit[0] is not table:
        if len(cache) >= 6:
            cache.pop(next(iter(cache)))
        hit = cache[id(table)] = (table, _lib.DeviceArray.from_host(np.ascontiguousarray(table)))
    return hit[1]


def lossy_table(g, k, crit_nodes, crit_offsets, start_at=None, stop_at=None):
    """uint16[n_nodes] of critical offsets c with 0 < c < k-1 (SURVEY.md 8a' E1), or None.  Raises for the critical
    points THIS run restarts from (numbers [start_at, stop_at), kmer_finder.py:192-205) at which the reference's output
    is undefined."""
    c = np.asarray(crit_offsets).astype(np.int64)
    n = np.asarray(crit_nodes).astype(np.int64)
    sel = (c > 0) & (c < k - 1)
    if not np.any(sel):
        return None
    number = np.arange(len(c))
    if start_at is not None and stop_at is not None and start_at > stop_at:
        stop_at = None                               # the stop point is already behind the run: never met (:220)
    restarted = (number >= (0 if start_at is None else start_at)) & (number < (len(c) if stop_at is None else stop_at))
    undefined = sel & restarted & (c >= 3) & (g.node_size[n] > 2 * k + 3)
    if np.any(undefined):
        raise ValueError(
            "critical point (%d, %d): a single-edge chain of %d bases precedes a node longer than 2k+3; the "
            "reference emits meaningless hashes for this graph (kmer_finder.py:272-273 enters the bulk path with "
            "fewer than k bases), so there is nothing to be bit-exact with"
            % (int(n[undefined][0]), int(c[undefined][0]), k - 1 - int(c[undefined][0])))
    table = np.full(g.n_nodes, 0xFFFF, dtype=np.uint16)
    table[n[sel]] = c[sel]
    return table


class DenseKmerFinder:
    """Finds all possible kmers in graph (kmer_finder.py:37-47 signature)."""

    def __init__(self, graph, k, critical_graph_paths=None, position_id=None, only_save_one_node_per_kmer=False,
                 max_variant_nodes=4, only_store_variant_nodes=False, start_at_critical_path_number=None,
                 stop_at_critical_path_number=None, whitelist=None, only_store_nodes=None, only_follow_nodes=None):
        self._graph = graph
        self._arrays = GraphArrays.from_obgraph(graph)
        self._k = int(k)
        if not 1 <= self._k <= 31:
            raise ValueError("k must be in 1..31")
        self._only_save_one_node_per_kmer = bool(only_save_one_node_per_kmer)
        self._max_variant_nodes = int(max_variant_nodes)
        if only_store_variant_nodes:
            raise NotImplementedError("only_store_variant_nodes is not usable in the reference either "
                                      "(kmer_finder.py:75-76 asserts on an undefined name)")
        self._critical_graph_paths = critical_graph_paths
        self._position_id = position_id
        self._start_at_critical_path_number = start_at_critical_path_number
        self._stop_at_critical_path_number = stop_at_critical_path_number
        self._whitelist = whitelist
        self._only_store_nodes = only_store_nodes
        self._store_table = None
        self._only_follow_nodes = only_follow_nodes   # kmer_finder.py:386-388 (unique_variant_kmers.py:91-96)
        self._params_cache = None
        self._whitelist_device = None
        self._cols = None          # host columns after find()
        self._device = None
        self._finder = None
        self.last_timings = {}

    # ------------------------------------------------------------------ device plumbing
    def _device_graph(self):
        if self._device is None:
            pid = self._position_id
            if pid is None:
                self._device = DeviceGraph.of(self._arrays)
            else:
                g = self._arrays
                base = np.asarray(pid.get(np.arange(g.n_nodes), np.zeros(g.n_nodes, dtype=np.int64))).astype(np.int64)
                self._device = DeviceGraph(g, position_base=base)
        return self._device

    def _finder_handle(self):
        if self._finder is None:
            h = C.c_void_p()
            _lib.check(_lib.load().gki_finder_create(self._device_graph().handle, C.byref(h)))
            self._finder = h
        return self._finder

    def close(self):
        if self._finder is not None:
            _lib.load().gki_finder_destroy(self._finder)
            self._finder = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _params(self):
        """Validated launch parameters; computed once per finder (graph checks are O(nodes + edges))."""
        if self._params_cache is None:
            self._params_cache = (self._make_params(),)
        return self._params_cache[0]

    def _make_params(self):
        g, k = self._arrays, self._k
        if self._critical_graph_paths is None:
            logging.info("Making critical graph paths since it's not specified.")
            self._critical_graph_paths = CriticalGraphPaths.from_graph(g, k)
        cp = self._critical_graph_paths
        crit_nodes = np.asarray(cp.nodes).astype(np.int64)
        flags, general = classify_nodes(g, k, self._max_variant_nodes, self._only_follow_nodes, crit_nodes)
        crit_offsets = np.asarray(cp.offsets).astype(np.int64)
        start_at, stop_at = self._start_at_critical_path_number, self._stop_at_critical_path_number
        lossy = lossy_table(g, k, crit_nodes, crit_offsets, start_at, stop_at)     # (raises for THIS run's restarts; the table
        patch = None                                                                #  itself is the same for every chunk)
        if start_at is not None and 0 < start_at < len(crit_nodes) and crit_offsets[start_at] == 0:
            # A run that STARTS at a critical point at offset 0 is not rewound either (:231-232): none of its windows
            # reaches before that node.  (In a full run the previous search passes through the point and emits them.)
            patch = int(crit_nodes[start_at])
        if lossy is not None or patch is not None:
            # one array object per distinct content, so that its device copy is uploaded once (_resident)
            known = self.__dict__.setdefault("_lossy_tables", {})
            if patch not in known:
                if len(known) >= 3:
                    known.pop(next(iter(known)))
                if lossy is None:
                    lossy = np.full(g.n_nodes, 0xFFFF, dtype=np.uint16)
                if patch is not None:
                    lossy[patch] = 0
                known[patch] = lossy
            lossy = known[patch]
        node_begin, off_begin, node_end, off_end = 0, 0, g.n_nodes, 0
        n_crit = len(crit_nodes)
        chunked = False
        if start_at is not None and start_at > 0:                     # kmer_finder.py:204-205
            chunked = True
            if start_at >= n_crit:
                node_begin, off_begin = g.n_nodes, 0                 # nothing left to start from
            else:
                node_begin, off_begin = int(crit_nodes[start_at]), int(crit_offsets[start_at])
        if stop_at is not None and stop_at < n_crit:                  # :193-194
            if not (start_at is not None and start_at > stop_at):     # stop node already behind us: never met (:220)
                chunked = True
                # The last search of the run ends where it SEES a critical position ahead (`is_critical(node,
                # offset + 1)`, :334-341), which a critical point at offset 0 never is: the run passes through it and
                # ends at the next critical point with offset >= 1 (its records then overlap the next chunk's).
                node_end, off_end = int(crit_nodes[stop_at]), int(crit_offsets[stop_at])
                from_graph_start = (start_at or 0) == 0 and g.node_size[g.first_node] <= k \
                    and g.first_node != node_end                      # the extra starting point of :208-211
                if ((start_at or 0) < stop_at or from_graph_start) and off_end == 0:     # (an empty run stays empty)
                    later = np.nonzero(crit_offsets[stop_at:] >= 1)[0]
                    if len(later):
                        stop_seen = stop_at + int(later[0])
                        node_end, off_end = int(crit_nodes[stop_seen]), int(crit_offsets[stop_seen])
                    else:
                        node_end, off_end = g.n_nodes, 0              # no critical point is seen any more
        rank = None
        if chunked and len(g.edges):
            if "_ids_increase_along_edges" not in g.__dict__:
                src = np.repeat(np.arange(g.n_nodes), np.diff(g.edge_start))
                g.__dict__["_ids_increase_along_edges"] = not np.any(g.edges <= src)
            if not g.__dict__["_ids_increase_along_edges"]:
                # the run is then not an id range: membership by topological rank (every node between two critical
                # points lies between them in any topological order)
                if "_topological_rank" not in g.__dict__:
                    r = np.zeros(g.n_nodes, dtype=np.int32)
                    _lib.check(_lib.load().gki_topological_rank(g.n_nodes, _lib.hptr(g.edge_start), _lib.hptr(g.edges),
                                                                _lib.hptr(r)))
                    g.__dict__["_topological_rank"] = r
                rank = g.__dict__["_topological_rank"]
                if not (start_at is not None and start_at > 0):
                    node_begin, off_begin = int(np.argmin(rank)), 0   # "from the graph start" = from rank 0
        general = general or getattr(self, "_force_general_kernels", False)    # bench.py --general: price of the flags
        store = None
        if self._only_store_nodes is not None:                     # kmer_finder.py:153, applied by the general kernels
            store = self._store_table
            if store is None:
                store = np.zeros(g.n_nodes, dtype=np.uint8)
                ids = np.fromiter((int(x) for x in self._only_store_nodes), dtype=np.int64)
                store[ids[(ids >= 0) & (ids < g.n_nodes)]] = 1
                self._store_table = store
            general = True
        if node_begin >= g.n_nodes:
            return None
        # The per-run tables stay resident in HBM: uploaded once per graph (flags, ranks) or per finder (store set, lossy
        # restarts), handed to gki_finder_count as device pointers -- a chunked run (`index -t N`, shards, one count per
        # chunk) re-uploaded up to 2 + 1 + 2 + 4 bytes per node per chunk before.
        d_flags = _resident(g, flags) if general else None
        d_rank = _resident(g, rank) if rank is not None else None
        d_store = _resident(self, store) if store is not None else None
        d_lossy = _resident(self, lossy) if lossy is not None else None
        INT_MAX = 2 ** 31 - 1
        rank_begin = int(rank[node_begin]) if rank is not None and node_begin < g.n_nodes else INT_MAX
        rank_end = int(rank[node_end]) if rank is not None and node_end < g.n_nodes else INT_MAX
        ptr = lambda d: None if d is None else d.ptr
        p = _lib.FindParams(k, self._max_variant_nodes, int(self._only_save_one_node_per_kmer), 0,
                            node_begin, off_begin, node_end, off_end, None, None, None, None,
                            ptr(d_lossy), ptr(d_rank), ptr(d_flags), ptr(d_store), rank_begin, rank_end)
        p._keep = (lossy, rank, flags, store, d_lossy, d_rank, d_flags, d_store)
        return p

    def set_critical_path_range(self, start_at_critical_path_number, stop_at_critical_path_number):
        """Point this finder at another chunk [start, stop) of critical-path numbers (kmer_finder.py:192-205) -- what the
        reference does with one DenseKmerFinder per chunk (command_line_interface.py:559-565).  The device handle and its
        per-graph arrays are kept, only the run parameters are rebuilt."""
        self._start_at_critical_path_number = start_at_critical_path_number
        self._stop_at_critical_path_number = stop_at_critical_path_number
        self._params_cache = None
        self._cols = None

    def _count(self, layout=0):
        p = self._params()
        if p is None:
            return 0
        p.layout = layout
        n = C.c_int64(0)
        _lib.check(_lib.load().gki_finder_count(self._finder_handle(), C.byref(p), C.byref(n)))
        return n.value

    # ------------------------------------------------------------------ reference API
    def find(self):
        """kmer_finder.py:179-244."""
        lib = _lib.load()
        n = self._count()
        dt = [np.int64, np.int32, np.int16, np.int32, np.float64]
        if n == 0:
            cols = [np.zeros(0, dtype=d) for d in dt]
        else:
            bufs = [_lib.DeviceArray(n, d) for d in dt]
            _lib.check(lib.gki_finder_emit_v2(self._finder_handle(), *[b.ptr for b in bufs]))
            _lib.check(lib.gki_finder_synchronize(self._finder_handle()))
            cols = [b.to_host() for b in bufs]
            for b in bufs:
                b.free()
        kmers, start_nodes, start_offsets, nodes, af = cols
        keep = None
        if self._whitelist is not None:                                # kmer_finder.py:130-132, 362-365
            keep = self._in_whitelist(kmers)
        if keep is not None:
            kmers, start_nodes, start_offsets, nodes, af = (c[keep] for c in (kmers, start_nodes, start_offsets, nodes, af))
        self._cols = dict(kmers=kmers, start_nodes=start_nodes, start_offsets=start_offsets, nodes=nodes, af=af)

    def find_flat_on_device(self, out=None, split_layout=True):
        """find() + get_flat_kmers(v="1") + FlatKmers.from_multiple_flat_kmers dtypes, columns left in
        HBM (the CLI `index` path, command_line_interface.py:559-614).  Returns DeviceFlatKmers.
        split_layout: records whose window lies inside one node first, then the others (GKI_LAYOUT_SPLIT) -- the
        same multiset, written as two dense streams; False gives find()'s by-node order."""
        n = self._count(layout=1 if split_layout else 0)
        if out is None or out.hashes.n < n:
            out = DeviceFlatKmers.allocate(n)
        out.n = n
        if n:
            _lib.check(_lib.load().gki_finder_emit_flat(self._finder_handle(), out.hashes.ptr, out.nodes.ptr,
                                                        out.ref_offsets.ptr, out.allele_frequencies.ptr))
        if self._whitelist is not None and n:
            # kmer_finder.py:130-132, 362-365: keep a record iff `kmer in whitelist`; on the device: membership probe of
            # every hash against the whitelist index, then a stable compaction of the four columns
            self.synchronize()
            flags = self._whitelist_index().contains(out.hashes.view(0, n))
            kept = out.compacted(flags)
            flags.free()
            out.free()
            out = kept
        return out

    def _in_whitelist(self, kmers):
        """bool per k-mer: `kmer in whitelist`, probed on the device."""
        kmers = np.ascontiguousarray(kmers)
        if len(kmers) == 0:
            return np.zeros(0, dtype=bool)
        flags = self._whitelist_index().contains(kmers.astype(np.int64).view(np.uint64))
        keep = flags.to_host(len(kmers)).astype(bool)
        flags.free()
        return keep

    def _whitelist_index(self):
        """The whitelist as a DeviceIndex: a CollisionFreeKmerIndex (what the reference's CLI passes,
        command_line_interface.py:634), a DeviceIndex, or any iterable of k-mer hashes."""
        if self._whitelist_device is None:
            from .collision_free_kmer_index import CollisionFreeKmerIndex, DeviceIndex
            from .flat_kmers import FlatKmers
            wl = self._whitelist
            if isinstance(wl, DeviceIndex):
                self._whitelist_device = wl
            elif isinstance(wl, CollisionFreeKmerIndex):
                self._whitelist_device = wl._device_index()
            else:
                kmers = np.unique(np.fromiter((int(x) for x in wl), dtype=np.int64)).astype(np.uint64)
                z = np.zeros(len(kmers), np.uint32)
                modulo = max(2 * len(kmers) + 1, 1009)
                self._whitelist_device = DeviceIndex.build(
                    DeviceFlatKmers.from_flat_kmers(FlatKmers(kmers, z, z.astype(np.uint64), z.astype(np.float32))),
                    modulo, skip_frequencies=True)
        return self._whitelist_device

    def synchronize(self):
        _lib.check(_lib.load().gki_finder_synchronize(self._finder_handle()))

    def kernel_ms(self, which):
        ms = C.c_float(0)
        _lib.check(_lib.load().gki_finder_kernel_ms(self._finder_handle(), which, C.byref(ms)))
        return ms.value

    def interior_records(self):
        return _lib.load().gki_finder_interior_records(self._finder_handle())

    def find_only_kmers_starting_at_position(self, node, offset):
        """kmer_finder.py:170-177: the first k-mer of every forward path from (node, offset); records accumulate
        over calls like in the reference.  Each call behaves like a call on a fresh reference finder (the
        reference's `_positions_treated` carries over between calls on one object)."""
        self.find_kmers_starting_at_positions([node], [offset])

    def find_kmers_starting_at_positions(self, nodes, offsets):
        """Batched form of the above (one kernel launch for many start positions)."""
        lib = _lib.load()
        g = self._arrays
        graph = self._device_graph()
        n_pos = len(nodes)
        d_nodes = _lib.DeviceArray.from_host(np.ascontiguousarray(nodes, dtype=np.int32))
        d_offs = _lib.DeviceArray.from_host(np.ascontiguousarray(offsets, dtype=np.int32))
        d_start = _lib.DeviceArray(n_pos + 1, np.int64)
        n = C.c_int64(0)
        d_follow = None
        if self._only_follow_nodes is not None:
            mask = np.zeros(g.n_nodes, dtype=np.uint8)
            mask[np.fromiter((int(x) for x in self._only_follow_nodes), dtype=np.int64)] = 1
            d_follow = _lib.DeviceArray.from_host(mask)
        args = (graph.handle, self._k, self._max_variant_nodes, int(self._only_save_one_node_per_kmer),
                None if d_follow is None else d_follow.ptr, d_nodes.ptr, d_offs.ptr, n_pos)
        _lib.check(lib.gki_forward_count(*args, d_start.ptr, C.byref(n)))
        dt = [np.int64, np.int32, np.int16, np.int32, np.float64]
        if n.value:
            bufs = [_lib.DeviceArray(n.value, d) for d in dt]
            _lib.check(lib.gki_forward_emit(*args, d_start.ptr, *[b.ptr for b in bufs]))
            cols = [b.to_host() for b in bufs]
            for b in bufs:
                b.free()
        else:
            cols = [np.zeros(0, dtype=d) for d in dt]
        for b in (d_nodes, d_offs, d_start) + (() if d_follow is None else (d_follow,)):
            b.free()
        kmers, start_nodes, start_offsets, out_nodes, af = cols
        keep = None
        if self._whitelist is not None:
            keep = self._in_whitelist(kmers)
        if self._only_store_nodes is not None:
            sel = np.isin(out_nodes, np.fromiter((int(x) for x in self._only_store_nodes), dtype=np.int64))
            keep = sel if keep is None else keep & sel
        if keep is not None:
            kmers, start_nodes, start_offsets, out_nodes, af = (c[keep] for c in (kmers, start_nodes, start_offsets, out_nodes, af))
        new = dict(kmers=kmers, start_nodes=start_nodes, start_offsets=start_offsets, nodes=out_nodes, af=af)
        if self._cols is None:
            self._cols = new
        else:
            self._cols = {k_: np.concatenate([self._cols[k_], new[k_]]) for k_ in new}

    def _require_found(self):
        if self._cols is None:
            raise RuntimeError("call find() first")
        return self._cols

    def get_found_kmers_and_nodes(self):
        c = self._require_found()
        return c["kmers"], c["nodes"]

    def get_flat_kmers(self, v="2"):
        c = self._require_found()
        if v == "0" or v == "1":
            if v == "1":
                if self._position_id is not None:
                    ref_offsets = self._position_id.get(c["start_nodes"], c["start_offsets"])       # :117
                else:
                    ref_offsets = self._arrays.position_id_base()[c["start_nodes"]] + c["start_offsets"]
            else:
                ref_offsets = np.asarray(self._graph.node_to_ref_offset)[c["start_nodes"]] + c["start_offsets"]  # :119
            return FlatKmers(c["kmers"], c["nodes"], ref_offsets, c["af"])
        return FlatKmers2(c["kmers"], c["start_nodes"], c["start_offsets"], c["nodes"], c["af"])

    @property
    def kmers_found(self):
        """First 500 windows as (None, set(nodes), start_node, kmer) -- debugging aid of the reference
        (kmer_finder.py:148-168), here in end-position order."""
        c = self._require_found()
        out, i, n = [], 0, len(c["kmers"])
        while i < n and len(out) < 500:
            j = i + 1
            while (j < n and c["kmers"][j] == c["kmers"][i] and c["start_nodes"][j] == c["start_nodes"][i]
                   and c["start_offsets"][j] == c["start_offsets"][i] and c["nodes"][j] > c["nodes"][j - 1]):
                j += 1
            out.append((None, set(int(x) for x in c["nodes"][i:j]), int(c["start_nodes"][i]), int(c["kmers"][i])))
            i = j
        return out
