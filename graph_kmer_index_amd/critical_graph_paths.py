"""CriticalGraphPaths with the reference's interface (critical_graph_paths.py:5-104).

`from_graph` runs the walk on the device over the resident graph (`gki_graph_critical_paths`: jump tables + prefix
sums, csrc/gki_critical.hip); the sequential host walk of the library (`gki_critical_paths`) serves hosts without a
device, e.g. a machine that only prepares the chunk table."""
import ctypes as C
import numpy as np

from . import _lib
from .graph import GraphArrays


class CriticalGraphPaths:
    """nodes[i], offsets[i] = the i-th critical point in graph order.  `is_critical` answers from a node-sorted copy
    by binary search; like the reference's dense table (critical_graph_paths.py:11-34) it reports offset 0 as critical
    for every non-critical node up to the largest critical node id."""

    def __init__(self, nodes, offsets, index=None):
        self.nodes = nodes
        self.offsets = offsets
        self._index = index                    # accepted for signature compatibility; the lookup below does not use it
        self._by_node = None

    @classmethod
    def empty(cls):
        return cls(np.array([]), np.array([]), np.array([]))

    def _lookup(self):
        if self._by_node is None:
            nodes = np.asarray(self.nodes).astype(np.int64)
            order = np.argsort(nodes, kind="stable")
            # a node listed twice keeps its LAST offset, as the reference's table assignment does
            self._by_node = (nodes[order], np.asarray(self.offsets).astype(np.int64)[order])
        return self._by_node

    def is_critical(self, node, offset):
        nodes, offsets = self._lookup()
        if len(nodes) == 0 or node > nodes[-1]:
            return False
        hi = int(np.searchsorted(nodes, node, side="right"))
        listed = hi > 0 and nodes[hi - 1] == node
        return (int(offsets[hi - 1]) if listed else 0) == offset

    def __iter__(self):
        return iter(zip(self.nodes, self.offsets))

    def __len__(self):
        return len(self.nodes)

    @classmethod
    def from_graph(cls, graph, k, on_device=None):
        """critical_graph_paths.py:42-104.  on_device=None: the data-parallel walk over the graph resident in HBM
        (gki_graph_critical_paths; the graph handle is created on first use and cached on the graph, the finder needs
        it anyway) when an MI355X is visible and the graph has at most 64 chromosomes, else the sequential host walk
        of the library (gki_critical_paths).  Both give the reference's points in its order and raise where it does."""
        g = GraphArrays.from_obgraph(graph)
        lib = _lib.load()
        chrom = np.ascontiguousarray(list(g.chromosome_start_nodes.values()), dtype=np.int32)
        nodes = np.zeros(g.n_nodes, dtype=np.uint32)
        offsets = np.zeros(g.n_nodes, dtype=np.uint16)
        n = C.c_int64(0)
        if on_device is None:
            on_device = len(chrom) <= 64 and _lib.device_count() > 0
        if on_device:
            from .device_graph import DeviceGraph
            rc = lib.gki_graph_critical_paths(DeviceGraph.of(g).handle, _lib.hptr(chrom), len(chrom), int(k), _lib.hptr(nodes),
                                              _lib.hptr(offsets), C.byref(n))
        else:
            rc = lib.gki_critical_paths(g.n_nodes, _lib.hptr(g.node_size), _lib.hptr(g.edge_start), _lib.hptr(g.edges),
                                        _lib.hptr(g.rev_start), _lib.hptr(g.is_ref), _lib.hptr(chrom), len(chrom), int(k),
                                        _lib.hptr(nodes), _lib.hptr(offsets), C.byref(n))
        if rc != 0:
            # the reference raises here too (critical_graph_paths.py:96-100 Exception, :104 OverflowError)
            raise Exception(lib.gki_last_error().decode())
        return cls(nodes[:n.value].copy(), offsets[:n.value].copy())
