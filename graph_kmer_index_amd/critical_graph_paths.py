"""CriticalGraphPaths with the reference's interface (critical_graph_paths.py:5-104).

`from_graph` runs the walk in libgki_hip.so's host-side `gki_critical_paths` on the flat graph
arrays (O(#linear nodes), sequential -- a graph-preparation step, not a per-base one)."""
import ctypes as C
import numpy as np

from . import _lib
from .graph import GraphArrays


class CriticalGraphPaths:
    def __init__(self, nodes, offsets, index=None):
        self.nodes = nodes
        self.offsets = offsets
        self._index = index

    def _make_index(self):
        if len(self.nodes) == 0:
            self._index = np.zeros(0)
            return
        self._index = np.zeros(int(np.max(self.nodes)) + 1, dtype=np.uint16)
        self._index[self.nodes] = self.offsets

    @classmethod
    def empty(cls):
        return cls(np.array([]), np.array([]), np.array([]))

    def is_critical(self, node, offset):
        if self._index is None:
            self._make_index()
        if node >= len(self._index):
            return False
        return self._index[node] == offset

    def __len__(self):
        return len(self.nodes)

    def __iter__(self):
        return ((node, offset) for node, offset in zip(self.nodes, self.offsets))

    @classmethod
    def from_graph(cls, graph, k):
        g = GraphArrays.from_obgraph(graph)
        lib = _lib.load()
        chrom = np.ascontiguousarray(list(g.chromosome_start_nodes.values()), dtype=np.int32)
        nodes = np.zeros(g.n_nodes, dtype=np.uint32)
        offsets = np.zeros(g.n_nodes, dtype=np.uint16)
        n = C.c_int64(0)
        rc = lib.gki_critical_paths(g.n_nodes, _lib.hptr(g.node_size), _lib.hptr(g.edge_start), _lib.hptr(g.edges),
                                    _lib.hptr(g.rev_start), _lib.hptr(g.is_ref), _lib.hptr(chrom), len(chrom), int(k),
                                    _lib.hptr(nodes), _lib.hptr(offsets), C.byref(n))
        if rc != 0:
            # the reference raises here too (critical_graph_paths.py:96-100 Exception, :104 OverflowError)
            raise Exception(lib.gki_last_error().decode())
        return cls(nodes[:n.value].copy(), offsets[:n.value].copy())
