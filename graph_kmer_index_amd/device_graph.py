"""Handle of a graph resident in HBM (wraps gki_graph_* of include/gki.h)."""
import ctypes as C
import numpy as np

from . import _lib
from .graph import GraphArrays


class DeviceGraph:
    def __init__(self, graph, position_base=None, d_seq=None):
        """graph: GraphArrays or an obgraph-like object.  position_base: int64[n_nodes] position id of
        (node, 0) (default: exclusive prefix sum of node sizes).  d_seq: DeviceArray with the uint8
        sequence already in HBM (then graph.seq may be None)."""
        _lib.require_device()
        g = GraphArrays.from_obgraph(graph)
        self.arrays = g
        lib = _lib.load()
        pb = None if position_base is None else np.ascontiguousarray(position_base, dtype=np.int64)
        h = C.c_void_p()
        args = (g.n_nodes, _lib.hptr(g.node_size))
        tail = (int(g.seq_start[-1]), _lib.hptr(g.edge_start), _lib.hptr(g.edges), _lib.hptr(g.rev_start),
                _lib.hptr(g.rev_edges), len(g.edges), _lib.hptr(g.is_ref), _lib.hptr(g.allele_freq), _lib.hptr(pb))
        if d_seq is not None:
            self._d_seq = d_seq
            _lib.check(lib.gki_graph_create_dseq(C.byref(h), *args, d_seq.ptr, *tail))
        else:
            _lib.check(lib.gki_graph_create(C.byref(h), *args, _lib.hptr(g.seq), *tail))
        self.handle = h

    @classmethod
    def of(cls, graph):
        """Device handle cached on the GraphArrays object (default position ids)."""
        g = GraphArrays.from_obgraph(graph)
        if g._device is None:
            g._device = cls(g)
        return g._device

    def prepare(self):
        _lib.check(_lib.load().gki_graph_prepare(self.handle))

    def close(self):
        if self.handle is not None:
            _lib.load().gki_graph_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
