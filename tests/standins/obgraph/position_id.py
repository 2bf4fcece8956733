"""TEST-ONLY stand-in for `obgraph.position_id.PositionId` (see package docstring).

Assumption (parity-unpinned, SURVEY.md section 8c caveat 4): a position id is the
exclusive cumulative node size over node ids plus the offset.
"""
import numpy as np


class PositionId:
    def __init__(self, index):
        self._index = index

    @classmethod
    def from_graph(cls, graph):
        sizes = np.asarray(graph.nodes, dtype=np.int64)
        index = np.zeros(len(sizes) + 1, dtype=np.int64)
        index[1:] = np.cumsum(sizes)
        return cls(index)

    def get(self, nodes, offsets):
        return self._index[np.asarray(nodes, dtype=np.int64)] + np.asarray(offsets, dtype=np.int64)
