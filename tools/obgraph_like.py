"""An object with obgraph.Graph's surface over a GraphArrays -- for tests and for bench.py's `setup_s` timing of the
adapter `GraphArrays.from_obgraph`: the accessor methods the reference calls and,
optionally, the whole-array attributes obgraph keeps beside them (ragged `edges` / `numeric_node_sequences` with
npstructures.RaggedArray's surface).  Zero-copy views; nothing here is on the product path."""
import numpy as np


class _Shape:
    def __init__(self, row_start):
        self.starts = row_start[:-1]
        self.ends = row_start[1:]
        self.lengths = np.diff(row_start)


class _Ragged:
    def __init__(self, row_start, flat):
        self.shape = _Shape(row_start)
        self._data = flat

    def ravel(self):
        return self._data

    def __len__(self):
        return len(self.shape.lengths)

    def __getitem__(self, row):
        return self._data[self.shape.starts[row]:self.shape.ends[row]]


class ObgraphLike:
    def __init__(self, g, with_arrays=True):
        self._g = g
        self.nodes = g.node_size
        self.chromosome_start_nodes = g.chromosome_start_nodes
        self.node_to_ref_offset = g.node_to_ref_offset
        if with_arrays:
            self.edges = _Ragged(g.edge_start, g.edges)
            self.numeric_node_sequences = _Ragged(g.seq_start, g.seq)

    def max_node_id(self):
        return self._g.n_nodes - 1

    def get_first_node(self):
        return self._g.get_first_node()

    def get_edges(self, node):
        return self._g.get_edges(node)

    def get_numeric_node_sequence(self, node):
        return self._g.get_numeric_node_sequence(node)

    def get_reverse_edges_hashtable(self):
        return self._g.get_reverse_edges_hashtable()

    def is_linear_ref_node_or_linear_ref_dummy_node(self, node):
        if np.ndim(node) > 0:
            return self._g.is_ref[np.asarray(node)] != 0
        return self._g.is_linear_ref_node_or_linear_ref_dummy_node(node)

    def get_node_allele_frequencies(self, nodes):
        return self._g.get_node_allele_frequencies(nodes)
