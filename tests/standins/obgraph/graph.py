"""TEST-ONLY stand-in for `obgraph.Graph` (see package docstring).

Semantics assumed: numeric base codes a=0 c=1 g=2 t=3;
an empty node is a "linear-ref dummy" iff none of its siblings (the other
successors of any of its predecessors) is a linear-ref node; allele
frequencies default to 1.0; one chromosome starting at the first linear node unless
`chromosome_start_nodes` lists the start node of every chromosome.
"""
import numpy as np


class VariantNotFoundException(Exception):
    pass


_CODE = {"a": 0, "c": 1, "g": 2, "t": 3, "n": 0, "m": 0}


class _RaggedShape:
    def __init__(self, lengths):
        self.lengths = np.asarray(lengths, dtype=np.int64)
        self.ends = np.cumsum(self.lengths)
        self.starts = self.ends - self.lengths


class _Ragged:
    """The slice of npstructures.RaggedArray's surface that obgraph's `edges` / `numeric_node_sequences` attributes show:
    flat data (`ravel()`, `_data`), `shape.starts` / `shape.lengths` / `shape.ends`, row indexing."""

    def __init__(self, rows, dtype):
        self.shape = _RaggedShape([len(r) for r in rows])
        self._data = np.concatenate([np.asarray(r, dtype=dtype) for r in rows]) if len(rows) else np.zeros(0, dtype)
        if self._data.dtype != dtype:
            self._data = self._data.astype(dtype)

    def ravel(self):
        return self._data

    def __len__(self):
        return len(self.shape.lengths)

    def __getitem__(self, row):
        return self._data[self.shape.starts[row]:self.shape.ends[row]]


class Graph:
    def __init__(self, node_sequences, edges, linear_ref_nodes, allele_frequencies=None, chromosome_start_nodes=None):
        self._seq = {int(n): np.array([_CODE[c] for c in s.lower()], dtype=np.uint8)
                     for n, s in node_sequences.items()}
        self._edges = {int(n): [int(x) for x in e] for n, e in edges.items()}
        self._linear = set(int(n) for n in linear_ref_nodes)
        self._linear_list = [int(n) for n in linear_ref_nodes]
        self.nodes = np.zeros(max(self._seq) + 1, dtype=np.int32)
        for n, s in self._seq.items():
            self.nodes[n] = len(s)
        self._rev = {n: [] for n in self._seq}
        for n, succ in self._edges.items():
            for m in succ:
                self._rev[m].append(n)
        self._af = np.ones(max(self._seq) + 1, dtype=float)
        if allele_frequencies is not None:
            for n, f in allele_frequencies.items():
                self._af[n] = f
        self._ref_or_dummy = None
        self.make_linear_ref_node_and_ref_dummy_node_index()
        # whole-array attributes, as obgraph keeps them beside its accessors (rows of absent ids are empty)
        n_ids = max(self._seq) + 1
        self.edges = _Ragged([self._edges.get(n, []) for n in range(n_ids)], np.int32)
        self.numeric_node_sequences = _Ragged([self._seq.get(n, []) for n in range(n_ids)], np.uint8)
        first = self._linear_list[0] if self._linear_list else min(self._seq)
        self.chromosome_start_nodes = {1: first} if chromosome_start_nodes is None else \
            {i + 1: int(n) for i, n in enumerate(chromosome_start_nodes)}
        # linear-ref offset of every linear node (cumulative along the linear path)
        self.node_to_ref_offset = np.zeros(max(self._seq) + 2, dtype=np.int64)
        off = 0
        for n in self._linear_list:
            self.node_to_ref_offset[n] = off
            off += len(self._seq[n])

    @classmethod
    def from_dicts(cls, node_sequences, edges, linear_ref_nodes, allele_frequencies=None, chromosome_start_nodes=None):
        return cls(node_sequences, edges, linear_ref_nodes, allele_frequencies, chromosome_start_nodes)

    def make_linear_ref_node_and_ref_dummy_node_index(self):
        flag = {}
        for n, s in self._seq.items():
            if n in self._linear:
                flag[n] = True
            elif len(s) == 0:
                siblings = set()
                for p in self._rev[n]:
                    siblings.update(self._edges.get(p, []))
                siblings.discard(n)
                flag[n] = not any(x in self._linear for x in siblings)
            else:
                flag[n] = False
        self._ref_or_dummy = flag

    # --- accessors used by the reference hot path -------------------------
    def linear_ref_nodes(self):
        return set(self._linear)

    def get_first_node(self):
        cand = [n for n in self._seq if len(self._rev[n]) == 0]
        return min(cand)

    def get_node_size(self, node):
        return len(self._seq[int(node)])

    def get_numeric_base_sequence(self, node, offset):
        return int(self._seq[int(node)][offset])

    def get_numeric_node_sequence(self, node):
        return self._seq[int(node)].copy()

    def get_edges(self, node):
        return list(self._edges.get(int(node), []))

    def is_linear_ref_node_or_linear_ref_dummy_node(self, node):
        if np.ndim(node) > 0:                    # (obgraph answers from an index array: vectorised for free)
            return np.array([bool(self._ref_or_dummy.get(int(n), False)) for n in node])
        return self._ref_or_dummy[int(node)]

    def get_node_allele_frequencies(self, nodes):
        return self._af[np.asarray(nodes, dtype=np.int64)]

    def get_node_allele_frequency(self, node):
        return float(self._af[int(node)])

    def get_reverse_edges_hashtable(self):
        return {n: list(v) for n, v in self._rev.items()}

    def max_node_id(self):
        return max(self._seq)

    def get_all_nodes(self):
        return sorted(self._seq)
