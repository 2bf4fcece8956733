"""ReverseKmerIndex with the reference's attributes, methods and .npz keys (reverse_kmer_index.py:5-83);
`from_flat_kmers` sorts on MI355X (gki_reverse_index_build: the index build's stable radix sort keyed on node).

Records of a node keep their input order (the reference's np.argsort leaves it unspecified).
"""
import logging
import os
import numpy as np

from . import _lib
from .flat_kmers import DeviceFlatKmers


def _as_u64(a, what):
    a = np.ascontiguousarray(np.asarray(a))
    if a.dtype.kind not in "iu":
        raise TypeError("%s must be an integer array, got %s" % (what, a.dtype))
    return a.view(np.uint64) if a.dtype.itemsize == 8 else a.astype(np.uint64)


class ReverseKmerIndex:
    """node -> the k-mers (and ref positions) of its records: records grouped by node, `nodes_to_index_positions[node]`
    the first record of the group and `nodes_to_n_hashes[node]` its length (reverse_kmer_index.py:5-45)."""
    properties = {"nodes_to_index_positions", "nodes_to_n_hashes", "hashes", "ref_positions"}

    def __init__(self, nodes_to_index_positions=None, nodes_to_n_hashes=None, hashes=None, ref_positions=None):
        self.nodes_to_index_positions = nodes_to_index_positions
        self.nodes_to_n_hashes = nodes_to_n_hashes
        self.hashes = hashes
        self.ref_positions = ref_positions

    def _group(self, node):
        """slice of the records of `node` (IndexError for a node beyond the table, as in the reference)"""
        first = int(self.nodes_to_index_positions[node])
        return slice(first, first + int(self.nodes_to_n_hashes[node]))

    def get_node_kmers(self, node):
        group = self._group(node)
        return self.hashes[group] if group.stop > group.start else []

    def get_node_kmers_and_ref_positions(self, node):
        try:
            group = self._group(node)
        except IndexError:
            logging.error("Invalid node %d" % node)
            raise
        if group.stop == group.start:
            return [[], []]
        return self.hashes[group], self.ref_positions[group]

    def __str__(self):
        return "\n".join("%s: %s" % (name, getattr(self, name)) for name in sorted(self.properties)) + "\n"

    def to_file(self, file_name):
        np.savez(file_name, **{name: getattr(self, name) for name in self.properties})

    @classmethod
    def from_file(cls, file_name):
        path = file_name if os.path.exists(file_name) else file_name + ".npz"
        with np.load(path) as data:
            return cls(**{name: data[name] for name in cls.properties})

    @classmethod
    def from_flat_kmers(cls, flat_kmers):
        """reverse_kmer_index.py:47-83.  Accepts a FlatKmers (host columns) or a DeviceFlatKmers."""
        _lib.require_device()
        if isinstance(flat_kmers, DeviceFlatKmers):
            n = flat_kmers.n
            if n == 0:
                raise ValueError("zero-size array to reduction operation maximum which has no identity")
            d_nodes, d_kmers, d_refs = flat_kmers.nodes, flat_kmers.hashes, flat_kmers.ref_offsets
            max_node = int(d_nodes.to_host(n).max())
            kmer_dtype = ref_dtype = np.dtype(np.uint64)
            owned = []
        else:
            nodes = np.asarray(flat_kmers._nodes)
            max_node = int(np.max(nodes))                   # raises on an empty input like the reference (:55)
            if nodes.min() < 0 or max_node >= 2 ** 32:
                raise ValueError("node ids must be in 0..2^32-1")
            kmer_dtype, ref_dtype = np.asarray(flat_kmers._hashes).dtype, np.asarray(flat_kmers._ref_offsets).dtype
            n = len(nodes)
            d_nodes = _lib.DeviceArray.from_host(np.ascontiguousarray(nodes).astype(np.uint32))
            d_kmers = _lib.DeviceArray.from_host(_as_u64(flat_kmers._hashes, "hashes"))
            d_refs = _lib.DeviceArray.from_host(_as_u64(flat_kmers._ref_offsets, "ref_offsets"))
            owned = [d_nodes, d_kmers, d_refs]
        logging.info("Max node: %d" % max_node)
        n_nodes = max_node + 1
        index_pos, n_hashes = _lib.DeviceArray(n_nodes, np.uint32), _lib.DeviceArray(n_nodes, np.uint16)
        out_kmers, out_refs = _lib.DeviceArray(n, np.uint64), _lib.DeviceArray(n, np.uint64)
        try:
            _lib.check(_lib.load().gki_reverse_index_build(d_nodes.ptr, d_kmers.ptr, d_refs.ptr, n, n_nodes, index_pos.ptr,
                                                           n_hashes.ptr, out_kmers.ptr, out_refs.ptr))
            cols = []
            for d, dt in ((out_kmers, kmer_dtype), (out_refs, ref_dtype)):
                h = d.to_host(n)
                cols.append(h.view(dt) if dt.itemsize == 8 else h.astype(dt))
            return cls(index_pos.to_host(), n_hashes.to_host(), cols[0], cols[1])
        finally:
            for a in owned + [index_pos, n_hashes, out_kmers, out_refs]:
                a.free()
