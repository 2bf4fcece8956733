"""Flat-array graph layout consumed by the HIP kernels (and by the oracle).

The reference walks an `obgraph.Graph` through per-node accessor methods
(/root/reference/graph_kmer_index/kmer_finder.py:50,62,259,279,350,384,138,143,374
and critical_graph_paths.py:46,52-53,62 -- SURVEY.md section 8b).  The device
path needs the same information as a handful of contiguous arrays that can be
uploaded to HBM once:

  node_size   int32 [n_nodes]      bases per node id (0 = empty "dummy" node or unused id)
  seq_start   int64 [n_nodes+1]    exclusive prefix sum of node_size: global base index of (node, 0)
  seq         uint8 [n_bases]      numeric bases a/n/m=0 c=1 g=2 t=3, nodes concatenated in id order
  edge_start  int64 [n_nodes+1]    CSR successors, in `get_edges` order
  edges       int32 [n_edges]
  rev_start   int64 [n_nodes+1]    CSR predecessors (ascending source id, stable)
  rev_edges   int32 [n_edges]
  is_ref      uint8 [n_nodes]      is_linear_ref_node_or_linear_ref_dummy_node
  allele_freq float64 [n_nodes]
  exists      uint8 [n_nodes]      id is a real node of the graph

`GraphArrays` also answers the obgraph accessor methods itself, so the same
object can be handed to code written against obgraph (used by the golden-vector
generator to drive the reference on synthetic graphs).
"""
import logging

import numpy as np

_LETTER_TO_CODE = np.zeros(256, dtype=np.uint8)
for _ch, _c in (("c", 1), ("g", 2), ("t", 3)):
    _LETTER_TO_CODE[ord(_ch)] = _c
    _LETTER_TO_CODE[ord(_ch.upper())] = _c


def encode_letters(text):
    """bytes/str -> uint8 codes (a,n,m and anything else -> 0; flat_kmers.py:134-145)."""
    if isinstance(text, str):
        text = text.encode("ascii")
    return _LETTER_TO_CODE[np.frombuffer(text, dtype=np.uint8)]


def _csr_from_lists(n_nodes, adjacency):
    start = np.zeros(n_nodes + 1, dtype=np.int64)
    for n, lst in adjacency.items():
        start[n + 1] = len(lst)
    np.cumsum(start, out=start)
    flat = np.zeros(int(start[-1]), dtype=np.int32)
    for n, lst in adjacency.items():
        flat[start[n]:start[n] + len(lst)] = lst
    return start, flat


def _ragged_rows(r, n_rows):
    """(row_start int64[n_rows + 1], flat) of a ragged array with npstructures.RaggedArray's surface -- flat data from
    `ravel()` or `_data`, row bounds from `shape.starts` / `shape.lengths` (`_shape` in older versions) -- cut or padded
    with empty rows to n_rows, the rows laid out one after the other.  None when the object offers no such thing."""
    if r is None:
        return None
    shape = getattr(r, "shape", None)
    if not hasattr(shape, "starts"):
        shape = getattr(r, "_shape", None)
    starts, lengths = getattr(shape, "starts", None), getattr(shape, "lengths", None)
    flat = r.ravel() if hasattr(r, "ravel") else getattr(r, "_data", None)
    if starts is None or lengths is None or flat is None:
        return None
    starts, lengths, flat = np.asarray(starts, dtype=np.int64), np.asarray(lengths, dtype=np.int64), np.asarray(flat)
    if starts.ndim != 1 or starts.shape != lengths.shape or flat.ndim != 1:
        return None
    m = min(n_rows, len(lengths))
    row_len = np.zeros(n_rows, dtype=np.int64)
    row_len[:m] = lengths[:m]
    row_start = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(row_len, out=row_start[1:])
    total = int(row_start[-1])
    if m and (lengths[:m].min() < 0 or starts[:m].min() < 0 or int((starts[:m] + lengths[:m]).max()) > len(flat)):
        return None
    if np.array_equal(starts[:m], row_start[:m]):                   # the usual case: rows already back to back
        return row_start, flat[:total]
    src = np.repeat(starts[:m] - row_start[:m], row_len[:m]) + np.arange(total, dtype=np.int64)
    return row_start, flat[src]


def _reverse_csr(n_nodes, edge_start, edges):
    src = np.repeat(np.arange(n_nodes, dtype=np.int32), np.diff(edge_start))
    order = np.argsort(edges, kind="stable")
    rev_edges = src[order].astype(np.int32)
    counts = np.bincount(edges, minlength=n_nodes)
    rev_start = np.zeros(n_nodes + 1, dtype=np.int64)
    np.cumsum(counts, out=rev_start[1:])
    return rev_start, rev_edges


class GraphArrays:
    def __init__(self, node_size, seq, edge_start, edges, is_ref, allele_freq=None,
                 exists=None, first_node=None, chromosome_start_nodes=None,
                 node_to_ref_offset=None, rev_start=None, rev_edges=None):
        self.node_size = np.ascontiguousarray(node_size, dtype=np.int32)
        self.n_nodes = len(self.node_size)
        self.seq_start = np.zeros(self.n_nodes + 1, dtype=np.int64)
        np.cumsum(self.node_size, out=self.seq_start[1:])
        self.seq = np.ascontiguousarray(seq, dtype=np.uint8)
        assert len(self.seq) == self.seq_start[-1]
        self.edge_start = np.ascontiguousarray(edge_start, dtype=np.int64)
        self.edges = np.ascontiguousarray(edges, dtype=np.int32)
        assert len(self.edge_start) == self.n_nodes + 1
        if rev_start is None:
            rev_start, rev_edges = _reverse_csr(self.n_nodes, self.edge_start, self.edges)
        self.rev_start = np.ascontiguousarray(rev_start, dtype=np.int64)
        self.rev_edges = np.ascontiguousarray(rev_edges, dtype=np.int32)
        self.is_ref = np.ascontiguousarray(is_ref, dtype=np.uint8)
        if allele_freq is None:
            allele_freq = np.ones(self.n_nodes, dtype=np.float64)
        self.allele_freq = np.ascontiguousarray(allele_freq, dtype=np.float64)
        if exists is None:
            exists = np.ones(self.n_nodes, dtype=np.uint8)
        self.exists = np.ascontiguousarray(exists, dtype=np.uint8)
        if first_node is None:
            indeg = np.diff(self.rev_start)
            cand = np.nonzero((indeg == 0) & (self.exists != 0))[0]
            first_node = int(cand[0])
        self.first_node = int(first_node)
        if chromosome_start_nodes is None:
            chromosome_start_nodes = [self.first_node]
        self._chromosome_start_nodes = [int(x) for x in chromosome_start_nodes]
        self.node_to_ref_offset = node_to_ref_offset
        self._device = None   # cache slot owned by graph_kmer_index_amd (device handle)

    # ------------------------------------------------------------------ builders
    @classmethod
    def from_dicts(cls, node_sequences, edges, linear_ref_nodes, allele_frequencies=None, chromosome_start_nodes=None):
        """Same call shape as obgraph's `Graph.from_dicts` (tests/test_kmer_finder.py:12-16).

        An empty node counts as linear-ref dummy iff none of its siblings is a
        linear-ref node (assumption shared with the test stand-in)."""
        n_nodes = max(int(n) for n in node_sequences) + 1
        node_size = np.zeros(n_nodes, dtype=np.int32)
        exists = np.zeros(n_nodes, dtype=np.uint8)
        for n, s in node_sequences.items():
            node_size[int(n)] = len(s)
            exists[int(n)] = 1
        seq_start = np.zeros(n_nodes + 1, dtype=np.int64)
        np.cumsum(node_size, out=seq_start[1:])
        seq = np.zeros(int(seq_start[-1]), dtype=np.uint8)
        for n, s in node_sequences.items():
            seq[seq_start[int(n)]:seq_start[int(n) + 1]] = encode_letters(s)
        adjacency = {int(n): [int(x) for x in e] for n, e in edges.items()}
        edge_start, flat_edges = _csr_from_lists(n_nodes, adjacency)
        linear = set(int(n) for n in linear_ref_nodes)
        rev = {n: [] for n in range(n_nodes)}
        for n, succ in adjacency.items():
            for m in succ:
                rev[m].append(n)
        is_ref = np.zeros(n_nodes, dtype=np.uint8)
        for n in range(n_nodes):
            if not exists[n]:
                continue
            if n in linear:
                is_ref[n] = 1
            elif node_size[n] == 0:
                siblings = set()
                for p in rev[n]:
                    siblings.update(adjacency.get(p, []))
                siblings.discard(n)
                is_ref[n] = 0 if any(x in linear for x in siblings) else 1
        af = np.ones(n_nodes, dtype=np.float64)
        if allele_frequencies is not None:
            for n, f in allele_frequencies.items():
                af[int(n)] = f
        linear_list = [int(n) for n in linear_ref_nodes]
        ntro = np.zeros(n_nodes + 1, dtype=np.int64)
        off = 0
        for n in linear_list:
            ntro[n] = off
            off += int(node_size[n])
        indeg0 = [n for n in range(n_nodes) if exists[n] and len(rev[n]) == 0]
        first = min(indeg0)
        chrom = [linear_list[0]] if linear_list else [first]
        if chromosome_start_nodes is not None:
            chrom = [int(n) for n in chromosome_start_nodes]
        return cls(node_size, seq, edge_start, flat_edges, is_ref, af, exists, first, chrom, ntro)

    @classmethod
    def from_obgraph(cls, graph, check_nodes=2000):
        """The flat arrays of an obgraph `Graph`.

        The reference reads the graph node by node inside its search (kmer_finder.py:259, :279, :350, :384, :138, :143);
        the device wants it whole.  Two ways:
          * whole arrays, when the object offers them (duck-typed, obgraph's private layout is not verifiable offline):
            ragged `edges` / `numeric_node_sequences` with npstructures' RaggedArray surface (`ravel()` or `_data`,
            `shape.starts` / `shape.lengths`), the per-node accessors called with an array of all nodes
            (`is_linear_ref_node_or_linear_ref_dummy_node`, `get_node_allele_frequencies`).  What they give is compared
            with the accessor methods on `check_nodes` random nodes (plus the first and last ones) before it is trusted;
          * otherwise -- or when that comparison fails -- the accessor methods node by node: two Python
            loops, minutes at the 1.5e7 nodes of a human graph."""
        if isinstance(graph, cls):
            return graph
        fast = cls._from_obgraph_arrays(graph)
        if fast is not None:
            bad = cls._first_disagreement(graph, fast, check_nodes)
            if bad is None:
                return fast
            logging.warning("from_obgraph: the graph's whole arrays disagree with its accessors (%s): reading node by node" % bad)
        else:
            logging.info("from_obgraph: the graph object offers no whole arrays this adapter knows (ragged `edges` and "
                         "`numeric_node_sequences`): reading it node by node through its accessor methods")
        return cls._from_obgraph_accessors(graph)

    @classmethod
    def _obgraph_head(cls, graph):
        n_nodes = int(graph.max_node_id()) + 1
        sizes = np.asarray(graph.nodes)
        node_size = np.zeros(n_nodes, dtype=np.int32)
        m = min(n_nodes, len(sizes))
        node_size[:m] = sizes[:m]
        return n_nodes, node_size

    @classmethod
    def _obgraph_tail(cls, graph, n_nodes, node_size, seq, edge_start, flat_edges, is_ref, exists):
        af = np.asarray(graph.get_node_allele_frequencies(np.arange(n_nodes)), dtype=np.float64)
        chrom = list(graph.chromosome_start_nodes.values())
        ntro = getattr(graph, "node_to_ref_offset", None)
        return cls(node_size, seq, edge_start, flat_edges, is_ref, af, exists, int(graph.get_first_node()), chrom, ntro)

    @classmethod
    def _from_obgraph_arrays(cls, graph):
        """None unless the object offers everything as whole arrays."""
        try:
            n_nodes, node_size = cls._obgraph_head(graph)
            e = _ragged_rows(getattr(graph, "edges", None), n_nodes)
            q = _ragged_rows(getattr(graph, "numeric_node_sequences", None), n_nodes)
            if e is None or q is None:
                return None
            edge_start, flat_edges = e
            seq_start, seq = q
            if not np.array_equal(np.diff(seq_start), node_size):
                return None
            if len(flat_edges) and (flat_edges.min() < 0 or flat_edges.max() >= n_nodes):
                return None
            ref = np.asarray(graph.is_linear_ref_node_or_linear_ref_dummy_node(np.arange(n_nodes)))
            if ref.shape != (n_nodes,):
                return None
            has_pred = np.bincount(flat_edges, minlength=n_nodes) > 0
            exists = ((node_size > 0) | (np.diff(edge_start) > 0) | has_pred).astype(np.uint8)
            is_ref = ((ref != 0) & (exists != 0)).astype(np.uint8)
            return cls._obgraph_tail(graph, n_nodes, node_size, seq, edge_start, flat_edges, is_ref, exists)     # (no copies: views of the object's arrays)
        except (AttributeError, TypeError, ValueError, IndexError, KeyError):
            return None

    @staticmethod
    def _first_disagreement(graph, arrays, check_nodes):
        """A node on which the accessor methods and `arrays` differ (as text), or None."""
        n = arrays.n_nodes
        rng = np.random.default_rng(n)
        nodes = np.unique(np.concatenate([rng.integers(0, n, size=min(n, int(check_nodes))), np.arange(min(n, 8)),
                                          np.arange(max(0, n - 8), n)]))
        for v in nodes.tolist():
            if [int(x) for x in graph.get_edges(v)] != arrays.get_edges(v):
                return "edges of node %d" % v
            if arrays.node_size[v] > 0 and not np.array_equal(np.asarray(graph.get_numeric_node_sequence(v)),
                                                            arrays.get_numeric_node_sequence(v)):
                return "sequence of node %d" % v
            if arrays.exists[v] and bool(graph.is_linear_ref_node_or_linear_ref_dummy_node(v)) != bool(arrays.is_ref[v]):
                return "linear-ref flag of node %d" % v
        return None

    @classmethod
    def _from_obgraph_accessors(cls, graph):
        """Through the accessor methods the reference itself uses, node by node."""
        n_nodes, node_size = cls._obgraph_head(graph)
        rev = graph.get_reverse_edges_hashtable()
        adjacency, seqs = {}, []
        exists = np.zeros(n_nodes, dtype=np.uint8)
        is_ref = np.zeros(n_nodes, dtype=np.uint8)
        for n in range(n_nodes):
            e = [int(x) for x in graph.get_edges(n)]
            if e:
                adjacency[n] = e
            if node_size[n] > 0:
                seqs.append(np.asarray(graph.get_numeric_node_sequence(n), dtype=np.uint8))
        for n in range(n_nodes):
            try:
                has_pred = len(rev[n]) > 0
            except (KeyError, IndexError):
                has_pred = False
            if node_size[n] > 0 or n in adjacency or has_pred:
                exists[n] = 1
                is_ref[n] = 1 if graph.is_linear_ref_node_or_linear_ref_dummy_node(n) else 0
        seq = np.concatenate(seqs) if seqs else np.zeros(0, dtype=np.uint8)
        edge_start, flat_edges = _csr_from_lists(n_nodes, adjacency)
        return cls._obgraph_tail(graph, n_nodes, node_size, seq, edge_start, flat_edges, is_ref, exists)

    # ------------------------------------------------------------------ .npz round trip
    _FILE_KEYS = ("node_size", "seq", "edge_start", "edges", "is_ref", "allele_freq", "exists")

    def to_file(self, file_name):
        np.savez(file_name, first_node=self.first_node, chromosome_start_nodes=np.array(self._chromosome_start_nodes),
                 node_to_ref_offset=(np.zeros(0) if self.node_to_ref_offset is None else np.asarray(self.node_to_ref_offset)),
                 **{k: getattr(self, k) for k in self._FILE_KEYS})

    @classmethod
    def from_file(cls, file_name):
        try:
            d = np.load(file_name)
        except FileNotFoundError:
            d = np.load(file_name + ".npz")
        ntro = d["node_to_ref_offset"]
        return cls(d["node_size"], d["seq"], d["edge_start"], d["edges"], d["is_ref"], d["allele_freq"], d["exists"],
                   int(d["first_node"]), d["chromosome_start_nodes"].tolist(), ntro if len(ntro) else None)

    _DIR_KEYS = ("node_size", "seq", "edge_start", "edges", "rev_start", "rev_edges", "is_ref", "allele_freq", "exists")

    def to_dir(self, path):
        """One .npy per array under `path` (e.g. a /dev/shm directory): the other processes of a node map the graph with
        `from_dir` instead of generating or unpickling a copy each -- what the reference does with
        shared_memory_wrapper.object_to_shared_memory before its process pool (command_line_interface.py:585)."""
        import json
        import os
        os.makedirs(path, exist_ok=True)
        for k in self._DIR_KEYS:
            np.save(os.path.join(path, k + ".npy"), getattr(self, k))
        meta = {"first_node": self.first_node, "chromosome_start_nodes": self._chromosome_start_nodes,
                "has_node_to_ref_offset": self.node_to_ref_offset is not None}
        if self.node_to_ref_offset is not None:
            np.save(os.path.join(path, "node_to_ref_offset.npy"), np.asarray(self.node_to_ref_offset))
        with open(os.path.join(path, "meta.json.tmp"), "w") as fh:
            json.dump(meta, fh)
        os.replace(os.path.join(path, "meta.json.tmp"), os.path.join(path, "meta.json"))    # written last: "complete"

    @classmethod
    def from_dir(cls, path, mmap=True):
        import json
        import os
        with open(os.path.join(path, "meta.json")) as fh:
            meta = json.load(fh)
        d = {k: np.load(os.path.join(path, k + ".npy"), mmap_mode="r" if mmap else None) for k in cls._DIR_KEYS}
        ntro = np.load(os.path.join(path, "node_to_ref_offset.npy"), mmap_mode="r" if mmap else None) \
            if meta["has_node_to_ref_offset"] else None
        return cls(d["node_size"], d["seq"], d["edge_start"], d["edges"], d["is_ref"], d["allele_freq"], d["exists"],
                   meta["first_node"], meta["chromosome_start_nodes"], ntro, d["rev_start"], d["rev_edges"])

    # ------------------------------------------- obgraph-compatible accessor surface
    @property
    def nodes(self):
        return self.node_size

    @property
    def chromosome_start_nodes(self):
        return {i + 1: n for i, n in enumerate(self._chromosome_start_nodes)}

    def linear_ref_nodes(self):
        return set(np.nonzero((self.is_ref != 0) & (self.node_size > 0))[0].tolist())

    def get_first_node(self):
        return self.first_node

    def get_node_size(self, node):
        return int(self.node_size[node])

    def get_numeric_base_sequence(self, node, offset):
        return int(self.seq[self.seq_start[node] + offset])

    def get_numeric_node_sequence(self, node):
        return self.seq[self.seq_start[node]:self.seq_start[node + 1]]

    def get_edges(self, node):
        return self.edges[self.edge_start[node]:self.edge_start[node + 1]].tolist()

    def is_linear_ref_node_or_linear_ref_dummy_node(self, node):
        return bool(self.is_ref[node])

    def get_node_allele_frequencies(self, nodes):
        return self.allele_freq[np.asarray(nodes, dtype=np.int64)]

    def get_node_allele_frequency(self, node):
        return float(self.allele_freq[node])

    def get_reverse_edges_hashtable(self):
        return _ReverseEdges(self)

    def max_node_id(self):
        return self.n_nodes - 1

    def make_linear_ref_node_and_ref_dummy_node_index(self):
        pass

    def position_id_base(self):
        """Per-node base of the default position id (exclusive cumulative node size)."""
        return self.seq_start[:-1]


class _ReverseEdges:
    def __init__(self, g):
        self._g = g

    def __getitem__(self, node):
        g = self._g
        return g.rev_edges[g.rev_start[node]:g.rev_start[node + 1]].tolist()


# ----------------------------------------------------------------------- synthetic graphs
def _random_codes(n, seed_seq):
    """n uniform codes in {0..3}; 32 codes per raw 64-bit draw."""
    rng = np.random.Generator(np.random.PCG64(seed_seq))
    raw = rng.bit_generator.random_raw((n + 31) // 32)
    out = np.empty(len(raw) * 32, dtype=np.uint8)
    b = raw.view(np.uint8).reshape(-1, 8)
    o = out.reshape(-1, 4, 8)
    for j in range(4):
        np.bitwise_and(np.right_shift(b, 2 * j), 3, out=o[:, j, :])
    return out[:n]


def random_codes(n, seed, chunk=1 << 26):
    """Chunk-seeded so that a prefix of a larger draw equals a smaller draw (bench sample)."""
    out = np.empty(n, dtype=np.uint8)
    for c, lo in enumerate(range(0, n, chunk)):
        hi = min(n, lo + chunk)
        out[lo:hi] = _random_codes(hi - lo, [seed, c])
    return out


def synthetic_linear_graph(n_bases, node_len=25000, seed=1234):
    """BASELINE config 2: single-edge chain of linear-ref nodes."""
    assert node_len <= 32767
    n_nodes = (n_bases + node_len - 1) // node_len
    node_size = np.full(n_nodes, node_len, dtype=np.int32)
    node_size[-1] = n_bases - node_len * (n_nodes - 1)
    seq = random_codes(n_bases, seed)
    edge_start = np.minimum(np.arange(n_nodes + 1, dtype=np.int64), n_nodes - 1)
    edges = np.arange(1, n_nodes, dtype=np.int32)
    return GraphArrays(node_size, seq, edge_start, edges, np.ones(n_nodes, np.uint8),
                       first_node=0, chromosome_start_nodes=[0],
                       node_to_ref_offset=np.concatenate([[0], np.cumsum(node_size)])[:n_nodes + 1])


def synthetic_snp_sites(n_ref_bases, n_sites, k, seed, min_gap=2, max_sites_per_window=5):
    """Sorted SNP positions in [k, n_ref_bases-k): pairwise gap >= min_gap and at most
    `max_sites_per_window` sites in any k-bp window."""
    rng = np.random.default_rng([seed, 7])
    want = n_sites
    pos = np.unique(rng.integers(k, n_ref_bases - k, size=int(want * 1.02) + 16, dtype=np.int64))
    keep = np.ones(len(pos), dtype=bool)
    keep[1:] &= np.diff(pos) >= min_gap
    pos = pos[keep]
    m = max_sites_per_window
    if len(pos) > m:
        bad = np.zeros(len(pos), dtype=bool)
        bad[m:] = (pos[m:] - pos[:-m]) < k
        pos = pos[~bad]
    if len(pos) > want:
        sel = np.sort(rng.choice(len(pos), size=want, replace=False))
        pos = pos[sel]
        # thinning keeps both constraints valid
    return pos


def synthetic_snp_graph(n_ref_bases, n_sites, k=31, seed=1234, max_node_len=32767):
    """BASELINE config 3/4: linear reference + SNP bubbles.

    Node ids are topological: ref segment (split into chain nodes of > k bases when longer
    than `max_node_len`), ref-allele node, alt-allele node, next ref segment ...  Successor
    order is [ref_allele, alt_allele].  Ref nodes have allele frequency 1.0; the two alleles
    of a site get f and 1-f with f ~ U(0.01, 0.99)."""
    sites = synthetic_snp_sites(n_ref_bases, n_sites, k, seed)
    S = len(sites)
    # concatenated node sequence = reference with the alt base inserted right after each site
    seq = random_codes(n_ref_bases + S, seed)
    slot_ref = sites + np.arange(S, dtype=np.int64)          # index of the ref-allele base
    rng = np.random.default_rng([seed, 11])
    seq[slot_ref + 1] = (seq[slot_ref] + 1 + rng.integers(0, 3, size=S, dtype=np.uint8)) % 4
    # ref segments: [0,s0) (s0,s1) ... (s_last, G)
    seg_lo = np.concatenate([[0], sites + 1])
    seg_hi = np.concatenate([sites, [n_ref_bases]])
    seg_len = seg_hi - seg_lo
    assert np.all(seg_len >= 1)
    # split long segments into chain chunks, every chunk > k
    n_chunks = np.maximum(1, -(-seg_len // max_node_len))
    if np.any(n_chunks > 1):
        assert np.all(seg_len[n_chunks > 1] // n_chunks[n_chunks > 1] > k + 1)
    total_seg_nodes = int(n_chunks.sum())
    n_nodes = total_seg_nodes + 2 * S
    node_size = np.ones(n_nodes, dtype=np.int32)
    is_ref = np.ones(n_nodes, dtype=np.uint8)
    af = np.ones(n_nodes, dtype=np.float64)
    # node id of first chunk of each segment
    seg_first = np.zeros(S + 1, dtype=np.int64)
    seg_first[1:] = np.cumsum(n_chunks[:-1] + 2)
    # chunk sizes
    chunk_seg = np.repeat(np.arange(S + 1), n_chunks)
    chunk_idx = np.arange(total_seg_nodes) - np.repeat(np.cumsum(n_chunks) - n_chunks, n_chunks)
    base = seg_len[chunk_seg] // n_chunks[chunk_seg]
    rem = seg_len[chunk_seg] - base * n_chunks[chunk_seg]
    csize = base + (chunk_idx < rem)
    chunk_node = seg_first[chunk_seg] + chunk_idx
    node_size[chunk_node] = csize
    ref_allele = seg_first[:-1] + n_chunks[:-1]
    alt_allele = ref_allele + 1
    is_ref[alt_allele] = 0
    f = rng.uniform(0.01, 0.99, size=S)
    af[ref_allele] = f
    af[alt_allele] = 1.0 - f
    # edges: chunk -> next chunk (same segment) ; last chunk -> [ref, alt] ; alleles -> next seg first
    out_deg = np.ones(n_nodes, dtype=np.int64)
    last_chunk = seg_first + n_chunks - 1
    out_deg[last_chunk[:-1]] = 2
    out_deg[last_chunk[-1]] = 0
    edge_start = np.zeros(n_nodes + 1, dtype=np.int64)
    np.cumsum(out_deg, out=edge_start[1:])
    edges = np.zeros(int(edge_start[-1]), dtype=np.int32)
    inner = np.ones(total_seg_nodes, dtype=bool)
    inner[np.cumsum(n_chunks) - 1] = False
    edges[edge_start[chunk_node[inner]]] = chunk_node[inner] + 1
    edges[edge_start[last_chunk[:-1]]] = ref_allele
    edges[edge_start[last_chunk[:-1]] + 1] = alt_allele
    edges[edge_start[ref_allele]] = seg_first[1:]
    edges[edge_start[alt_allele]] = seg_first[1:]
    ntro = np.zeros(n_nodes + 1, dtype=np.int64)
    lin = np.nonzero(is_ref)[0]
    ntro[lin] = np.concatenate([[0], np.cumsum(node_size[lin])[:-1]])
    return GraphArrays(node_size, seq, edge_start, edges, is_ref, af,
                       first_node=0, chromosome_start_nodes=[0], node_to_ref_offset=ntro)


def synthetic_indel_graph(n_ref_bases, n_sites, k=31, seed=1234, p_del=0.1, p_ins=0.1, max_node_len=32767):
    """The SNP/indel variant of BASELINE config 3 (north star: "~5 M SNP/indel bubbles"): like
    `synthetic_snp_graph`, but a site is a 1-bp deletion with probability `p_del` (ref allele = the reference base, alt
    allele = an EMPTY node) or a 1-bp insertion with probability `p_ins` (ref allele = an empty linear-ref dummy node,
    alt allele = the inserted base; the reference base at the site opens the next segment).  Node ids are topological:
    segment chunks, ref allele, alt allele, next segment ...; successor order [ref_allele, alt_allele]."""
    sites = synthetic_snp_sites(n_ref_bases, n_sites, k, seed)
    S = len(sites)
    rng = np.random.default_rng([seed, 13])
    kind = rng.random(S)
    is_del = kind < p_del
    is_ins = (kind >= p_del) & (kind < p_del + p_ins)
    ref_len = np.where(is_ins, 0, 1).astype(np.int64)          # reference bases the ref allele holds
    alt_len = np.where(is_del, 0, 1).astype(np.int64)
    reference = random_codes(n_ref_bases, seed)
    # ref segments between the sites: [0, s0), [s0 + ref_len0, s1), ..., [s_last + ref_len_last, G)
    seg_lo = np.concatenate([[0], sites + ref_len])
    seg_hi = np.concatenate([sites, [n_ref_bases]])
    seg_len = seg_hi - seg_lo
    assert np.all(seg_len >= 1)
    n_chunks = np.maximum(1, -(-seg_len // max_node_len))
    if np.any(n_chunks > 1):
        assert np.all(seg_len[n_chunks > 1] // n_chunks[n_chunks > 1] > k + 1)
    total_seg_nodes = int(n_chunks.sum())
    n_nodes = total_seg_nodes + 2 * S
    node_size = np.zeros(n_nodes, dtype=np.int32)
    is_ref = np.ones(n_nodes, dtype=np.uint8)
    af = np.ones(n_nodes, dtype=np.float64)
    seg_first = np.zeros(S + 1, dtype=np.int64)
    seg_first[1:] = np.cumsum(n_chunks[:-1] + 2)
    chunk_seg = np.repeat(np.arange(S + 1), n_chunks)
    chunk_idx = np.arange(total_seg_nodes) - np.repeat(np.cumsum(n_chunks) - n_chunks, n_chunks)
    base = seg_len[chunk_seg] // n_chunks[chunk_seg]
    rem = seg_len[chunk_seg] - base * n_chunks[chunk_seg]
    chunk_node = seg_first[chunk_seg] + chunk_idx
    node_size[chunk_node] = base + (chunk_idx < rem)
    ref_allele = seg_first[:-1] + n_chunks[:-1]
    alt_allele = ref_allele + 1
    node_size[ref_allele] = ref_len
    node_size[alt_allele] = alt_len
    is_ref[alt_allele] = 0                                       # the empty ref allele of an insertion stays a ref dummy
    f = rng.uniform(0.01, 0.99, size=S)
    af[ref_allele] = f
    af[alt_allele] = 1.0 - f
    # sequence in node order: every base is a reference base except the alt alleles' (SNP: a different base, insertion:
    # any base)
    seq_start = np.zeros(n_nodes + 1, dtype=np.int64)
    np.cumsum(node_size, out=seq_start[1:])
    seq = np.empty(int(seq_start[-1]), dtype=np.uint8)
    alt_slot = seq_start[alt_allele[alt_len == 1]]
    is_alt_base = np.zeros(len(seq), dtype=bool)
    is_alt_base[alt_slot] = True
    seq[~is_alt_base] = reference
    ref_at_site = reference[sites[alt_len == 1]]
    shifted = (ref_at_site + 1 + rng.integers(0, 3, size=len(ref_at_site), dtype=np.uint8)) % 4
    anyb = rng.integers(0, 4, size=len(ref_at_site), dtype=np.uint8)
    seq[alt_slot] = np.where(is_ins[alt_len == 1], anyb, shifted)
    # edges as in synthetic_snp_graph
    out_deg = np.ones(n_nodes, dtype=np.int64)
    last_chunk = seg_first + n_chunks - 1
    out_deg[last_chunk[:-1]] = 2
    out_deg[last_chunk[-1]] = 0
    edge_start = np.zeros(n_nodes + 1, dtype=np.int64)
    np.cumsum(out_deg, out=edge_start[1:])
    edges = np.zeros(int(edge_start[-1]), dtype=np.int32)
    inner = np.ones(total_seg_nodes, dtype=bool)
    inner[np.cumsum(n_chunks) - 1] = False
    edges[edge_start[chunk_node[inner]]] = chunk_node[inner] + 1
    edges[edge_start[last_chunk[:-1]]] = ref_allele
    edges[edge_start[last_chunk[:-1]] + 1] = alt_allele
    edges[edge_start[ref_allele]] = seg_first[1:]
    edges[edge_start[alt_allele]] = seg_first[1:]
    ntro = np.zeros(n_nodes + 1, dtype=np.int64)
    lin = np.nonzero(is_ref)[0]
    ntro[lin] = np.concatenate([[0], np.cumsum(node_size[lin])[:-1]])
    return GraphArrays(node_size, seq, edge_start, edges, is_ref, af,
                       first_node=0, chromosome_start_nodes=[0], node_to_ref_offset=ntro)


def synthetic_nested_graph(n_ref_bases, n_sites, k=31, seed=1234, p_nest=0.2, max_node_len=32767, max_sites_per_window=2):
    """BASELINE config 3 with variants INSIDE alternative alleles: like `synthetic_snp_graph`, but a fraction `p_nest`
    of the sites are an insertion-like allele that itself contains a SNP: the alternative allele is the chain
    Z1 (1-4 bases) -> {za | zb} (one base each) -> Z2 (1-4 bases), all four nodes non-linear-ref, so za, zb and Z2 have
    no linear-ref predecessor and Z1, za, zb do not have exactly one linear-ref successor (what
    `gki_classify_nodes` calls NESTED / CHECK).  Node ids are topological: segment chunks, ref allele, then the alt
    allele's node(s), next segment ...; successor order [ref allele, alt entry].  At most `max_sites_per_window`
    sites per k bases, so that a window holds at most 3 * max_sites_per_window variant nodes."""
    sites = synthetic_snp_sites(n_ref_bases, n_sites, k, seed, max_sites_per_window=max_sites_per_window)
    S = len(sites)
    rng = np.random.default_rng([seed, 17])
    nested = rng.random(S) < p_nest
    seg_lo = np.concatenate([[0], sites + 1])
    seg_hi = np.concatenate([sites, [n_ref_bases]])
    seg_len = seg_hi - seg_lo
    assert np.all(seg_len >= 1)
    n_chunks = np.maximum(1, -(-seg_len // max_node_len))
    if np.any(n_chunks > 1):
        assert np.all(seg_len[n_chunks > 1] // n_chunks[n_chunks > 1] > k + 1)
    site_nodes = np.where(nested, 5, 2).astype(np.int64)           # ref allele + (alt | Z1, za, zb, Z2)
    total_seg_nodes = int(n_chunks.sum())
    n_nodes = total_seg_nodes + int(site_nodes.sum())
    node_size = np.ones(n_nodes, dtype=np.int32)
    is_ref = np.ones(n_nodes, dtype=np.uint8)
    af = np.ones(n_nodes, dtype=np.float64)
    seg_first = np.zeros(S + 1, dtype=np.int64)
    seg_first[1:] = np.cumsum(n_chunks[:-1] + site_nodes)
    chunk_seg = np.repeat(np.arange(S + 1), n_chunks)
    chunk_idx = np.arange(total_seg_nodes) - np.repeat(np.cumsum(n_chunks) - n_chunks, n_chunks)
    base = seg_len[chunk_seg] // n_chunks[chunk_seg]
    rem = seg_len[chunk_seg] - base * n_chunks[chunk_seg]
    chunk_node = seg_first[chunk_seg] + chunk_idx
    node_size[chunk_node] = base + (chunk_idx < rem)
    ref_allele = seg_first[:-1] + n_chunks[:-1]
    alt_in = ref_allele + 1
    z1 = alt_in[nested]
    za, zb, z2 = z1 + 1, z1 + 2, z1 + 3
    node_size[z1] = rng.integers(1, 5, size=len(z1))
    node_size[z2] = rng.integers(1, 5, size=len(z1))
    for arr in (alt_in, za, zb, z2):
        is_ref[arr] = 0
    f = rng.uniform(0.01, 0.99, size=S)
    af[ref_allele] = f
    af[alt_in] = 1.0 - f
    af[za] = af[zb] = af[z2] = 1.0 - f[nested]
    # sequence in node order, drawn as one stream (the reference is what the linear nodes spell); a SNP's alt base
    # differs from the ref allele's base right before it, zb's base from za's
    seq_start = np.zeros(n_nodes + 1, dtype=np.int64)
    np.cumsum(node_size, out=seq_start[1:])
    seq = random_codes(int(seq_start[-1]), seed)
    snp_alt = alt_in[~nested]
    seq[seq_start[snp_alt]] = (seq[seq_start[snp_alt] - 1] + 1 + rng.integers(0, 3, size=len(snp_alt), dtype=np.uint8)) % 4
    seq[seq_start[zb]] = (seq[seq_start[za]] + 1 + rng.integers(0, 3, size=len(zb), dtype=np.uint8)) % 4
    lin = np.nonzero(is_ref)[0]
    # edges
    out_deg = np.ones(n_nodes, dtype=np.int64)
    last_chunk = seg_first + n_chunks - 1
    out_deg[last_chunk[:-1]] = 2
    out_deg[last_chunk[-1]] = 0
    out_deg[z1] = 2
    edge_start = np.zeros(n_nodes + 1, dtype=np.int64)
    np.cumsum(out_deg, out=edge_start[1:])
    edges = np.zeros(int(edge_start[-1]), dtype=np.int32)
    inner = np.ones(total_seg_nodes, dtype=bool)
    inner[np.cumsum(n_chunks) - 1] = False
    edges[edge_start[chunk_node[inner]]] = chunk_node[inner] + 1
    edges[edge_start[last_chunk[:-1]]] = ref_allele
    edges[edge_start[last_chunk[:-1]] + 1] = alt_in
    edges[edge_start[ref_allele]] = seg_first[1:]
    edges[edge_start[snp_alt]] = seg_first[1:][~nested]
    edges[edge_start[z1]] = za
    edges[edge_start[z1] + 1] = zb
    edges[edge_start[za]] = z2
    edges[edge_start[zb]] = z2
    edges[edge_start[z2]] = seg_first[1:][nested]
    ntro = np.zeros(n_nodes + 1, dtype=np.int64)
    ntro[lin] = np.concatenate([[0], np.cumsum(node_size[lin])[:-1]])
    return GraphArrays(node_size, seq, edge_start, edges, is_ref, af,
                       first_node=0, chromosome_start_nodes=[0], node_to_ref_offset=ntro)


def synthetic_haplotype_sequence(graph, seed=99):
    """Base codes along one random path of a `synthetic_snp_graph`: at every SNP bubble the ref or the alt allele
    with probability 1/2 (read simulation for the lookup benchmarks)."""
    alt_nodes = np.flatnonzero(graph.is_ref == 0)
    alt_slot = graph.seq_start[alt_nodes]                    # the ref allele's base sits right before it
    rng = np.random.default_rng([seed, 5])
    drop = alt_slot - rng.integers(0, 2, size=len(alt_slot))
    keep = np.ones(len(graph.seq), dtype=bool)
    keep[drop] = False
    return graph.seq[keep]
