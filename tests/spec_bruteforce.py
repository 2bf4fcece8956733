"""Order-free brute-force statement of DenseKmerFinder's output (test utility).

Follows SURVEY.md section 8(a'): for every end position and every backward window of exactly
k real bases, emit one row per distinct window node when the window is admissible.  On graphs
where every node with predecessors has at least one linear-ref(-dummy) predecessor, the
admissibility predicate of the reference's variant limit (kmer_finder.py:391-403) reduces to
"number of distinct non-linear-ref nodes in the window <= max_variant_nodes".

Exception E1 (SURVEY.md 8a'): windows that contain both (N, c-1) and (N, c) for a critical
point (N, c) with 0 < c < k-1 are never produced by the reference and are dropped here too.
"""
from collections import Counter


def spec_rows(g, k, max_variant_nodes=4, one_node=False, critical=None):
    """g: GraphArrays.  critical: optional dict node -> critical offset.
    Returns Counter of (hash, start_node, start_offset, node, allele_freq)."""
    rows = Counter()
    crit = critical or {}
    size = g.node_size

    def preds(n):
        return g.rev_edges[g.rev_start[n]:g.rev_start[n + 1]].tolist()

    def emit(bases_rev, nodes, n_end, o_end):
        h = 0
        for i, b in enumerate(reversed(bases_rev)):
            h += int(b) << (2 * i)
        uniq = sorted(set(nodes))
        v = sum(1 for x in uniq if not g.is_ref[x])
        if v > max_variant_nodes:
            return
        af = min(float(g.allele_freq[x]) for x in uniq)
        out_nodes = uniq[:1] if one_node else uniq
        for x in out_nodes:
            rows[(h, n_end, o_end, x, af)] += 1

    def walk(node, off, bases_rev, nodes, n_end, o_end):
        # take base (node, off)
        bases_rev = bases_rev + [g.seq[g.seq_start[node] + off]]
        nodes = nodes + [node]
        if len(bases_rev) == k:
            emit(bases_rev, nodes, n_end, o_end)
            return
        if off > 0:
            c = crit.get(node)
            if c is not None and 0 < c < k - 1 and off == c:
                return                      # E1: window would span (N,c-1),(N,c)
            walk(node, off - 1, bases_rev, nodes, n_end, o_end)
            return
        # offset 0: go through predecessors, chaining through empty nodes
        stack = [(p, []) for p in preds(node)]
        while stack:
            p, dummies = stack.pop()
            if size[p] == 0:
                for q in preds(p):
                    stack.append((q, dummies + [p]))
            else:
                walk(p, int(size[p]) - 1, bases_rev, nodes + dummies, n_end, o_end)

    for n in range(g.n_nodes):
        for o in range(int(size[n])):
            walk(n, o, [], [], n, o)
    return rows
