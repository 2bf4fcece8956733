"""Order-free statement of DenseKmerFinder.find() for ANY DAG (test utility; the rule the GENERAL kernels implement).

The reference's DFS (kmer_finder.py:254-417) reaches a k-window P ending at position e iff some path from a search
start ends with P and every step X -> Y of that path into a node Y whose entry is not free satisfied the variant
limit when it was taken: `#distinct non-linear-ref nodes in W(X) < max_variant_nodes` (:391-403), W(X) = the last k
real bases ending at X's end (interior dummy entries included, leading ones not).  Entry into Y is free when Y is a
linear-ref(-dummy) node, or Y is in `only_follow_nodes` (:386-388: if any successor of X is forced, ONLY forced
successors are followed -- the other out-edges of X do not exist for the search -- and the limit is bypassed).

The `_positions_treated` prune (:311-319) only removes re-arrivals with an identical window, whose future is
identical, so the output is: every reachable (e, P) exactly once.

Backward form.  Per node, given the alive edges:
  T       linear-ref node reachable by a history that is all linear-ref for its last k bases (or back to a search
          root): that history contributes no variant node to any later window and is always allowed -- it dominates
          every other history, so a backward enumeration may stop there.
  SIMPLE  not T, but one predecessor is T (stop after this node, choosing that predecessor).
  NESTED  not T, has predecessors, none of them T: histories have to be enumerated through its predecessors.
A window whose first node is T or SIMPLE is decided by its own nodes; one whose first node is NESTED needs
`history_ok`: some backward path through NESTED nodes down to a T / SIMPLE node on which every open constraint holds.

The reference asserts that a node whose window is at the limit has exactly one linear-ref successor (:402); a
reachable state violating that is an AssertionError there and `SpecError` here.
"""
from collections import Counter

REF, FORCED, T, SIMPLE, NESTED, CHECK, HFS, DEAD = 1, 2, 4, 8, 16, 32, 64, 128


class SpecError(Exception):
    pass


def topological_order(g):
    indeg = [int(g.rev_start[n + 1] - g.rev_start[n]) for n in range(g.n_nodes)]
    import heapq
    heap = [n for n in range(g.n_nodes) if indeg[n] == 0]
    heapq.heapify(heap)
    order = []
    while heap:
        n = heapq.heappop(heap)
        order.append(n)
        for m in g.edges[g.edge_start[n]:g.edge_start[n + 1]].tolist():
            indeg[m] -= 1
            if indeg[m] == 0:
                heapq.heappush(heap, m)
    return order


def history_ok(g, flags, k, M, path):
    """path: window nodes from the end node back to its first node q = path[-1], whole nodes.  Is there a history
    before q's entry -- a backward path over alive edges through reachable nodes -- on which every step into a
    non-free node keeps to the limit?  The enumeration ends at a node p when
      * p is T or SIMPLE (the all-linear-ref history behind it adds nothing), or
      * every open constraint closes inside p: then all that is left is "p is entered at all", which the DEAD flag of
        the already classified nodes says (classification runs in topological order)."""
    size = g.node_size
    nonref = lambda n: 0 if flags[n] & REF else 1
    nonfree = lambda n: not (flags[n] & (REF | FORCED))

    def preds(n):
        out = []
        for p in g.rev_edges[g.rev_start[n]:g.rev_start[n + 1]].tolist():
            if flags[p] & DEAD:
                continue
            if (flags[p] & HFS) and not (flags[n] & FORCED):
                continue
            out.append(p)
        return out

    q = path[-1]
    cons = []                                        # (reach t beyond q's entry, budget m)
    for i, y in enumerate(path):
        if not nonfree(y):
            continue
        between = sum(int(size[x]) for x in path[i + 1:])        # bases between y's entry and q's entry
        c = sum(nonref(x) for x in path[i + 1:])
        if M - c <= 0:
            return False
        cons.append((k - between, M - c))

    def ok(hist, dist):
        for t, m in cons:
            if sum(nonref(x) for x, dx in zip(hist, dist) if dx < t) >= m:
                return False
        for l, y in enumerate(hist):
            if not nonfree(y):
                continue
            if M < 1:
                return False
            if sum(nonref(x) for x, dx in zip(hist[l + 1:], dist[l + 1:]) if dx - dist[l + 1] < k) >= M:
                return False
        return True

    def all_closed(hist, dist):
        end = dist[-1] + int(size[hist[-1]])         # bases of the history up to and including its last node
        if any(t > end for t, m in cons):
            return False
        for l, y in enumerate(hist[:-1]):
            if nonfree(y) and end - dist[l + 1] < k:
                return False
        return True

    def rec(node, hist, dist):
        for p in preds(node):
            h2 = hist + [p]
            d2 = dist + [dist[-1] + int(size[hist[-1]]) if hist else 0]
            if not ok(h2, d2):
                continue
            if flags[p] & (T | SIMPLE):
                return True
            if all_closed(h2, d2):
                return True
            if rec(p, h2, d2):
                return True
        return False

    return rec(q, [], [])


def classify(g, k, M, follow=None, critical_nodes=()):
    """Per-node flag byte, in topological order.  DEAD is exact: the node is never entered by the search.  Every
    critical node starts a search of its own with no history (kmer_finder.py:190-232), like a chromosome start."""
    N = g.n_nodes
    F = set(int(x) for x in follow) if follow is not None else set()
    size = g.node_size
    succ = lambda n: g.edges[g.edge_start[n]:g.edge_start[n + 1]].tolist()
    pred = lambda n: g.rev_edges[g.rev_start[n]:g.rev_start[n + 1]].tolist()
    flags = [0] * N
    roots = set(int(x) for x in g.chromosome_start_nodes.values()) | {g.first_node} | set(int(x) for x in critical_nodes)
    for n in range(N):
        if g.is_ref[n]:
            flags[n] |= REF
        if n in F:
            flags[n] |= FORCED
        s = succ(n)
        hfs = any(m in F for m in s)
        if hfs:
            flags[n] |= HFS
        if s and not hfs and sum(1 for m in s if g.is_ref[m]) != 1:
            flags[n] |= CHECK
    INF = 1 << 40
    clean = [0] * N             # all-linear-ref bases of the best history before the node's entry
    for n in topological_order(g):
        if n in roots:
            clean[n] = INF
            flags[n] |= T
            continue
        ps = [p for p in pred(n) if not (flags[p] & DEAD) and (not (flags[p] & HFS) or (flags[n] & FORCED))]
        if not g.exists[n] or not ps:
            flags[n] |= DEAD
            continue
        any_t = any(flags[p] & T for p in ps)
        if g.is_ref[n]:
            best = max([min(INF, clean[p] + int(size[p])) for p in ps if g.is_ref[p]] + [0])
            clean[n] = best
            if best >= k:
                flags[n] |= T
        if not (flags[n] & T):
            flags[n] |= SIMPLE if any_t else NESTED
        if not (flags[n] & (REF | FORCED)):          # not free to enter: is there an admissible history at all?
            if M < 1 or (not any_t and not history_ok(g, flags, k, M, [n])):
                flags[n] = (flags[n] & ~(SIMPLE | NESTED)) | DEAD
    return flags


def spec_rows_general(g, k, max_variant_nodes=4, one_node=False, critical=None, follow=None):
    """Counter of (hash, start_node, start_offset, node, allele_freq); raises SpecError where the reference asserts."""
    M = max_variant_nodes
    crit = critical or {}
    flags = classify(g, k, M, follow, crit.keys())
    size = g.node_size
    rows = Counter()

    def preds(n):
        """alive predecessors of n that are not dead"""
        out = []
        for p in g.rev_edges[g.rev_start[n]:g.rev_start[n + 1]].tolist():
            if flags[p] & DEAD:
                continue
            if (flags[p] & HFS) and not (flags[n] & FORCED):
                continue
            out.append(p)
        return out

    nonref = lambda n: 0 if flags[n] & REF else 1
    nonfree = lambda n: not (flags[n] & (REF | FORCED))

    def reachable(n):
        return not (flags[n] & DEAD)

    def admissible(path):
        """path: window nodes from the end node back to the first node (incl. interior dummies)."""
        if not p_internal_ok(path):
            return False
        f = flags[path[-1]]
        if f & (T | SIMPLE):
            return True
        return bool(f & NESTED) and history_ok(g, flags, k, M, path)

    def emit(hash_, path, n_end, o_end):
        uniq = sorted(set(path))
        af = min(float(g.allele_freq[x]) for x in uniq)
        for x in (uniq[:1] if one_node else uniq):
            rows[(hash_, n_end, o_end, x, af)] += 1

    def take(node, upto, bases_rev, path, need, on_complete, on_partial):
        """Backward enumeration: take bases node[upto-1], node[upto-2], ... until `need` are collected, continuing
        through every alive predecessor.  on_complete(path, bases_rev) per window; on_partial(path) where the graph
        ends first (graph start)."""
        c = crit.get(node)
        off = upto - 1
        while need > 0 and off >= 0:
            bases_rev = bases_rev + [int(g.seq[g.seq_start[node] + off])]
            need -= 1
            if need > 0 and c is not None and 0 < c < k - 1 and off == c:
                # E1: a window never spans (N, c-1), (N, c) -- the search restarted at (N, c) with no history, so what it
                # holds at the end position is this shorter window (it still decides the assertion of :402)
                if on_partial is not None:
                    on_partial(path)
                return
            off -= 1
        if need == 0:
            on_complete(path, bases_rev)
            return
        ps = preds(node)
        if not ps and on_partial is not None:
            on_partial(path)
        for p in ps:
            take(p, int(size[p]), bases_rev, path + [p], need, on_complete, on_partial)

    def p_internal_ok(path):
        v = 0
        for i in range(len(path) - 1, -1, -1):
            y = path[i]
            if nonfree(y) and i < len(path) - 1 and v >= M:
                return False
            v += nonref(y)
        return True

    for n in range(g.n_nodes):
        if not g.exists[n] or flags[n] & DEAD:
            continue
        sz = int(size[n])
        for o in range(sz):
            def complete(path, bases_rev, n=n, o=o):
                if not admissible(path):
                    return
                h = 0
                for i, b in enumerate(reversed(bases_rev)):
                    h += b << (2 * i)
                emit(h, path, n, o)
            take(n, o + 1, [], [n], k, complete, None)
        if flags[n] & CHECK:                 # the assertion of :402 at the node's end
            def check(path, n=n):
                if sum(nonref(x) for x in set(path)) >= M:
                    raise SpecError("node %d: window at the variant limit and not exactly one linear-ref successor" % n)

            def complete_c(path, bases_rev):
                if admissible(path):
                    check(path)

            def partial_c(path):
                # fewer than k bases exist before this position (graph start): the history is what there is
                if p_internal_ok(path) and (len(path) > 1 or reachable(path[0])):
                    check(path)

            take(n, sz, [], [n], k, complete_c, partial_c)    # a dummy node (sz 0): the k bases before it + itself
    return rows
