"""Random variation-graph generators for parity tests (test utility)."""
import numpy as np

_L = "ACGT"


def _rand_seq(rng, n):
    return "".join(_L[i] for i in rng.integers(0, 4, size=n))


def random_bubble_graph(rng, n_var=None, min_ref=1, max_ref=14, p_indel=0.5,
                        first_ref=None, last_ref=None, shuffle_succ=True, with_af=False,
                        chain_after=None):
    """Chain of bubbles.  Each bubble is a SNP (two 1-bp alleles), a deletion (ref allele of
    1-3 bp on the linear ref, empty alt allele) or an insertion (empty ref allele = linear-ref
    dummy, alt allele 1-3 bp).  Node ids increase along the graph.

    chain_after: optional dict {bubble_index: chain_len} -> the ref segment after that bubble is
    split into a first node of `chain_len` bases followed by the rest (single-edge chain).
    Returns (node_sequences, edges, linear_ref_nodes, allele_frequencies)."""
    if n_var is None:
        n_var = int(rng.integers(2, 8))
    seqs, edges, linear, af = {}, {}, [], {}
    nid = 0

    def add(seq, is_lin, freq=1.0):
        nonlocal nid
        seqs[nid] = seq
        af[nid] = freq
        if is_lin:
            linear.append(nid)
        nid += 1
        return nid - 1

    def ref_segment(length, chain_len=None):
        s = _rand_seq(rng, length)
        if chain_len is not None and 0 < chain_len < length:
            a = add(s[:chain_len], True)
            b = add(s[chain_len:], True)
            edges[a] = [b]
            return a, b
        a = add(s, True)
        return a, a

    n0 = first_ref if first_ref is not None else int(rng.integers(min_ref, max_ref + 1))
    head, tail = ref_segment(n0, (chain_after or {}).get(-1))
    for b in range(n_var):
        kind = rng.random()
        f = float(rng.uniform(0.01, 0.99)) if with_af else 1.0
        if kind >= p_indel:                       # SNP
            r = _rand_seq(rng, 1)
            a = _L[(_L.index(r) + 1 + int(rng.integers(0, 3))) % 4]
            ref_a = add(r, True, f)
            alt_a = add(a, False, 1.0 - f if with_af else 1.0)
        elif kind < p_indel / 2:                  # deletion: alt allele empty
            ref_a = add(_rand_seq(rng, int(rng.integers(1, 4))), True, f)
            alt_a = add("", False, 1.0 - f if with_af else 1.0)
        else:                                     # insertion: ref allele empty (ref dummy)
            ref_a = add("", False, f)             # NOT listed in linear_ref_nodes
            alt_a = add(_rand_seq(rng, int(rng.integers(1, 4))), False, 1.0 - f if with_af else 1.0)
        succ = [ref_a, alt_a]
        if shuffle_succ and rng.random() < 0.5:
            succ = succ[::-1]
        edges[tail] = succ
        is_last = b == n_var - 1
        n = (last_ref if (is_last and last_ref is not None)
             else int(rng.integers(min_ref, max_ref + 1)))
        head, new_tail = ref_segment(n, (chain_after or {}).get(b))
        edges[ref_a] = [head]
        edges[alt_a] = [head]
        tail = new_tail
    return seqs, edges, linear, (af if with_af else None)


def overlapping_bubble_graph(rng, n_var=4, min_ref=2, max_ref=8):
    """Bubbles plus a long alternative allele that skips over a whole bubble
    (shape of tests/test_kmer_finder.py:51-62 `test_nested_paths`)."""
    seqs, edges, linear, _ = random_bubble_graph(rng, n_var, min_ref, max_ref, p_indel=0.3)
    # find ref segments (linear nodes with 2 successors) and add a skip allele across one bubble
    branch = [n for n in linear if len(edges.get(n, [])) == 2]
    if len(branch) >= 2:
        i = int(rng.integers(0, len(branch) - 1))
        src = branch[i]
        # join node two bubbles ahead = successor of an allele of branch[i+1]
        dst = edges[edges[branch[i + 1]][0]][0]
        new = max(seqs) + 1
        seqs[new] = _rand_seq(rng, int(rng.integers(1, 5)))
        edges[src] = edges[src] + [new]
        edges[new] = [dst]
    return seqs, edges, linear, None


def nested_bubble_graph(rng, n_var=4, min_ref=2, max_ref=10, p_nest=0.6, p_chain=0.0):
    """Bubbles whose alt allele may itself contain a bubble (a variant inside an insertion):
    R -> {ref allele | Z1 -> {za | zb} -> Z2} -> R'.  Nodes inside the alt allele have no linear-ref predecessor.
    p_chain: probability that a ref segment is split into a single-edge chain of two nodes (critical points with a
    lossy restart right behind a nested bubble)."""
    seqs, edges, linear = {}, {}, []
    nid = 0

    def add(seq, is_lin):
        nonlocal nid
        seqs[nid] = seq
        if is_lin:
            linear.append(nid)
        nid += 1
        return nid - 1

    tail = add(_rand_seq(rng, int(rng.integers(min_ref, max_ref + 1))), True)
    for _ in range(n_var):
        ref_a = add(_rand_seq(rng, int(rng.integers(1, 4))), True)
        if rng.random() < p_nest:
            z1 = add(_rand_seq(rng, int(rng.integers(1, 6))), False)
            za = add(_rand_seq(rng, int(rng.integers(0, 3))), False)
            zb = add(_rand_seq(rng, int(rng.integers(1, 3))), False)
            z2 = add(_rand_seq(rng, int(rng.integers(1, 6))), False)
            edges[z1] = [za, zb]
            edges[za] = [z2]
            edges[zb] = [z2]
            alt_in, alt_out = z1, z2
        else:
            alt_in = alt_out = add(_rand_seq(rng, int(rng.integers(1, 3))), False)
        edges[tail] = [ref_a, alt_in]
        head = add(_rand_seq(rng, int(rng.integers(min_ref, max_ref + 1))), True)
        edges[ref_a] = [head]
        edges[alt_out] = [head]
        tail = head
        if p_chain and rng.random() < p_chain:
            tail = add(_rand_seq(rng, int(rng.integers(min_ref, max_ref + 1))), True)
            edges[head] = [tail]
    return seqs, edges, linear, None


def deep_nested_graph(rng, n_var=4, max_depth=2, min_ref=1, max_ref=8, p_nest=0.5, max_allele=3):
    """Bubbles nested to `max_depth`: an alternative allele is a chain  segment -> bubble -> segment ...  whose inner
    bubbles have two non-linear alleles (either may be empty, or complex again).  Gives multi-node alleles, nodes
    with no linear-ref predecessor several levels deep, empty nodes inside alleles and successor lists in random
    order.  Returns (node_sequences, edges, linear_ref_nodes, None)."""
    seqs, edges, linear = {}, {}, []
    nid = 0

    def add(seq, is_lin):
        nonlocal nid
        seqs[nid] = seq
        if is_lin:
            linear.append(nid)
        nid += 1
        return nid - 1

    def link(a, b):
        edges.setdefault(a, []).append(b)

    def allele(depth, allow_empty):
        """a non-linear allele: returns (entry node, exit node)"""
        if depth < max_depth and rng.random() < p_nest:
            first = add(_rand_seq(rng, int(rng.integers(1, max_allele + 2))), False)
            tail = first
            for _ in range(int(rng.integers(1, 3))):
                a_in, a_out = allele(depth + 1, True)
                b_in, b_out = allele(depth + 1, seqs[a_in] != "" or a_in != a_out)
                succ = [a_in, b_in] if rng.random() < 0.5 else [b_in, a_in]
                for s in succ:
                    link(tail, s)
                tail = add(_rand_seq(rng, int(rng.integers(1, max_allele + 2))), False)
                link(a_out, tail)
                link(b_out, tail)
            return first, tail
        lo = 0 if allow_empty else 1
        n = add(_rand_seq(rng, int(rng.integers(lo, max_allele + 1))), False)
        return n, n

    tail = add(_rand_seq(rng, int(rng.integers(min_ref, max_ref + 1))), True)
    for _ in range(n_var):
        ref_len = int(rng.integers(0, max_allele + 1))
        ref_a = add(_rand_seq(rng, ref_len), ref_len > 0)          # empty: a linear-ref dummy, not listed as linear
        alt_in, alt_out = allele(0, ref_len > 0)
        succ = [ref_a, alt_in] if rng.random() < 0.5 else [alt_in, ref_a]
        for s in succ:
            link(tail, s)
        head = add(_rand_seq(rng, int(rng.integers(min_ref, max_ref + 1))), True)
        link(ref_a, head)
        link(alt_out, head)
        tail = head
    return seqs, edges, linear, None


def empty_chain_graph(rng, n_sites, first_ref=8, last_ref=40, max_ins=2, p_plain=0.2, p_snp=0.1, with_af=False):
    """A run of `n_sites` variant sites with NOTHING between them, so that one k-window crosses all of them: each site is
    an insertion (an empty linear-ref dummy beside an inserted allele of 1..max_ins bases), with probability p_plain a
    single empty node, with probability p_snp a SNP (two 1-bp alleles, which use up window bases).  The windows that take
    the empty allele everywhere span n_sites + 2 nodes -- the shape behind GKI_MAX_WINDOW_NODES and the finder's slow path.
    Returns (node_sequences, edges, linear_ref_nodes, allele_frequencies or None)."""
    seqs, edges, linear, af = {}, {}, [], {}

    def add(seq, is_lin, freq=1.0):
        nid = len(seqs)
        seqs[nid] = seq
        af[nid] = freq
        if is_lin:
            linear.append(nid)
        return nid

    tails = [add(_rand_seq(rng, first_ref), True)]
    for _ in range(n_sites):
        u = rng.random()
        f = float(rng.uniform(0.01, 0.99)) if with_af else 1.0
        if u < p_plain:
            alleles = [add("", False, f)]
        elif u < p_plain + p_snp:
            r = _rand_seq(rng, 1)
            alleles = [add(r, True, f), add(_L[(_L.index(r) + 1 + int(rng.integers(0, 3))) % 4], False, 1.0 - f if with_af else 1.0)]
        else:
            alleles = [add("", False, f), add(_rand_seq(rng, int(rng.integers(1, max_ins + 1))), False, 1.0 - f if with_af else 1.0)]
        succ = alleles[::-1] if rng.random() < 0.5 else list(alleles)
        for t in tails:
            edges[t] = list(succ)
        tails = alleles
    end = add(_rand_seq(rng, last_ref), True)
    for t in tails:
        edges[t] = [end]
    return seqs, edges, linear, (af if with_af else None)
