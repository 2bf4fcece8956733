#!/usr/bin/env python3
"""Randomised soak test (not part of the suite): GPU DenseKmerFinder.find() against the oracle on many random graphs,
k in [2, 31], every max_variant_nodes regime, both node modes, random critical-path chunks, optional whitelist.
  python tools/soak_parity.py --seconds 120 --seed 1
Exits non-zero at the first mismatch (prints the seed and parameters to reproduce)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from graph_kmer_index_amd import DenseKmerFinder, CriticalGraphPaths, GraphArrays
from graph_kmer_index_amd.graph import synthetic_indel_graph
from graphgen import random_bubble_graph, overlapping_bubble_graph
from gpu_util import assert_same_records, finder_cols
from oracle import oracle


def make_graph(rng, k):
    mode = rng.choice(["bubble", "overlap", "chain", "indel"])
    if mode == "bubble":
        seqs, edges, lin, af = random_bubble_graph(rng, n_var=int(rng.integers(2, 40)), min_ref=1, max_ref=int(rng.integers(2, 3 * k + 3)),
                                                   p_indel=float(rng.choice([0.0, 0.3, 0.7])), with_af=True)
        return mode, GraphArrays.from_dicts(seqs, edges, lin, af)
    if mode == "overlap":
        seqs, edges, lin, af = overlapping_bubble_graph(rng, n_var=int(rng.integers(3, 14)), min_ref=2, max_ref=max(3, k))
        return mode, GraphArrays.from_dicts(seqs, edges, lin, af)
    if mode == "chain":
        nv = int(rng.integers(1, 8))
        seqs, edges, lin, af = random_bubble_graph(rng, n_var=nv, min_ref=1, max_ref=3 * k + 8, p_indel=0.3,
                                                   chain_after={int(rng.integers(-1, nv)): int(rng.integers(1, k + 2))})
        return mode, GraphArrays.from_dicts(seqs, edges, lin, af)
    G = int(rng.integers(40 * k, 400 * k))
    return mode, synthetic_indel_graph(G, max(2, G // int(rng.integers(8, 60))), k=k, seed=int(rng.integers(0, 1 << 30)),
                                       p_del=float(rng.uniform(0, 0.4)), p_ins=float(rng.uniform(0, 0.4)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    t_end = time.time() + args.seconds
    it = checked = refused = 0
    while time.time() < t_end:
        seed = args.seed * 1_000_003 + it
        it += 1
        rng = np.random.default_rng(seed)
        k = int(rng.integers(2, 32))
        M = int(rng.choice([0, 1, 2, 3, 4, 5, 100]))
        one = bool(rng.integers(0, 2))
        try:
            mode, g = make_graph(rng, k)
        except AssertionError:
            continue
        desc = "seed %d mode k=%d M=%d one=%s" % (seed, k, M, one)
        try:
            crit = oracle.critical_paths(g, k)
        except oracle.OracleError:
            continue                                              # reference crash E2
        kw = {}
        if len(crit[0]) > 2 and rng.random() < 0.4:
            a = int(rng.integers(0, len(crit[0])))
            b = int(rng.integers(a, len(crit[0]) + 1))
            kw = dict(start_at_critical_path_number=a, stop_at_critical_path_number=b)
        try:
            full, flags = oracle.find(g, k, crit, one, M, return_flags=True, **kw)
        except oracle.OracleError:
            continue                                              # the reference hits its recursion limit here
        wl = None
        if len(full["kmers"]) and rng.random() < 0.25:
            wl = set(int(x) for x in full["kmers"][rng.random(len(full["kmers"])) < 0.5])
            full = oracle.find(g, k, crit, one, M, whitelist=wl, **kw)
        cp = CriticalGraphPaths(crit[0], crit[1])
        f = DenseKmerFinder(g, k, critical_graph_paths=cp, only_save_one_node_per_kmer=one, max_variant_nodes=M, whitelist=wl, **kw)
        try:
            f.find()
        except (ValueError, NotImplementedError) as e:
            if flags & oracle.ORC_FLAG_UNDEFINED_BULK or isinstance(e, NotImplementedError):
                refused += 1
                continue
            print("UNEXPECTED refusal:", desc, mode, kw, e); sys.exit(1)
        if flags & oracle.ORC_FLAG_UNDEFINED_BULK:
            print("library accepted a graph the oracle flags as undefined:", desc, mode, kw); sys.exit(1)
        try:
            assert_same_records(finder_cols(f), full)
        except AssertionError as e:
            print("MISMATCH:", desc, mode, kw, "whitelist" if wl else "", e); sys.exit(1)
        f.close()
        checked += 1
    print("soak ok: %d graphs compared, %d refused (undefined in the reference / unsupported), %d drawn" % (checked, refused, it))


if __name__ == "__main__":
    main()
