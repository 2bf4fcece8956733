#!/usr/bin/env python3
"""Secondary benchmark (not the headline): the batched early-stop search, SURVEY.md 8(f) row 4.
  python tools/bench_forward.py --bases 1e9 --sites 1.67e6
Workload = what UniqueVariantKmersFinder does per variant (unique_variant_kmers.py:119-140): for every SNP site of the
synthetic graph, one `find_only_kmers_starting_at_position` from each of the linear-ref positions 2, 6, ... 26 bases
before the variant ([variant.position - i for i in range(2, k-2)][::4], k=31: seven starts per variant), here as ONE
batch: gki_forward_count + gki_forward_emit over all start positions, start arrays and the graph resident in HBM, output
columns allocated once.  A step = count + emit of the whole batch.  Checks (size-independent): every start yields at
least one record; every record's k-mer starts with the start position's own bases; emit fills exactly the counted slots.
Prints one JSON object."""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from graph_kmer_index_amd import _lib, DenseKmerFinder
from graph_kmer_index_amd.graph import synthetic_snp_graph


def start_positions(g, k):
    """(nodes int32, offsets int32): per SNP site the positions i = 2, 6, ... < k-2 bases before the variant base, on the
    ref segment in front of the bubble (sites whose segment is shorter than i are skipped for that i)."""
    alt = np.nonzero((g.is_ref == 0) & (g.node_size == 1))[0]
    has_one_pred = (g.rev_start[alt + 1] - g.rev_start[alt]) == 1
    alt = alt[has_one_pred]
    seg = g.rev_edges[g.rev_start[alt]].astype(np.int64)             # the segment node in front of the bubble
    size = g.node_size[seg].astype(np.int64)
    nodes, offs = [], []
    for i in list(range(2, k - 2))[::4][::-1]:
        ok = size >= i
        nodes.append(seg[ok]); offs.append(size[ok] - i)
    nodes = np.concatenate(nodes); offs = np.concatenate(offs)
    order = np.argsort(nodes * 64 + (offs & 63), kind="stable")       # site order, like a loop over the variants
    return nodes[order].astype(np.int32), offs[order].astype(np.int32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bases", type=float, default=1e9)
    ap.add_argument("--sites", type=float, default=1.67e6)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--max-variant-nodes", type=int, default=4)       # constructor default, kmer_finder.py
    ap.add_argument("--one-node", action="store_true", help="only_save_one_node_per_kmer=True")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--verify", action="store_true", help="copy the records back and run the checks (slow at full size)")
    args = ap.parse_args()
    lib = _lib.load()
    k = args.k
    t0 = time.time()
    g = synthetic_snp_graph(int(args.bases), int(args.sites), k=k, seed=1234)
    nodes, offs = start_positions(g, k)
    n_pos = len(nodes)
    finder = DenseKmerFinder(g, k, only_save_one_node_per_kmer=args.one_node, max_variant_nodes=args.max_variant_nodes)
    graph = finder._device_graph()
    t_prep = time.time() - t0
    d_nodes = _lib.DeviceArray.from_host(nodes)
    d_offs = _lib.DeviceArray.from_host(offs)
    d_start = _lib.DeviceArray(n_pos + 1, np.int64)
    n = C.c_int64(0)
    head = (graph.handle, k, args.max_variant_nodes, int(args.one_node), None, d_nodes.ptr, d_offs.ptr, n_pos)
    _lib.check(lib.gki_forward_count(*head, d_start.ptr, C.byref(n)))
    n_rec = n.value
    dt = [np.int64, np.int32, np.int16, np.int32, np.float64]
    bufs = [_lib.DeviceArray(max(1, n_rec), d) for d in dt]

    def step():
        m = C.c_int64(0)
        _lib.check(lib.gki_forward_count(*head, d_start.ptr, C.byref(m)))          # synchronous: returns the total
        _lib.check(lib.gki_forward_emit(*head, d_start.ptr, *[b.ptr for b in bufs]))  # synchronises before returning
        return m.value

    for _ in range(args.warmup):
        step()
    times = []
    for _ in range(args.steps):
        t = time.perf_counter(); m = step(); times.append(time.perf_counter() - t)
        assert m == n_rec
    sec = float(np.median(times))
    checks = None
    if args.verify:
        rec_start = d_start.to_host()
        per = np.diff(rec_start)
        kmers = bufs[0].to_host(n_rec)
        first = rec_start[:-1]
        # the first min(avail, k) bases of every record of a start are that start's own bases
        avail = np.minimum(g.node_size[nodes].astype(np.int64) - offs, k)
        base = g.seq_start[nodes].astype(np.int64) + offs
        own = np.zeros(n_pos, dtype=np.int64)
        for j in range(int(avail.max())):
            use = avail > j
            own[use] |= g.seq[base[use] + j].astype(np.int64) << (2 * j)
        mask = (np.int64(1) << (2 * avail)) - 1
        owner = np.repeat(np.arange(n_pos), per)
        checks = {"every_start_has_a_record": bool(per.min() >= 1), "records_start_with_the_start_bases": bool(np.all((kmers & mask[owner]) == own[owner])),
                  "rec_start_is_a_prefix_sum": bool(first[0] == 0 and rec_start[-1] == n_rec)}
    print(json.dumps({
        "metric": "early_stop_search_starts_per_s", "value": n_pos / sec, "unit": "start positions/s",
        "records_per_s": n_rec / sec, "ms_per_step": 1e3 * sec, "ms_per_step_all": [round(1e3 * t, 3) for t in times],
        "steps": args.steps, "warmup": args.warmup, "dtype": "int64",
        "config": {"workload": "seven early-stop searches per SNP site (unique_variant_kmers.py:119-140 pattern) over the synthetic %.3g bp + %.3g SNP graph, one batch"
                               % (args.bases, args.sites), "k": k, "max_variant_nodes": args.max_variant_nodes,
                   "only_save_one_node_per_kmer": bool(args.one_node), "start_positions": int(n_pos), "records": int(n_rec),
                   "records_per_start": n_rec / n_pos},
        "timed_region": "gki_forward_count + gki_forward_emit of the whole batch; graph, start arrays and output columns resident in HBM",
        "graph_and_positions_prepare_s": t_prep, "checks": checks}))
    for b in bufs + [d_nodes, d_offs, d_start]:
        b.free()


if __name__ == "__main__":
    main()
