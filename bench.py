#!/usr/bin/env python3
"""Headline benchmark: k-mers hashed+indexed per second, k=31, synthetic 3 Gbp linear-ref obgraph + 5 M SNP
bubbles (BASELINE.json configs[2]), DenseKmerFinder -> FlatKmers columns resident in HBM (what the reference's
`graph_kmer_index index` command computes, command_line_interface.py:553-622).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the environment under torch.distributed.run).  Every rank
holds the whole graph and runs the critical-path range `sharding.shard_range` gives it (strong scaling: the 3 Gbp
graph is fixed); there is no collective in the timed region.  torch.distributed (gloo) is used only for the barrier
and the max-over-ranks of the timing -- the compute path is libgki_hip.so through ctypes.

A step = gki_finder_count (boundary count kernel + prefix sums) + gki_finder_emit_flat (interior + boundary emit
kernels) over the rank's shard, inputs (graph arrays) already in HBM.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
BYTES_PER_RECORD = 25            # 1 B of node sequence read + 24 B FlatKmers row written


def pmc_traffic(n_ref_bases, n_sites, k):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary (collected with
    tools/collect_pmc.sh on the same workload; counters cannot be read from inside this process)."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_3gbp.json")
    try:
        with open(path) as fh:
            d = json.load(fh)
        w = d["workload"]
        if (w["n_ref_bases"], w["n_snp_bubbles"], w["k"]) != (n_ref_bases, n_sites, k):
            return None, None
        return d["dominant_kernel_traffic_bytes_per_launch"], "profiles/r01_pmc_3gbp.json"
    except (OSError, KeyError, ValueError):
        return None, None


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


_CPU = {}


def _cpu_chunk(rng_):
    from oracle import oracle
    a, b = rng_
    out = oracle.find(_CPU["g"], _CPU["k"], _CPU["crit"], True, _CPU["M"], start_at_critical_path_number=a,
                      stop_at_critical_path_number=b)
    return len(out["kmers"])


def cpu_baseline(sample_bases, k, max_variant_nodes, cores):
    """The oracle (scalar C restatement of the reference's DFS) on a down-scaled graph from the same generator and seed,
    spread over `cores` processes the way the reference spreads its own work: contiguous ranges of critical-path
    numbers (command_line_interface.py:588-614).  A reported baseline, not the thing measured below.  Runs BEFORE the
    GPU is touched (the workers are forked)."""
    import multiprocessing as mp
    from graph_kmer_index_amd.graph import synthetic_snp_graph
    from graph_kmer_index_amd.sharding import critical_path_cuts
    from graph_kmer_index_amd.critical_graph_paths import CriticalGraphPaths
    from oracle import oracle
    sites = max(1, sample_bases // 600)
    g = synthetic_snp_graph(sample_bases, sites, k=k, seed=1234)
    crit = oracle.critical_paths(g, k)
    # many more chunks than processes: the DFS's visited set makes a chunk's cost superlinear in its size, and small
    # chunks balance the load
    cuts = critical_path_cuts(g, CriticalGraphPaths(crit[0], crit[1]), 16 * cores)
    chunks = [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
    _CPU.update(g=g, k=k, crit=crit, M=max_variant_nodes)
    counts = None
    if cores > 1:
        try:
            with mp.get_context("fork").Pool(cores) as pool:
                pool.map(_cpu_chunk, chunks[:cores], chunksize=1)      # untimed: workers up, graph pages touched
                t0 = time.perf_counter()
                counts = pool.map(_cpu_chunk, chunks, chunksize=1)
                dt = time.perf_counter() - t0
        except (OSError, RuntimeError) as e:                           # no process pool here: one core, stated as such
            log("cpu baseline: process pool unavailable (%s), using one core" % e)
            cores = 1
    if counts is None:
        t0 = time.perf_counter()
        counts = [_cpu_chunk(c) for c in chunks]
        dt = time.perf_counter() - t0
    n = int(sum(counts))
    _CPU.clear()
    return {"value": n / dt, "unit": "k-mers/s", "cores": cores, "kind": "port",
            "sample": "oracle/gki_oracle.c DenseKmerFinder restatement (find + v2 columns), same generator/seed, "
                      "%d ref bases + %d SNP bubbles, %d records in %.1f s wall on %d processes (critical-path chunks)"
                      % (sample_bases, sites, n, dt, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bases", type=float, default=3e9, help="reference bases of the synthetic graph")
    ap.add_argument("--sites", type=float, default=5e6, help="SNP bubbles")
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--max-variant-nodes", type=int, default=5)      # CLI `index` default, command_line_interface.py:637
    ap.add_argument("--cpu-sample-bases", type=float, default=6e8)     # ~20 CPU-seconds on the GPU box
    ap.add_argument("--cpu-cores", type=int, default=0, help="processes of the CPU baseline (0: the host's share, at most 16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--linear", action="store_true", help="diagnostic: linear chain graph without variants (BASELINE configs[1] shape)")
    ap.add_argument("--indels", type=float, default=0.0, help="diagnostic: this fraction of the sites each become 1-bp "
                    "deletions and insertions (north star's SNP/indel mix; empty nodes in the graph)")
    ap.add_argument("--all-nodes", action="store_true", help="diagnostic: only_save_one_node_per_kmer=False")
    ap.add_argument("--pretend-shard", default=None, help="diagnostic: R/W -> run only rank R's shard of W on this one GPU")
    ap.add_argument("--verify", action="store_true", help="size-independent checks on the full output (slow)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("WORLD_SIZE=%d but --gpus=%d; using WORLD_SIZE" % (world, args.gpus))
    cpu = None
    if world == 1 and not args.no_cpu_baseline:       # N=1 only, and before anything touches the GPU
        cores = args.cpu_cores or min(16, len(os.sched_getaffinity(0)))
        cpu = cpu_baseline(int(args.cpu_sample_bases), args.k, args.max_variant_nodes, cores)
        log("cpu baseline: %.3g k-mers/s on %d cores" % (cpu["value"], cpu["cores"]))
    dist = None
    if world > 1:
        import torch.distributed as dist            # gloo: barrier + max of a scalar only
        # gloo prints its connection banner on stdout; keep stdout clean for the one JSON line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    from graph_kmer_index_amd import _lib, DenseKmerFinder, CriticalGraphPaths, DeviceGraph
    from graph_kmer_index_amd.graph import synthetic_snp_graph, synthetic_linear_graph, synthetic_indel_graph
    from graph_kmer_index_amd.sharding import shard_range
    lib = _lib.load()
    _lib.require_device()
    n_dev = _lib.device_count()
    _lib.check(lib.gki_set_device(local_rank % n_dev))

    G, S, k = int(args.bases), int(args.sites), args.k
    t0 = time.perf_counter()
    if args.linear:
        g = synthetic_linear_graph(G, 25000, seed=1234)
    elif args.indels > 0:
        g = synthetic_indel_graph(G, S, k=k, seed=1234, p_del=args.indels, p_ins=args.indels)
    else:
        g = synthetic_snp_graph(G, S, k=k, seed=1234)
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    cp = CriticalGraphPaths.from_graph(g, k)
    t_crit = time.perf_counter() - t0
    t0 = time.perf_counter()
    dg = DeviceGraph(g)
    t_up = time.perf_counter() - t0
    if rank == 0:
        log("graph: %d nodes, %d bases (+%d alt), %d critical points; generate %.1fs, critical paths %.2fs, "
            "upload+prepare %.2fs" % (g.n_nodes, G, len(g.seq) - G, len(cp), t_gen, t_crit, t_up))
    g._device = dg
    shard_r, shard_w = rank, world
    if args.pretend_shard:
        shard_r, shard_w = (int(x) for x in args.pretend_shard.split("/"))
    a, b = shard_range(g, cp, shard_r, shard_w)
    finder = DenseKmerFinder(g, k, critical_graph_paths=cp, only_save_one_node_per_kmer=not args.all_nodes,
                             max_variant_nodes=args.max_variant_nodes,
                             start_at_critical_path_number=a if shard_w > 1 else None,
                             stop_at_critical_path_number=b if shard_w > 1 else None)

    def barrier():
        _lib.check(lib.gki_device_synchronize())
        if dist is not None:
            dist.barrier()

    out = None
    interior_ms = []
    for _ in range(args.warmup):
        out = finder.find_flat_on_device(out)
        finder.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = finder.find_flat_on_device(out)
        finder.synchronize()
        interior_ms.append(finder.kernel_ms(1))
    barrier()
    elapsed = time.perf_counter() - t0
    n_local = out.n
    kern = {name: finder.kernel_ms(i) for i, name in enumerate(["count_boundary", "emit_interior", "emit_boundary", "setup_scans"])}
    n_interior = finder.interior_records()

    if dist is not None:
        import torch
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0])
        nn = torch.tensor([n_local], dtype=torch.int64)
        dist.all_reduce(nn, op=dist.ReduceOp.SUM)
        n_total = int(nn[0])
    else:
        n_total = n_local

    checks = None
    if args.verify:
        checks = verify(out, g, k, n_interior)

    if rank == 0:
        ms_step = 1000.0 * elapsed / args.steps
        value = n_total * args.steps / elapsed
        avg_int_ms = float(np.mean(interior_ms))
        achieved = BYTES_PER_RECORD * n_interior / (avg_int_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(G, S, k) if world == 1 and not args.linear and not args.indels else (None, None)
        res = {
            "metric": "k-mers hashed+indexed per second (k=31, 3 Gbp graph)", "value": value, "unit": "k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: synthetic %.3g bp linear-ref obgraph + %.3g SNP bubbles, k=%d, "
                                   "DenseKmerFinder -> FlatKmers (hash u64, node u32, ref_offset u64, af f32) in HBM"
                                   % (G, S, k),
                       "n_ref_bases": G, "n_snp_bubbles": int(S), "k": k, "max_variant_nodes": args.max_variant_nodes,
                       "only_save_one_node_per_kmer": True, "records_per_step": n_total, "n_nodes": int(g.n_nodes),
                       "sharding": "critical-path ranges balanced by bases, whole graph resident on every GPU"},
            "roofline": {"bound": "hbm", "kernel": "k_emit_interior_runs", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "bytes_per_record": BYTES_PER_RECORD, "records_per_launch": int(n_interior),
                         "avg_launch_ms": avg_int_ms},
            "kernels_ms_rank0_last_step": kern,
            "setup_s": {"generate_graph_host": t_gen, "critical_paths_host": t_crit, "upload_and_prepare": t_up},
        }
        if checks is not None:
            res["verify"] = checks
        res["cpu_baseline"] = cpu
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def verify(out, g, k, n_interior):
    """Size-independent properties of the full-size output (tests/ compare small sizes with the oracle).
    Layout: n_interior records whose window lies inside one node (by position), then the boundary records (by
    end node)."""
    import numpy as np
    n = out.n
    step = max(1, n // 2_000_000)
    h = out.hashes.to_host(n)[::step]
    ro = out.ref_offsets.to_host(n)[::step].astype(np.int64)         # default position id == global base index
    nodes = out.nodes.to_host(n)[::step]
    n_int_s = (n_interior + step - 1) // step                          # sampled records that are interior ones
    node_of = np.searchsorted(g.seq_start, ro, side="right") - 1
    ok_sorted = bool(np.all(np.diff(ro[:n_int_s]) > 0)) and bool(np.all(np.diff(node_of[n_int_s:]) >= 0))
    ok_range = bool(h.max() < 4 ** k)
    off = ro - g.seq_start[node_of]
    ok_split = bool(np.all(off[:n_int_s] >= k - 1)) and bool(np.all(off[n_int_s:] < k - 1))
    # the hash of an interior record equals the k bases read backwards from its end position
    idx = np.arange(0, n_int_s, max(1, n_int_s // 200000))
    win = ro[idx][:, None] - (k - 1) + np.arange(k)[None, :]
    expect = (g.seq[win].astype(np.uint64) << (2 * np.arange(k, dtype=np.uint64))[None, :]).sum(axis=1)
    ok_hash = bool(np.array_equal(expect, h[idx])) and bool(np.array_equal(nodes[idx], node_of[idx].astype(np.uint32)))
    return {"sampled": int(len(h)), "interior_by_position_then_boundary_by_node": ok_sorted,
            "interior_offsets_ge_k-1_boundary_lt_k-1": ok_split, "hash_lt_4^k": ok_range,
            "interior_hash_recomputed": ok_hash}


if __name__ == "__main__":
    main()
