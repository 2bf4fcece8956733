#!/usr/bin/env python3
"""Headline benchmark: k-mers hashed+indexed per second, k=31, synthetic 3 Gbp linear-ref obgraph + 5 M SNP
bubbles (BASELINE.json configs[2]), DenseKmerFinder -> FlatKmers columns resident in HBM (what the reference's
`graph_kmer_index index` command computes, command_line_interface.py:553-622).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment, as
torch.distributed.run exports them).  Every rank holds the whole graph and runs the critical-path range
`sharding.shard_range` gives it (strong scaling: the 3 Gbp graph is fixed); there is no collective in the timed
region.  The barrier and the max-over-ranks of the timing go over `parallel.SocketControlPlane` (plain TCP to rank 0);
neither torch nor any other framework is imported -- the compute path is libgki_hip.so through ctypes.

A step = gki_finder_count (boundary count kernel + prefix sums) + gki_finder_emit_flat (interior + boundary emit
kernels) over the rank's shard, inputs (graph arrays) already in HBM.  Rank 0 prints one JSON line.  At N=1 the same
line carries three secondary records measured after the timed region on the step's own output and graph: `index_build`
(CollisionFreeKmerIndex.from_flat_kmers of the variant index, collision_free_kmer_index.py:423-467) and `read_mapping`
(BASELINE configs[4]: reads -> k-mers of both strands -> CollisionFreeKmerIndex.get -> node counts,
read_kmers.py:67-70, collision_free_kmer_index.py:303-315) and `early_stop_search` (the batched
find_only_kmers_starting_at_position in UniqueVariantKmersFinder's call pattern, unique_variant_kmers.py:119-140).
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
    path = os.path.join(ROOT, "profiles", "r02_pmc_3gbp.json")
    try:
        with open(path) as fh:
            d = json.load(fh)
        w = d["workload"]
        if (w["n_ref_bases"], w["n_snp_bubbles"], w["k"]) != (n_ref_bases, n_sites, k):
            return None, None
        return d["dominant_kernel_traffic_bytes_per_launch"], "profiles/r02_pmc_3gbp.json"
    except (OSError, KeyError, ValueError):
        return None, None


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def secondary_records(lib, _lib, g, k, finder, out, n_reads, modulo=452930477, max_hits=10):
    """Index build and read mapping on the step's output (N=1, after the timed region).  Index = the records whose
    window crosses a node boundary, i.e. the KAGE-like variant index of SURVEY.md 8(d) C5 = the boundary section of
    the split layout."""
    import ctypes as C
    from graph_kmer_index_amd.flat_kmers import DeviceFlatKmers
    from graph_kmer_index_amd.collision_free_kmer_index import DeviceIndex
    from graph_kmer_index_amd.graph import synthetic_haplotype_sequence
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_reads import make_reads

    def sync():
        _lib.check(lib.gki_device_synchronize())

    n_int = finder.interior_records()
    nb = out.n - n_int
    bnd = DeviceFlatKmers(nb, out.hashes.view(n_int, nb), out.nodes.view(n_int, nb), out.ref_offsets.view(n_int, nb),
                          out.allele_frequencies.view(n_int, nb))
    idx = None
    for _ in range(2):                                   # the second build is the measurement (first: pool warm-up)
        if idx is not None:
            idx.free()
        sync()
        t = time.perf_counter()
        idx = DeviceIndex.build(bnd, modulo)
        sync()
        dt = time.perf_counter() - t
    passes = -(-int(modulo - 1).bit_length() // 8)
    # algorithmic bytes of the build as implemented (csrc/gki_index.hip), per record: bucket keys 8 R + 8 W; per radix
    # pass 4 R (histogram) + 8 R + 8 W (ranked scatter of key/index pairs); row pack 24 R + 32 W; row gather 4 + 32 R +
    # 24 W; directory 4 R; frequencies 16 R + 2 W; plus the directory itself, 2 x 4 B x modulo written once
    per_record = 16 + 20 * passes + 56 + 60 + 4 + 18
    moved = per_record * nb + 8 * modulo
    index_build = {"records": int(nb), "ms": 1e3 * dt, "records_per_s": nb / dt, "modulo": modulo, "radix_passes": passes,
                   "frequencies": True, "bytes_moved_model": int(moved), "bytes_per_record_model": per_record,
                   "achieved_GBps": moved / dt / 1e9, "frac_of_hbm_peak": moved / dt / 1e9 / HBM_PEAK_GBS,
                   "timed": "wall clock around DeviceIndex.build incl. its allocations, device synchronised"}
    # scalar CollisionFreeKmerIndex.get (:303-315): one launch + one synchronisation per call (gki_index_get_small)
    some = bnd.hashes.view(0, min(nb, 4096)).to_host()
    t = time.perf_counter()
    for x in some[:2000]:
        idx.get_small([int(x)], 10)
    index_build["scalar_get_calls_per_s"] = min(len(some), 2000) / (time.perf_counter() - t)
    # ReverseKmerIndex.from_flat_kmers (reverse_kmer_index.py:47-83) of the same records: the same
    # stable radix sort keyed on the node id (gki_reverse_index_build), columns in HBM on both sides
    r_pos, r_cnt = _lib.DeviceArray(g.n_nodes, np.uint32), _lib.DeviceArray(g.n_nodes, np.uint16)
    r_kmers, r_refs = _lib.DeviceArray(nb, np.uint64), _lib.DeviceArray(nb, np.uint64)
    for _ in range(2):                                   # the second build is the measurement
        sync()
        t = time.perf_counter()
        _lib.check(lib.gki_reverse_index_build(bnd.nodes.ptr, bnd.hashes.ptr, bnd.ref_offsets.ptr, nb, g.n_nodes, r_pos.ptr,
                                               r_cnt.ptr, r_kmers.ptr, r_refs.ptr))
        sync()
        dt_rev = time.perf_counter() - t
    index_build["reverse_index"] = {"ms": 1e3 * dt_rev, "records_per_s": nb / dt_rev, "n_nodes": int(g.n_nodes),
                                    "run_lengths_sum_to_records": bool(int(r_cnt.to_host().astype(np.int64).sum()) == nb)}
    for b in (r_pos, r_cnt, r_kmers, r_refs):
        b.free()
    log("index build: %d records in %.1f ms; scalar get %.0f calls/s; reverse index %.1f ms" % (nb, 1e3 * dt, index_build["scalar_get_calls_per_s"], 1e3 * dt_rev))

    t = time.perf_counter()
    letters = make_reads(synthetic_haplotype_sequence(g), n_reads, np.random.default_rng(99))
    t_reads = time.perf_counter() - t
    d_letters = _lib.DeviceArray.from_host(letters)
    d_start = _lib.DeviceArray.from_host(np.arange(n_reads + 1, dtype=np.int64) * 150)
    table = idx.probe_table()
    counts = _lib.DeviceArray(g.n_nodes, np.uint32)
    nk, nh = C.c_int64(0), C.c_int64(0)
    for _ in range(3):                                   # the last launch is the measurement
        counts.zero()
        sync()
        t = time.perf_counter()
        _lib.check(lib.gki_probe_reads_count_nodes(table, d_letters.ptr, d_start.ptr, n_reads, k, 3, max_hits, counts.ptr,
                                                   g.n_nodes, C.byref(nk), C.byref(nh)))
        sync()
        dt_map = time.perf_counter() - t
    rate = C.c_double(0.0)
    _lib.check(lib.gki_measure_random_loads(2 << 30, 1 << 31, C.byref(rate)))       # ~40 ms on a 2 GB table
    sectors = nk.value + nh.value                        # one directory sector per k-mer + at least one row sector per hit
    read_mapping = {"reads": n_reads, "read_length": 150, "strands": 2, "kmers": nk.value, "hits": nh.value,
                    "ms": 1e3 * dt_map, "kmers_per_s": nk.value / dt_map, "reads_per_s": n_reads / dt_map,
                    "sectors_per_s_lower_bound": sectors / dt_map, "random_loads_per_s_measured": rate.value,
                    "frac_of_random_request_rate": sectors / dt_map / rate.value if rate.value else None,
                    "index_records": int(nb), "kernel": "k_probe_reads (letters -> both strands -> probe -> node counts, fused)",
                    "note": "BASELINE configs[4] names 1e8 reads; %d are mapped here so that the default run stays within "
                            "minutes (their host-side simulation takes %.0f s); the rate does not depend on the count"
                            % (n_reads, t_reads)}
    log("read mapping: %d reads, %.3g k-mers/s; random-request rate %.3g/s" % (n_reads, nk.value / dt_map, rate.value))
    for b in (d_letters, d_start, counts):
        b.free()
    idx.free()
    return index_build, read_mapping, early_stop_record(lib, _lib, g, k, finder)


def early_stop_record(lib, _lib, g, k, finder, max_variant_nodes=4):
    """SURVEY.md 8(f) row 4 on the step's own graph: the batched early-stop search in UniqueVariantKmersFinder's call
    pattern (unique_variant_kmers.py:119-140) -- seven find_only_kmers_starting_at_position per SNP site, 2, 6, ... 26
    bases before the variant, constructor defaults (all window nodes, max_variant_nodes 4) -- as one batch through
    gki_forward_count + gki_forward_emit.  Same generator as tools/bench_forward.py."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_forward import start_positions
    nodes, offs = start_positions(g, k)
    n_pos = len(nodes)
    graph = finder._device_graph()
    d_nodes, d_offs = _lib.DeviceArray.from_host(nodes), _lib.DeviceArray.from_host(offs)
    d_start = _lib.DeviceArray(n_pos + 1, np.int64)
    n = C.c_int64(0)
    head = (graph.handle, k, max_variant_nodes, 0, None, d_nodes.ptr, d_offs.ptr, n_pos)
    _lib.check(lib.gki_forward_count(*head, d_start.ptr, C.byref(n)))
    n_rec = n.value
    bufs = [_lib.DeviceArray(max(1, n_rec), d) for d in (np.int64, np.int32, np.int16, np.int32, np.float64)]
    times = []
    for _ in range(4):                                   # the first pass warms the pool
        t = time.perf_counter()
        _lib.check(lib.gki_forward_count(*head, d_start.ptr, C.byref(n)))             # synchronous: returns the total
        _lib.check(lib.gki_forward_emit(*head, d_start.ptr, *[b.ptr for b in bufs]))  # synchronises before returning
        times.append(time.perf_counter() - t)
    dt = float(np.median(times[1:]))
    first_counts = np.diff(d_start.to_host(min(n_pos, 1 << 20) + 1))
    rec = {"start_positions": int(n_pos), "records": int(n_rec), "ms": 1e3 * dt, "start_positions_per_s": n_pos / dt,
           "records_per_s": n_rec / dt, "only_save_one_node_per_kmer": False, "max_variant_nodes": max_variant_nodes,
           "every_start_has_a_record": bool(first_counts.min() >= 1),
           "workload": "seven early-stop searches per SNP site (unique_variant_kmers.py:119-140), one batch",
           "timed": "wall clock around gki_forward_count + gki_forward_emit, graph / start arrays / output columns in HBM"}
    log("early-stop search: %d start positions, %d records in %.2f ms" % (n_pos, n_rec, 1e3 * dt))
    for b in bufs + [d_nodes, d_offs, d_start]:
        b.free()
    return rec


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
    ap.add_argument("--nested", type=float, default=0.0, help="diagnostic: this fraction of the sites get an alternative allele "
                    "that contains a SNP itself (nodes with no linear-ref predecessor: the general kernels); "
                    "max_variant_nodes is raised to 8 so that the reference's linear-successor assertion stays quiet")
    ap.add_argument("--all-nodes", action="store_true", help="diagnostic: only_save_one_node_per_kmer=False")
    ap.add_argument("--pretend-shard", default=None, help="diagnostic: R/W -> run only rank R's shard of W on this one GPU")
    ap.add_argument("--general", action="store_true", help="diagnostic: run the general-graph kernel variants (node flags of "
                    "gki_classify_nodes) on this graph, which does not need them: what the flags cost")
    ap.add_argument("--verify", action="store_true", help="size-independent checks on the full output (slow)")
    ap.add_argument("--reads", type=float, default=4e6, help="reads of the read_mapping record (0: skip the secondary records)")
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
    from graph_kmer_index_amd.parallel import SocketControlPlane
    plane = SocketControlPlane(rank, world)           # barrier + max / sum of two scalars; a no-op at world 1

    from graph_kmer_index_amd import _lib, DenseKmerFinder, CriticalGraphPaths, DeviceGraph
    from graph_kmer_index_amd.graph import synthetic_snp_graph, synthetic_linear_graph, synthetic_indel_graph, synthetic_nested_graph
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
    elif args.nested > 0:
        g = synthetic_nested_graph(G, S, k=k, seed=1234, p_nest=args.nested)
        args.max_variant_nodes = max(args.max_variant_nodes, 8)
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

    finder._force_general_kernels = args.general

    def barrier():
        _lib.check(lib.gki_device_synchronize())
        plane.barrier()

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

    elapsed = max(plane.allgather_float(elapsed))
    n_total = sum(plane.allgather_int(n_local))

    checks = None
    if args.verify:
        checks = verify(out, g, k, n_interior)
    secondary = None
    if (world == 1 and args.reads > 0 and not (args.linear or args.indels or args.nested or args.all_nodes or args.pretend_shard or args.general)
            and out.n - n_interior > 0):
        secondary = secondary_records(lib, _lib, g, k, finder, out, int(args.reads))

    if rank == 0:
        ms_step = 1000.0 * elapsed / args.steps
        value = n_total * args.steps / elapsed
        avg_int_ms = float(np.mean(interior_ms))
        achieved = BYTES_PER_RECORD * n_interior / (avg_int_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(G, S, k) if world == 1 and not (args.linear or args.indels or args.nested) else (None, None)
        res = {
            "metric": "k-mers hashed+indexed per second (k=31, 3 Gbp graph)", "value": value, "unit": "k-mers/s",
            "timed_region": "enumerate + hash + FlatKmers rows in HBM (gki_finder_count + gki_finder_emit_flat); the index "
                            "build and the read side are the separate records index_build / read_mapping below",
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
        if secondary is not None:
            res["index_build"], res["read_mapping"], res["early_stop_search"] = secondary
        res["cpu_baseline"] = cpu
        print(json.dumps(res), flush=True)
    plane.barrier()
    plane.close()


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
