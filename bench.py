#!/usr/bin/env python3
"""Headline benchmark: k-mers hashed+indexed per second, k=31, synthetic 3 Gbp linear-ref obgraph + 5 M SNP
bubbles (BASELINE.json configs[2]), DenseKmerFinder -> FlatKmers columns resident in HBM (what the reference's
`graph_kmer_index index` command computes, command_line_interface.py:553-622).

    python bench.py --gpus N --steps K --warmup W

One process per GPU.  Under a launcher (torch.distributed.run exports RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT) this process IS a rank.  Without one, `--gpus N` with N > 1 makes this process the launcher: before it
touches the GPU in any way it generates the graph once, stores it under /dev/shm, starts N fresh rank processes
(`subprocess.Popen` of this file with the rank environment; no exec of a process that has initialised the GPU), and
exits with the worst of their exit codes; rank 0's JSON line goes to its stdout.  Ranks map LOCAL_RANK onto the visible
devices round-robin, so `--gpus 2` rehearses on a one-GPU box.  The graph is generated once per node, not per rank.

Every rank holds the whole graph and runs the critical-path range `sharding.shard_range` gives it (strong scaling: the
3 Gbp graph is fixed).  The barrier and the max-over-ranks of the timing go over `parallel.SocketControlPlane` (plain TCP
to rank 0); neither torch nor any other framework is imported -- the compute path is libgki_hip.so through ctypes.

A step = gki_finder_count (boundary count kernel + prefix sums) + gki_finder_emit_flat (interior + boundary emit
kernels) over the rank's shard, inputs (graph arrays) already in HBM; there is no collective inside it (the
enumeration shards without a halo).  Rank 0 prints one JSON line.  Records measured after the timed region:
  N = 1: `index_build` (CollisionFreeKmerIndex.from_flat_kmers of the variant index, collision_free_kmer_index.py:423-467),
         `full_index` (ALL records of the step indexed: bucket-range slices, since one int32 directory stops at 2^31),
         `read_mapping` (BASELINE configs[4]: reads -> k-mers of both strands -> CollisionFreeKmerIndex.get -> node
         counts, read_kmers.py:67-70, collision_free_kmer_index.py:303-315), `early_stop_search` (the batched
         find_only_kmers_starting_at_position in UniqueVariantKmersFinder's call pattern, unique_variant_kmers.py:119-140),
         each with the oracle's rate on the host cores beside it;
  N > 1: `sharded_build` (BASELINE configs[3]): the exchange and the table build with ranks -- all-gather(v) of the
         variant-index records + CollisionFreeKmerIndex build on every rank (north_star's wording), and the
         bucket-range partitioned build of ALL records (partition -> all-to-all(v) -> slice build), per phase the
         slowest rank's time and the bytes per xGMI link.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
XGMI_LINK_GBS = 153.0            # per direction and link (7 links per GPU, fully connected)
BYTES_PER_RECORD = 25            # 1 B of node sequence read + 24 B FlatKmers row written
PMC_FILE = "profiles/r04_pmc_3gbp.json"
FULL_INDEX_GROUP_BITS = 7         # --full-index-group-bits
FULL_INDEX_ROWS = True             # --full-index-columns turns it off


def pmc_traffic(n_ref_bases, n_sites, k):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary (collected with
    tools/collect_pmc.sh on the same workload; counters cannot be read from inside this process)."""
    for name in (PMC_FILE, "profiles/r03_pmc_3gbp.json"):
        try:
            with open(os.path.join(ROOT, name)) as fh:
                d = json.load(fh)
            w = d["workload"]
            if (w["n_ref_bases"], w["n_snp_bubbles"], w["k"]) != (n_ref_bases, n_sites, k):
                continue
            return d["dominant_kernel_traffic_bytes_per_launch"], name + (" @ " + d["commit"] if "commit" in d else "")
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def pmc_index_traffic(shape, n):
    """HBM bytes per build of the index kernels from the committed PMC summary (tools/collect_pmc_index.sh, random records
    of the same shape: `sparse` = the variant index, `dense` = one slice of the whole-genome index), or (None, None)."""
    try:
        with open(os.path.join(ROOT, "profiles/r04_pmc_index.json")) as fh:
            d = json.load(fh)
        if abs(d[shape]["records"] - n) > 0.02 * n:
            return None, None
        return d[shape]["traffic_bytes_per_build"], "profiles/r04_pmc_index.json @ %s (%s, %d random records)" % (d.get("commit"), shape, d[shape]["records"])
    except (OSError, KeyError, ValueError):
        return None, None


def index_roofline(n, n_buckets, nonempty, seconds, hashing, passes=1):
    """SURVEY.md 8(d)'s ALGORITHMIC bytes of the index build: every 24-byte row read and written once per sort pass (the
    lower bound is one pass; the implementation's count is stated beside it), the frequency column 2 B per record, the
    directory initialised (2 x 4 B x buckets) and its non-empty entries written (8 B each); with `hashing` the 25 B per
    record of enumerate + hash (1 B of sequence read, the 24-byte row written) in front."""
    per_record = (BYTES_PER_RECORD if hashing else 0) + passes * 48 + 2
    alg = per_record * n + 8 * n_buckets + 8 * nonempty
    return {"bound": "hbm", "achieved": alg / seconds / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": alg / seconds / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes": int(alg),
            "bytes_per_record": per_record, "sort_passes_counted": passes,
            "model": "SURVEY.md 8(d): %s%d pass x 2 x 24 B + 2 B frequency per record, + 8 B x %d buckets + 8 B x %d non-empty"
                     % ("25 B enumerate+hash + " if hashing else "", passes, n_buckets, nonempty)}


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def make_graph(args):
    from graph_kmer_index_amd.graph import synthetic_snp_graph, synthetic_linear_graph, synthetic_indel_graph, synthetic_nested_graph
    G, S, k = int(args.bases), int(args.sites), args.k
    if args.linear:
        return synthetic_linear_graph(G, 25000, seed=1234)
    if args.indels > 0:
        return synthetic_indel_graph(G, S, k=k, seed=1234, p_del=args.indels, p_ins=args.indels)
    if args.nested > 0:
        return synthetic_nested_graph(G, S, k=k, seed=1234, p_nest=args.nested)
    return synthetic_snp_graph(G, S, k=k, seed=1234)


# ------------------------------------------------------------------------------------------ launcher (no GPU here)
def _free_port_pair():
    import socket
    for _ in range(64):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        p = s.getsockname()[1]
        s.close()
        if p >= 32768 or p + 1 > 65535:       # stay below the ephemeral range: a dialling rank must not grab the port
            p = 20000 + (p * 7919 + os.getpid()) % 10000
        ok = True
        for q in (p, p + 1):
            t = socket.socket()
            try:
                t.bind(("127.0.0.1", q))
            except OSError:
                ok = False
            t.close()
        if ok:
            return p
    raise RuntimeError("no free port pair on 127.0.0.1")


def launch_ranks(args):
    """--gpus N without a launcher's environment: this process starts the N ranks.  It never loads libgki_hip.so and
    never touches the GPU."""
    world = args.gpus
    t0 = time.perf_counter()
    g = make_graph(args)
    gdir = "/dev/shm/gki_bench_%d" % os.getpid()
    if not os.path.isdir("/dev/shm"):
        import tempfile
        gdir = os.path.join(tempfile.gettempdir(), "gki_bench_%d" % os.getpid())
    g.to_dir(gdir)
    del g
    log("launcher: graph generated once and stored under %s in %.1f s; starting %d ranks" % (gdir, time.perf_counter() - t0, world))
    port = _free_port_pair()
    token = "bench-%d-%d" % (os.getpid(), int(time.time()))
    procs = []
    try:
        for r in range(world):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GKI_JOB_TOKEN=token, GKI_BENCH_GRAPH_DIR=gdir,
                       GKI_BENCH_GRAPH_SECONDS="%.3f" % (time.perf_counter() - t0))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
        rc = 0
        while any(p.poll() is None for p in procs):
            time.sleep(0.2)
            bad = [p for p in procs if p.poll() not in (None, 0)]
            if bad:                                       # one rank failed: the others would wait for it at a barrier
                # rank 0 prints the line (its watchdog fires on the same timer as the other ranks'): give it time to
                # do so before the survivors are ended
                t_bad = time.perf_counter()
                while procs[0].poll() is None and time.perf_counter() - t_bad < 20.0:
                    time.sleep(0.2)
                for p in procs:
                    if p.poll() is None:
                        p.kill()
        rc = max(abs(p.wait()) for p in procs)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(gdir, ignore_errors=True)
    sys.exit(rc)


def node_graph(args, plane, rank, local_rank):
    """The graph of this node: generated by one rank (or by the launcher) and mapped by the others."""
    from graph_kmer_index_amd.graph import GraphArrays
    t0 = time.perf_counter()
    gdir = os.environ.get("GKI_BENCH_GRAPH_DIR")
    if gdir:                                              # the launcher stored it
        g = GraphArrays.from_dir(gdir)
        return g, float(os.environ.get("GKI_BENCH_GRAPH_SECONDS", "0")), time.perf_counter() - t0, None
    if plane.world == 1:
        g = make_graph(args)
        return g, time.perf_counter() - t0, 0.0, None
    # under an external launcher: local rank 0 generates, everyone else waits at the barrier and maps
    base = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    gdir = os.path.join(base, "gki_bench_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "x")))
    t_gen = 0.0
    if local_rank == 0:
        shutil.rmtree(gdir, ignore_errors=True)
        g = make_graph(args)
        t_gen = time.perf_counter() - t0
        g.to_dir(gdir)
    plane.barrier()
    t1 = time.perf_counter()
    if local_rank != 0:
        g = GraphArrays.from_dir(gdir)
    return g, t_gen, time.perf_counter() - t1, (gdir if local_rank == 0 else None)


def adapter_times(g, args):
    """GraphArrays.from_obgraph at this graph's size: an obgraph-like object over the synthetic graph
    (tools/obgraph_like.py: the accessor methods + obgraph's whole-array attributes), (a) through its whole arrays --
    the sample check against the accessors included --, (b) node by node through the accessors on a 2e5-node graph of
    the same kind, scaled by nodes (the full walk takes minutes: the reason for (a))."""
    if args.linear or args.indels or args.nested:
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from obgraph_like import ObgraphLike
    from graph_kmer_index_amd.graph import GraphArrays, synthetic_snp_graph
    t = time.perf_counter()
    got = GraphArrays.from_obgraph(ObgraphLike(g))
    t_fast = time.perf_counter() - t
    same = all(np.array_equal(getattr(got, c), getattr(g, c)) for c in ("node_size", "edge_start", "edges", "is_ref", "rev_start", "rev_edges"))
    small = synthetic_snp_graph(40_000_000, 66_667, k=args.k, seed=1234)
    t = time.perf_counter()
    GraphArrays._from_obgraph_accessors(ObgraphLike(small, with_arrays=False))
    t_small = time.perf_counter() - t
    return {"nodes": int(g.n_nodes), "whole_arrays_s": t_fast, "equals_the_graph": bool(same),
            "accessor_walk_s_scaled": t_small * g.n_nodes / small.n_nodes,
            "accessor_walk_measured": {"nodes": int(small.n_nodes), "s": t_small}}


# ------------------------------------------------------------------------------------------ CPU baselines (oracle)
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


def cpu_baselines_secondary(k, max_hits=10):
    """The oracle's own rate for the three secondary records, one host core each, on down-scaled inputs from the same
    generators: oracle.index_build (records/s), oracle.map_reads = loop of read_kmers + index_get (k-mers/s),
    oracle.find_from_positions = loop of find_only_kmers_starting_at_position (start positions/s)."""
    from graph_kmer_index_amd.graph import synthetic_snp_graph, synthetic_haplotype_sequence
    from oracle import oracle
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_reads import make_reads
    from bench_forward import start_positions
    out = {}
    g = synthetic_snp_graph(12_000_000, 20_000, k=k, seed=1234)
    rec = oracle.find(g, k, None, True, 5)
    bnd = rec["start_offsets"] < k - 1                       # windows that reach into a predecessor: the variant index
    kmers = rec["kmers"][bnd].astype(np.uint64)
    nodes = rec["nodes"][bnd].astype(np.uint32)
    refs = (g.seq_start[rec["start_nodes"][bnd]] + rec["start_offsets"][bnd]).astype(np.uint64)
    af = rec["allele_frequencies"][bnd].astype(np.float32)
    n = len(kmers)
    modulo = int(n / 0.685) | 1                              # the bench's records per bucket (3.1e8 / 452930477)
    t0 = time.perf_counter()
    ix = oracle.index_build(kmers, nodes, refs, af, modulo=modulo)
    dt = time.perf_counter() - t0
    out["index_build"] = {"value": n / dt, "unit": "records/s", "cores": 1, "kind": "port",
                          "sample": "oracle.index_build (stable merge sort + directory + frequencies): %d variant-index records "
                                    "of a 12 Mbp + 20 000 SNP graph, modulo %d (same records per bucket as the bench), %.2f s" % (n, modulo, dt)}
    n_reads = 20000
    letters = make_reads(synthetic_haplotype_sequence(g), n_reads, np.random.default_rng(99))
    rs = np.arange(n_reads + 1, dtype=np.int64) * 150
    t0 = time.perf_counter()
    _, nk, nh = oracle.map_reads(ix, letters, rs, k, g.n_nodes, 3, max_hits)
    dt = time.perf_counter() - t0
    out["read_mapping"] = {"value": nk / dt, "unit": "k-mers/s", "cores": 1, "kind": "port",
                           "sample": "oracle.map_reads (read_kmers on both strands + index_get per k-mer + node counts in C): %d reads "
                                     "x 150 bp against that index, %d k-mers, %d hits, %.2f s" % (n_reads, nk, nh, dt)}
    sn, so = start_positions(g, k)
    t0 = time.perf_counter()
    nrec = oracle.find_from_positions(g, k, sn, so, False, 4)
    dt = time.perf_counter() - t0
    out["early_stop_search"] = {"value": len(sn) / dt, "unit": "start positions/s", "cores": 1, "kind": "port",
                                "sample": "oracle.find_from_positions (loop of find_only_kmers_starting_at_position): %d start "
                                          "positions of that graph, %d records, %.2f s" % (len(sn), nrec, dt)}
    return out


# ------------------------------------------------------------------------------------------ N = 1 secondary records
def secondary_records(lib, _lib, g, k, cp, finder, out, n_reads, cpu2, max_variant_nodes, full, modulo=452930477, max_hits=10):
    """Index build, full index, read mapping and early-stop search on the step's output (N=1, after the timed region).
    Variant index = the records whose window crosses a node boundary, i.e. the KAGE-like index of SURVEY.md 8(d) C5 =
    the boundary section of the split layout."""
    import ctypes as C
    from graph_kmer_index_amd.flat_kmers import DeviceFlatKmers
    from graph_kmer_index_amd.collision_free_kmer_index import DeviceIndex
    from graph_kmer_index_amd.graph import synthetic_haplotype_sequence

    def sync():
        _lib.check(lib.gki_device_synchronize())

    cpu2 = cpu2 or {}
    n_int = finder.interior_records()
    nb = out.n - n_int
    bnd = DeviceFlatKmers(nb, out.hashes.view(n_int, nb), out.nodes.view(n_int, nb), out.ref_offsets.view(n_int, nb),
                          out.allele_frequencies.view(n_int, nb))
    idx = None
    for _ in range(3):                                   # the last build is the measurement (first: pool warm-up)
        if idx is not None:
            idx.free()
        sync()
        pool0 = _lib.pool_stats()
        t = time.perf_counter()
        idx = DeviceIndex.build(bnd, modulo)
        sync()
        dt = time.perf_counter() - t
        pool1 = _lib.pool_stats()
    kb = int(modulo - 1).bit_length()
    # bytes of the build AS IMPLEMENTED (csrc/gki_index_rows.hip), per record: first histogram 8 R (key from the k-mer, not
    # stored); first partition pass 24 R + 28 W (columns -> row + key), second 4 R (histogram) + 28 R + 28 W; group bounds
    # 4 R; finish 28 R + 26 W (four columns + frequency); plus the directory itself, 2 x 4 B x modulo, streamed once
    per_record = 8 + 52 + 60 + 4 + 54
    moved = per_record * nb + 8 * modulo
    n_kmers_host = idx.n_kmers.to_host()
    nonempty = int(np.count_nonzero(n_kmers_host))
    del n_kmers_host
    index_build = {"records": int(nb), "ms": 1e3 * dt, "records_per_s": nb / dt, "modulo": modulo, "key_bits": kb,
                   "form": "row-carrying: 2 stable partition passes (10 + 9 bits) + in-LDS finish on 10 bits",
                   "frequencies": True, "bytes_moved_model": int(moved), "bytes_per_record_model": per_record,
                   "achieved_GBps": moved / dt / 1e9, "frac_of_hbm_peak": moved / dt / 1e9 / HBM_PEAK_GBS,
                   "roofline": dict(index_roofline(nb, modulo, nonempty, dt, hashing=False),
                                    **dict(zip(("traffic", "traffic_source"), pmc_index_traffic("sparse", nb)))),
                   "timed": "wall clock around DeviceIndex.build incl. its allocations, device synchronised",
                   "ms_in_device_allocator": (pool1[2] - pool0[2]) + (pool1[3] - pool0[3]),
                   "cpu_baseline": cpu2.get("index_build")}
    # scalar CollisionFreeKmerIndex.get (:303-315) on the whole variant index: the class answers one k-mer from its own
    # host arrays (the device serves the batched getters); half the queries hit, as in BASELINE.md section 2's probe
    from graph_kmer_index_amd import CollisionFreeKmerIndex
    t = time.perf_counter()
    host_index = CollisionFreeKmerIndex(idx.hashes_to_index.to_host(), idx.n_kmers.to_host(), idx.nodes.to_host(nb),
                                        idx.ref_offsets.to_host(nb), idx.kmers.to_host(nb), modulo, idx.frequencies.to_host(nb),
                                        idx.allele_frequencies.to_host(nb))
    t_host = time.perf_counter() - t
    rng = np.random.default_rng(5)
    some = np.concatenate([host_index._kmers[rng.integers(0, nb, size=10000)], rng.integers(0, 4 ** k, size=10000, dtype=np.uint64)])
    rng.shuffle(some)
    some = [int(x) for x in some]
    n_found = 0
    t = time.perf_counter()
    for x in some:
        n_found += host_index.get(x)[0] is not None
    index_build["scalar_get_calls_per_s"] = len(some) / (time.perf_counter() - t)
    index_build["scalar_get"] = {"calls": len(some), "found": int(n_found), "index_to_host_s": t_host,
                                 "what": "CollisionFreeKmerIndex.get on the object's host arrays (all %d records), one Python call per k-mer" % nb}
    t = time.perf_counter()
    for x in some[:2000]:
        idx.get_small([x], 10)
    index_build["scalar_get_device_calls_per_s"] = 2000 / (time.perf_counter() - t)     # DeviceIndex.get_small: one launch + one sync
    del host_index
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

    # ---- read mapping at BASELINE configs[4]'s size: the reads are simulated on the device (gki_simulate_reads)
    hap = synthetic_haplotype_sequence(g)
    d_hap = _lib.DeviceArray.from_host(hap)
    t = time.perf_counter()
    d_letters = _lib.DeviceArray(n_reads * 150, np.uint8)
    _lib.check(lib.gki_simulate_reads(d_hap.ptr, len(hap), n_reads, 150, 99, 0.01, 0.1, 0, d_letters.ptr))
    t_reads = time.perf_counter() - t
    d_hap.free()
    d_start = _lib.DeviceArray.from_host(np.arange(n_reads + 1, dtype=np.int64) * 150)
    table = idx.probe_table()
    counts = _lib.DeviceArray(g.n_nodes, np.uint32)
    nk, nh = C.c_int64(0), C.c_int64(0)
    for _ in range(2):                                   # the last launch is the measurement
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
                    "reads_from": "gki_simulate_reads on the device (%.2f s for %.1f GB of letters): 90 %% from a random path of "
                                  "the graph, either strand, 1 %% substitutions; 10 %% uniform random; seed 99" % (t_reads, n_reads * 150 / 1e9),
                    "cpu_baseline": cpu2.get("read_mapping")}
    log("read mapping: %d reads, %.3g k-mers/s; random-request rate %.3g/s" % (n_reads, nk.value / dt_map, rate.value))
    for b in (d_letters, d_start, counts):
        b.free()
    idx.free()
    early = early_stop_record(lib, _lib, g, k, finder)
    early["cpu_baseline"] = cpu2.get("early_stop_search")

    # ---- every record of the step indexed: the reference's int32 directory (collision_free_kmer_index.py:453) stops at
    # 2^31 records, so the full set goes through bucket-range slices, here all on this one GPU
    full_index = None
    if full:
        try:
            full_index = full_index_record(lib, _lib, g, k, cp, out, max_variant_nodes, modulo)
        except _lib.GkiError as e:                          # e.g. not enough HBM beside a caller's other allocations
            full_index = {"skipped": str(e)}
    return index_build, read_mapping, early, full_index


def full_index_record(lib, _lib, g, k, cp, out, max_variant_nodes, modulo, n_slices=8):
    """Every record of the graph hashed AND indexed on this one GPU -- BASELINE's metric in its own words.  One int32
    directory stops at 2^31 records (collision_free_kmer_index.py:453), so the index is built in bucket-range slices
    (the class is collision_free_kmer_index.PartitionedDeviceIndex): the whole graph is enumerated and
    hashed (gki_finder_count + gki_finder_emit_flat, the step of the headline), its 3.16e9 records are partitioned by
    owning slice in one call (gki_partition_by_bucket_range, no limit on the number of records: two passes of < 2^31 rows
    laid out behind each other), and every slice is built with frequencies (gki_index_build_range).  On one GPU there is
    nothing to exchange: round 3 timed a device copy per slice as a stand-in for the all-to-all of configs[3]; that
    exchange is `sharded_build`'s subject (N > 1), not this record's."""
    from graph_kmer_index_amd.flat_kmers import DeviceFlatKmers
    from graph_kmer_index_amd.collision_free_kmer_index import (PartitionedDeviceIndex, DeviceRows, partition_rows_by_bucket_range,
                                                                partition_by_bucket_range)
    def sync():
        _lib.check(lib.gki_device_synchronize())
    n = out.n
    MASK = (1 << 64) - 1
    cols = ("hashes", "nodes", "ref_offsets", "allele_frequencies")
    W = n_slices
    G = FULL_INDEX_GROUP_BITS              # > 0: the grouped flow (rows between partition and build)
    kw = dict(only_save_one_node_per_kmer=True, max_variant_nodes=max_variant_nodes)
    from graph_kmer_index_amd import DenseKmerFinder
    finder = DenseKmerFinder(g, k, critical_graph_paths=cp, **kw)
    finder._params()
    if finder._count(layout=1) != n:
        raise _lib.GkiError(2, "full_index: the graph holds %d records, the step wrote %d" % (finder._count(layout=1), n))
    ROWS = FULL_INDEX_ROWS or G > 0        # the partitioned records stay 24-byte rows + keys between the partition and the slice builds
    parts = DeviceRows(n) if ROWS else DeviceFlatKmers.allocate(n)      # the partitioned records beside the step's own columns
    # everything once untimed: sizes the library's memory pool (hipMalloc / hipFree of tens of GB cost seconds on this stack,
    # DESIGN.md section 6 "Device memory pool"; the index_build record measures its third build for the same reason)
    do_partition = (lambda: partition_rows_by_bucket_range(out, modulo, W, group_bits=G, out=parts)) if ROWS else \
        (lambda: partition_by_bucket_range(out, modulo, W, out=parts))
    _, start = do_partition()
    biggest = max(range(W), key=lambda p: start[(p + 1) << G] - start[p << G])
    sl = PartitionedDeviceIndex.build_slice(parts, start, modulo, W, biggest, G)
    sl.free()
    out = finder.find_flat_on_device(out)        # (the find once untimed too: the timed one below then follows device work, not a pause)
    finder.synchronize()
    sync()
    pool0 = _lib.pool_stats()
    t = time.perf_counter()
    out = finder.find_flat_on_device(out)
    finder.synchronize()
    t_find = time.perf_counter() - t
    want = [getattr(out, c).checksum(n) for c in cols]
    sync()
    t = time.perf_counter()
    _, start = do_partition()
    sync()
    t_part = time.perf_counter() - t
    got = [(0, 0)] * 4
    sizes, build_ms, directories = [], [], []
    t_build = 0.0
    for p in range(W):
        sync()
        t = time.perf_counter()
        sl = PartitionedDeviceIndex.build_slice(parts, start, modulo, W, p, G)
        sync()
        build_ms.append(1e3 * (time.perf_counter() - t))
        t_build += time.perf_counter() - t
        # between two slice builds only device work (the checksum kernels): a host-side pause here lets the clocks drop and
        # the next build's first kernels pay for it (1.4 ms per slice when the non-empty buckets were counted on the host
        # inside this loop).  The directories (226 MB each) are kept and counted after the last build.
        for i, colname in enumerate(("kmers", "nodes", "ref_offsets", "allele_frequencies")):
            s_, x_ = getattr(sl, colname).checksum(sl.n)
            got[i] = ((got[i][0] + s_) & MASK, got[i][1] ^ x_)
        sizes.append(sl.n)
        directories.append(sl.n_kmers)
        sl.n_kmers = _lib.DeviceArray(1, np.uint32)
        sl.free()                    # checksummed and released before the next slice is built
    nonempty = 0
    for nk in directories:
        nonempty += int(np.count_nonzero(nk.to_host()))
        nk.free()
    pool1 = _lib.pool_stats()
    parts.free()
    total = sum(sizes)
    dt = t_find + t_part + t_build
    # as implemented, per record: the step 25 B; partition 8 R (histogram) + 24 R + 28 W (rows + keys; 24 W as columns); slice
    # build: per pass 4 R (histogram; 8 R from k-mers) + 28 R + 28 W, two passes (one when grouped), group bounds 4 R, finish 28 R + 26 W
    moved = (BYTES_PER_RECORD + (60 if ROWS else 56) + (118 if G else 178)) * total + 8 * modulo
    rec = {"records": int(total), "slices": W, "records_per_slice": sizes, "exceeds_int32_directory": bool(total >= 2 ** 31),
           "find_ms": 1e3 * t_find, "partition_ms": 1e3 * t_part, "build_slices_ms": 1e3 * t_build,
           "build_ms_per_slice": [round(x, 2) for x in build_ms], "ms": 1e3 * dt, "records_per_s": total / dt,
           "roofline": index_roofline(total, modulo, nonempty, dt, hashing=True),
           "roofline_at_the_passes_run": index_roofline(total, modulo, nonempty, dt, hashing=True, passes=2 if G else 3),
           "group_bits": G, "partitioned_records_as": "rows + keys" if ROWS else "four columns",
           "bytes_moved_model": int(moved), "achieved_GBps_as_implemented": moved / dt / 1e9,
           "frac_of_hbm_peak_as_implemented": moved / dt / 1e9 / HBM_PEAK_GBS,
           "device_allocator": {"hipMalloc_calls": pool1[0] - pool0[0], "hipFree_calls": pool1[1] - pool0[1],
                                "ms_in_hipMalloc": pool1[2] - pool0[2], "ms_in_hipFree": pool1[3] - pool0[3],
                                "note": "inside the timed phases, after one untimed partition and one untimed slice build sized "
                                        "the library's pool: what still reaches hipMalloc / hipFree"},
           "equals_step_output": bool(total == n),
           "payload_equals_flat_multiset": [tuple(w) for w in want] == [tuple(x) for x in got],
           "timed": "wall clock, device synchronised per phase: the whole graph enumerated and hashed (gki_finder_count + "
                    "gki_finder_emit_flat), its records partitioned by bucket range into %d slices (gki_partition_by_bucket_range, one "
                    "call; group_bits %d > 0: grouped and left as rows, gki_partition_rows_by_bucket_range), every slice built with "
                    "frequencies (gki_index_build_range / _from_rows); hashed AND indexed, end to end; checksums of the slices outside "
                    "the timed phases" % (W, G)}
    log("full index: %d records in %d slices: find %.1f + partition %.1f + build %.1f = %.1f ms" % (total, W, 1e3 * t_find, 1e3 * t_part, 1e3 * t_build, 1e3 * dt))
    return rec


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
           "form": "all-nodes mode: the count pass walks on 32-byte node records and writes every finished k-mer down (<= 4 entries "
                   "of 32-48 B per start position, piece-major), the emit pass expands them one lane per record; start positions "
                   "that do not fit are listed and walked again",
           "timed": "wall clock around gki_forward_count + gki_forward_emit, graph / start arrays / output columns in HBM"}
    log("early-stop search: %d start positions, %d records in %.2f ms" % (n_pos, n_rec, 1e3 * dt))
    for b in bufs + [d_nodes, d_offs, d_start]:
        b.free()
    return rec


# ------------------------------------------------------------------------------------------ N > 1: exchange + build
def sharded_build_record(lib, _lib, plane, finder, out, modulo, same_device_ranks):
    """BASELINE configs[3] with ranks: what follows the enumeration in the reference's CLI (gather the chunks,
    command_line_interface.py:607-614; build the table, :156-174 / collision_free_kmer_index.py:423-467).
    (A) all-gather(v) of every rank's variant-index records, then the CollisionFreeKmerIndex build on every rank;
    (B) ALL records: partition by bucket range -> all-to-all(v) -> every rank builds its slice of the directory.
    Phase times are the slowest rank's."""
    from graph_kmer_index_amd.flat_kmers import DeviceFlatKmers
    from graph_kmer_index_amd.collision_free_kmer_index import DeviceIndex, bucket_range, partition_by_bucket_range
    from graph_kmer_index_amd.parallel import make_comm
    rank, world = plane.rank, plane.world
    MASK = (1 << 64) - 1

    def sync():
        _lib.check(lib.gki_device_synchronize())

    def slowest(x):
        return max(plane.allgather_float(x))

    comm = make_comm(plane)
    info = comm.info()
    rec = dict(info)
    rec["ranks"] = world
    # which GPU every rank really sits on (a LOCAL_RANK % n_devices that maps two ranks to one card shows here), and what
    # RCCL says about the communicator on every rank
    from graph_kmer_index_amd.parallel import device_identity
    rec["device_of_rank"] = [b.decode() for b in plane._allgather_bytes(device_identity().encode())]
    rec["rccl_ranks_seen_by_rank"] = plane.allgather_int(int(info.get("rccl_ranks") or 0))
    n_int = finder.interior_records()
    nb = out.n - n_int
    bnd = DeviceFlatKmers(nb, out.hashes.view(n_int, nb), out.nodes.view(n_int, nb), out.ref_offsets.view(n_int, nb),
                          out.allele_frequencies.view(n_int, nb))
    # ---- (A) all-gather + build; twice, the second round is the measurement (the first sizes the memory pool: a cold
    # hipMalloc of GBs costs more than the exchange, and bench.py's other records measure their warm repeat too)
    for _ in range(2):
        plane.barrier()
        sync()
        t = time.perf_counter()
        everything, counts = comm.allgather_flat(bnd)
        sync()
        t_gather = slowest(time.perf_counter() - t)
        plane.barrier()
        t = time.perf_counter()
        idx = DeviceIndex.build(everything, modulo)
        sync()
        t_build = slowest(time.perf_counter() - t)
        total_variant = int(sum(counts))
        ok_count = idx.n == total_variant
        idx.free()
        everything.free()
    link_bytes = max(counts) * 24                     # what the largest shard puts on each of its links (one copy per peer)
    rec["variant_index_allgather"] = {
        "records_total": total_variant, "records_per_rank": [int(c) for c in counts], "allgather_ms": 1e3 * t_gather,
        "build_ms": 1e3 * t_build, "bytes_per_link": int(link_bytes), "bytes_received_per_rank": int((total_variant - min(counts)) * 24),
        "GBps_per_link": link_bytes / t_gather / 1e9 if t_gather > 0 else None,
        "frac_of_xgmi_link": link_bytes / t_gather / 1e9 / XGMI_LINK_GBS if t_gather > 0 else None,
        "index_holds_every_record": bool(ok_count),
        "what": ("gki_comm_allgather_flat (counts over the control plane, then one ncclSend/ncclRecv pair per peer and column "
                 "in a group)" if info["exchange"] == "rccl" else
                 "SharedDeviceComm.allgather_flat (HIP IPC handles over the control plane, device-to-device copies; every rank's "
                 "work shares ONE device here, the link figures are not xGMI figures)")
                + " + gki_index_build of the gathered records on every rank; second of two rounds"}
    # ---- (B) bucket-range partitioned build of ALL records
    free_b, total_b = _mem_info(lib, _lib)
    need = (out.n * 24 * 2 + out.n * 64) * same_device_ranks      # partitioned + received copies, build temporaries, per rank on this device
    if out.n >= (1 << 31):
        rec["full_index_partitioned"] = {"skipped": "a rank's shard holds %d records: partition at most 2^31-1 at a time" % out.n}
    elif min(plane.allgather_int(int(need <= free_b))) == 0:
        rec["full_index_partitioned"] = {"skipped": "%d ranks share this device: %.0f GB needed, %.0f GB free" % (same_device_ranks, need / 1e9, free_b / 1e9)}
    else:
        want = [getattr(out, c).checksum(out.n) for c in ("hashes", "nodes", "ref_offsets", "allele_frequencies")]
        plane.barrier()
        sync()
        t = time.perf_counter()
        by_dest, send_start = partition_by_bucket_range(out, modulo, world)
        sync()
        t_part = slowest(time.perf_counter() - t)
        plane.barrier()
        t = time.perf_counter()
        received, recv_start = comm.alltoall_flat(by_dest, send_start)
        sync()
        t_a2a = slowest(time.perf_counter() - t)
        by_dest.free()
        lo, hi = bucket_range(modulo, world, rank)
        plane.barrier()
        t = time.perf_counter()
        sl = DeviceIndex.build(received, modulo, bucket_begin=lo, n_buckets=hi - lo)
        sync()
        t_slice = slowest(time.perf_counter() - t)
        received.free()
        got = [getattr(sl, c).checksum(sl.n) for c in ("kmers", "nodes", "ref_offsets", "allele_frequencies")]
        flat = plane.allgather_ints([v for pair in want for v in (pair[0] & MASK, pair[1])] + [out.n])
        mine = plane.allgather_ints([v for pair in got for v in (pair[0] & MASK, pair[1])] + [sl.n])
        def fold(rows):
            sums = [0] * 4
            xors = [0] * 4
            for row in rows:
                for i in range(4):
                    sums[i] = (sums[i] + (row[2 * i] & MASK)) & MASK
                    xors[i] ^= row[2 * i + 1] & MASK
            return sums, xors
        send = [send_start[r + 1] - send_start[r] for r in range(world)]
        link = max(send[r] for r in range(world) if r != rank) * 24 if world > 1 else 0
        link = max(plane.allgather_int(link))
        rec["full_index_partitioned"] = {
            "records_total": int(sum(row[8] for row in flat)), "records_per_slice": [int(row[8]) for row in mine],
            "partition_ms": 1e3 * t_part, "alltoall_ms": 1e3 * t_a2a, "slice_build_ms": 1e3 * t_slice,
            "ms": 1e3 * (t_part + t_a2a + t_slice), "bytes_per_link_max": int(link),
            "GBps_per_link": link / t_a2a / 1e9 if t_a2a > 0 else None,
            "frac_of_xgmi_link": link / t_a2a / 1e9 / XGMI_LINK_GBS if t_a2a > 0 else None,
            "payload_equals_flat_multiset": fold([[v & MASK for v in row[:8]] for row in flat]) == fold([[v & MASK for v in row[:8]] for row in mine]),
            "slices_hold_every_record": int(sum(row[8] for row in flat)) == int(sum(row[8] for row in mine)),
            "what": "gki_partition_by_bucket_range -> " + ("gki_comm_alltoall_flat (one ncclSend/ncclRecv pair per peer and column)"
                                                            if info["exchange"] == "rccl" else "SharedDeviceComm.alltoall_flat (HIP IPC)")
                    + " -> gki_index_build_range of the rank's 1/world of the directory, frequencies on; one round, cold allocations "
                      "included (the columns of a second round would not fit beside the first on every size)"}
        sl.free()
    comm.close()
    return rec


def _mem_info(lib, _lib):
    import ctypes as C
    f, t = C.c_int64(0), C.c_int64(0)
    _lib.check(lib.gki_mem_info(C.byref(f), C.byref(t)))
    return f.value, t.value


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
    ap.add_argument("--reads", type=float, default=1e8, help="reads of the read_mapping record, BASELINE configs[4]: 1e8 "
                    "(0: skip the secondary records)")
    ap.add_argument("--no-full-index", action="store_true", help="skip the full_index record (N=1)")
    ap.add_argument("--full-index-columns", action="store_true",
                    help="full_index with the partitioned records as four columns (gki_partition_by_bucket_range -> gki_index_build_range) "
                         "instead of rows + keys (gki_partition_rows_by_bucket_range -> gki_index_build_range_from_rows)")
    ap.add_argument("--full-index-group-bits", type=int, default=7,
                    help="full_index: the partition also groups every slice by this many top key bits (8 slices x 2^7 = the pass's 1024 "
                         "digits) and the slice builds start from there with one pass less; 0: plain bucket-range partition")
    ap.add_argument("--no-sharded-build", action="store_true", help="skip the sharded_build record (N>1)")
    ap.add_argument("--sharded-build-budget", type=float, default=300.0, help="seconds the sharded_build record may take before "
                    "rank 0 prints the line without it")
    ap.add_argument("--modulo", type=int, default=452930477)
    args = ap.parse_args()
    global FULL_INDEX_GROUP_BITS, FULL_INDEX_ROWS
    FULL_INDEX_GROUP_BITS = 0 if args.full_index_columns else args.full_index_group_bits      # (the grouping exists for rows only)
    FULL_INDEX_ROWS = not args.full_index_columns
    if args.nested > 0:
        args.max_variant_nodes = max(args.max_variant_nodes, 8)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.pretend_shard:
        launch_ranks(args)                                # does not return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("WORLD_SIZE=%d but --gpus=%d; using WORLD_SIZE" % (world, args.gpus))
    plain = not (args.linear or args.indels or args.nested or args.all_nodes or args.pretend_shard or args.general)
    cpu = cpu2 = None
    if world == 1 and not args.no_cpu_baseline:       # N=1 only, and before anything touches the GPU
        cores = args.cpu_cores or min(16, len(os.sched_getaffinity(0)))
        cpu = cpu_baseline(int(args.cpu_sample_bases), args.k, args.max_variant_nodes, cores)
        log("cpu baseline: %.3g k-mers/s on %d cores" % (cpu["value"], cpu["cores"]))
        if args.reads > 0 and plain:
            cpu2 = cpu_baselines_secondary(args.k)
            log("cpu baselines (1 core): index build %.3g records/s, read mapping %.3g k-mers/s, early-stop search %.3g starts/s"
                % (cpu2["index_build"]["value"], cpu2["read_mapping"]["value"], cpu2["early_stop_search"]["value"]))
    from graph_kmer_index_amd.parallel import SocketControlPlane
    plane = SocketControlPlane(rank, world)           # barrier + max / sum of two scalars; a no-op at world 1

    g, t_gen, t_map, cleanup_dir = node_graph(args, plane, rank, local_rank)
    adapter = adapter_times(g, args) if rank == 0 and world == 1 else None

    from graph_kmer_index_amd import _lib, DenseKmerFinder, CriticalGraphPaths, DeviceGraph
    from graph_kmer_index_amd.sharding import shard_range
    from graph_kmer_index_amd.parallel import device_identity
    lib = _lib.load()
    _lib.require_device()
    n_dev = _lib.device_count()
    _lib.check(lib.gki_set_device(local_rank % n_dev))
    ids = plane._allgather_bytes(device_identity().encode())
    same_device_ranks = sum(1 for x in ids if x == ids[rank])

    G, S, k = int(args.bases), int(args.sites), args.k
    t0 = time.perf_counter()
    dg = DeviceGraph.of(g)                           # upload + device-side preparation; cached on the graph object
    t_up = time.perf_counter() - t0
    t0 = time.perf_counter()
    cp = CriticalGraphPaths.from_graph(g, k)         # on the device, over the resident graph (csrc/gki_critical.hip)
    t_crit = time.perf_counter() - t0
    if rank == 0:
        log("graph: %d nodes, %d bases (+%d alt), %d critical points; generate %.1fs (once per node), critical paths %.2fs, "
            "upload+prepare %.2fs" % (g.n_nodes, G, len(g.seq) - G, len(cp), t_gen, t_crit, t_up))
    shard_r, shard_w = rank, world
    if args.pretend_shard:
        shard_r, shard_w = (int(x) for x in args.pretend_shard.split("/"))
    a, b = shard_range(g, cp, shard_r, shard_w)
    finder = DenseKmerFinder(g, k, critical_graph_paths=cp, only_save_one_node_per_kmer=not args.all_nodes,
                             max_variant_nodes=args.max_variant_nodes,
                             start_at_critical_path_number=a if shard_w > 1 else None,
                             stop_at_critical_path_number=b if shard_w > 1 else None)

    finder._force_general_kernels = args.general
    t0 = time.perf_counter()
    finder._params()                                 # node classification (device) + run parameters, once per finder
    t_cls = time.perf_counter() - t0

    def barrier():
        _lib.check(lib.gki_device_synchronize())
        plane.barrier()

    out = None
    interior_ms = []
    for _ in range(args.warmup):
        out = finder.find_flat_on_device(out)
        finder.synchronize()
    # the box's own store ceiling for the four-column pattern, into the output columns the warm-up just sized
    ceiling = None
    if out is not None and out.n >= (1 << 20):
        import ctypes as C
        bw = C.c_double(0.0)
        _lib.check(lib.gki_measure_store_bw(out.hashes.ptr, out.nodes.ptr, out.ref_offsets.ptr, out.allele_frequencies.ptr,
                                            out.n, C.byref(bw)))
        ceiling = bw.value / 1e9
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
    n_per_rank = plane.allgather_int(n_local)
    n_total = sum(n_per_rank)
    ceilings = plane.allgather_float(ceiling or 0.0)

    checks = None
    if args.verify:
        checks = verify(out, g, k, n_interior)
    secondary = None
    if world == 1 and args.reads > 0 and plain and out.n - n_interior > 0:
        secondary = secondary_records(lib, _lib, g, k, cp, finder, out, int(args.reads), cpu2, args.max_variant_nodes,
                                      not args.no_full_index, args.modulo)
    res = None
    if rank == 0:
        ms_step = 1000.0 * elapsed / args.steps
        value = n_total * args.steps / elapsed
        avg_int_ms = float(np.mean(interior_ms))
        achieved = BYTES_PER_RECORD * n_interior / (avg_int_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(G, S, k) if world == 1 and plain else (None, None)
        res = {
            "metric": "k-mers hashed+indexed per second (k=31, 3 Gbp graph)", "value": value, "unit": "k-mers/s",
            "timed_region": "enumerate + hash + FlatKmers rows in HBM (gki_finder_count + gki_finder_emit_flat); the index "
                            "build, the exchange with ranks and the read side are the separate records below",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[%d]: synthetic %.3g bp linear-ref obgraph + %.3g SNP bubbles, k=%d, "
                                   "DenseKmerFinder -> FlatKmers (hash u64, node u32, ref_offset u64, af f32) in HBM%s"
                                   % (2 if world == 1 else 3, G, S, k, "" if world == 1 else ", sharded over %d ranks" % world),
                       "n_ref_bases": G, "n_snp_bubbles": int(S), "k": k, "max_variant_nodes": args.max_variant_nodes,
                       "only_save_one_node_per_kmer": not args.all_nodes, "records_per_step": n_total, "n_nodes": int(g.n_nodes),
                       "records_per_rank": n_per_rank, "ranks_sharing_rank0_device": same_device_ranks,
                       "sharding": "critical-path ranges balanced by bases, whole graph resident on every GPU"},
            "roofline": {"bound": "hbm", "kernel": "k_emit_interior_runs", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "bytes_per_record": BYTES_PER_RECORD, "records_per_launch": int(n_interior),
                         "avg_launch_ms": avg_int_ms,
                         "ceiling_measured": ceiling, "frac_of_ceiling": (achieved / ceiling) if ceiling else None,
                         "ceiling_what": "gki_measure_store_bw on this device before the timed region: the same four-column "
                                         "store pattern alone in a kernel, 24 B per record, best of two launches"
                                         + ("; per rank: %s" % ["%.0f" % c for c in ceilings] if world > 1 else "")},
            "kernels_ms_rank0_last_step": kern,
            "setup_s": {"generate_graph_host_once_per_node": t_gen, "map_shared_graph": t_map, "critical_paths_device": t_crit, "classify_nodes_and_run_parameters": t_cls,
                        "upload_and_prepare": t_up, "from_obgraph_adapter": adapter},
        }
        if checks is not None:
            res["verify"] = checks
        if secondary is not None:
            res["index_build"], res["read_mapping"], res["early_stop_search"], res["full_index"] = secondary
            fi = res["full_index"]
            if fi and "records_per_s" in fi:
                # the metric's own words: every record of the graph hashed AND indexed, end to end (`value` above times
                # enumeration + hashing; the index build of all records is the full_index record)
                res["value_hashed_and_indexed"] = fi["records_per_s"]
                res["hashed_and_indexed_roofline_frac"] = fi["roofline"]["frac"]
        res["cpu_baseline"] = cpu
        res["cpu_baselines_note"] = ("`cpu_baseline` (beside `value`) is the oracle on ALL host cores (`cores` in the object); the "
                                     "`cpu_baseline` objects inside index_build / read_mapping / early_stop_search are the oracle on "
                                     "ONE core: compare ratios across records only after scaling by `cores`")
    if world > 1 and plain and not args.no_sharded_build:
        # The exchange has never run between real GPUs (one GPU per builder box): if a rank fails or a collective never
        # returns, the headline line must still come out.  A watchdog prints it (rank 0) and ends the rank.
        import threading

        def give_up():
            # a hang is a FINDING, not a pass: the line says so at its top level (`incomplete`), the headline numbers above
            # it stand (they were measured before the exchange started), and the exit code is the launcher's to report:
            # rank 0 leaves with 0 only so that its line is not discarded with it, every other rank with 3
            if rank == 0:
                res["sharded_build"] = {"timed_out_after_s": args.sharded_build_budget, "failed": "no answer from the exchange "
                                        "or the build within the budget; stderr has rank 0's '[gki comm]' phase lines"}
                res["incomplete"] = ["sharded_build"]
                print(json.dumps(res), flush=True)
            os._exit(0 if rank == 0 else 3)
        dog = threading.Timer(args.sharded_build_budget, give_up)
        dog.daemon = True
        dog.start()
        try:
            sharded = sharded_build_record(lib, _lib, plane, finder, out, args.modulo, same_device_ranks)
        except Exception as e:                        # noqa: BLE001 -- reported in the line, the step's numbers stand
            sharded = {"failed": "%s: %s" % (type(e).__name__, e)}
            log("rank %d: sharded_build failed: %s" % (rank, sharded["failed"]))
        dog.cancel()
        if rank == 0:
            res["sharded_build"] = sharded
    if rank == 0:
        print(json.dumps(res), flush=True)
    try:
        plane.barrier()
    except (OSError, ConnectionError):                # a rank that gave up above: the line is out, nothing left to agree on
        pass
    plane.close()
    if cleanup_dir:
        shutil.rmtree(cleanup_dir, ignore_errors=True)


def verify(out, g, k, n_interior):
    """Size-independent properties of the full-size output (tests/ compare small sizes with the oracle).
    Layout: n_interior records whose window lies inside one node (by position), then the boundary records (by
    end node)."""
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
