"""world_size-2 tests on CPU (gloo): the sharding of the enumeration and the bookkeeping of the exchange.
The HIP kernels cannot run here, so each rank's shard is computed by the oracle (checker standing in for the
device in this test only); what is under test is graph_kmer_index_amd.sharding / parallel's control plane."""
import os
import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from graph_kmer_index_amd.graph import synthetic_snp_graph
from graph_kmer_index_amd.sharding import critical_path_cuts, shard_range
from graph_kmer_index_amd import CriticalGraphPaths
from oracle import oracle


def _worker(rank, world, port, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from graph_kmer_index_amd.parallel import TorchControlPlane
    cp_plane = TorchControlPlane()
    g = synthetic_snp_graph(60000, 700, k=31, seed=21)
    cp = CriticalGraphPaths.from_graph(g, 31)
    a, b = shard_range(g, cp, rank, world)
    mine = oracle.find(g, 31, (cp.nodes, cp.offsets), True, 5, start_at_critical_path_number=a,
                       stop_at_critical_path_number=b)
    counts = cp_plane.allgather_int(len(mine["kmers"]))
    ident = cp_plane.broadcast_bytes(bytes(range(128)) if rank == 0 else None, 0)
    # all-to-all bookkeeping of the bucket-range partitioned build: row r = what rank r sends to each rank
    modulo = 100003
    bucket = mine["kmers"].astype(np.uint64) % np.uint64(modulo)
    begins = np.array([modulo * p // world for p in range(world)], dtype=np.uint64)
    send = np.bincount(np.searchsorted(begins, bucket, side="right") - 1, minlength=world)
    matrix = cp_plane.allgather_ints(send.tolist())
    np.savez(os.path.join(tmpdir, "r%d.npz" % rank), counts=np.array(counts), ident=np.frombuffer(ident, np.uint8),
             send=send, matrix=np.array(matrix), **mine)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_control_plane(tmp_path):
    world, port = 2, 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(str(tmp_path / ("r%d.npz" % r))) for r in range(world)]
    g = synthetic_snp_graph(60000, 700, k=31, seed=21)
    full = oracle.find(g, 31, None, True, 5)
    for key in ("kmers", "nodes", "start_nodes", "start_offsets"):
        assert np.array_equal(np.concatenate([p[key] for p in parts]), full[key])      # rank order == full run
    for p in parts:
        assert p["counts"].tolist() == [len(q["kmers"]) for q in parts]
        assert p["ident"].tolist() == list(range(128))
    # the counts matrix every rank holds: row r = rank r's send counts; its column r = what rank r receives
    for p in parts:
        assert np.array_equal(p["matrix"], np.stack([q["send"] for q in parts]))
    assert parts[0]["matrix"].sum() == len(full["kmers"])
    # shards are balanced by bases
    assert abs(len(parts[0]["kmers"]) - len(parts[1]["kmers"])) < 0.1 * len(full["kmers"])


def test_cuts_cover_all_critical_points():
    g = synthetic_snp_graph(200000, 2500, k=31, seed=3)
    cp = CriticalGraphPaths.from_graph(g, 31)
    for w in (1, 2, 3, 8, 64):
        cuts = critical_path_cuts(g, cp, w)
        assert cuts[0] == 0 and cuts[-1] == len(cp) and len(cuts) == w + 1
        assert all(x <= y for x, y in zip(cuts[:-1], cuts[1:]))
        total = 0
        for r in range(w):
            a, b = shard_range(g, cp, r, w)
            total += len(oracle.find(g, 31, (cp.nodes, cp.offsets), True, 5, start_at_critical_path_number=a,
                                     stop_at_critical_path_number=b)["kmers"]) if w <= 3 else 0
        if w <= 3:
            assert total == len(oracle.find(g, 31, (cp.nodes, cp.offsets), True, 5)["kmers"])


def test_shards_partition_small_graphs_with_few_or_offset0_critical_points():
    # found by tools/soak_parity.py: no critical point at all, a first critical point far into the graph (several cuts
    # at 0) and critical points at offset 0 (chunks that overlap in the reference) must not break the partition
    from collections import Counter
    from graph_kmer_index_amd.graph import GraphArrays
    from graphgen import random_bubble_graph
    rng = np.random.default_rng(5)
    seen_no_crit = seen_offset0 = 0
    for it in range(300):
        k = int(rng.integers(2, 12))
        nv = int(rng.integers(1, 6))
        seqs, edges, lin, af = random_bubble_graph(rng, n_var=nv, min_ref=1, max_ref=int(rng.integers(2, 3 * k + 3)), p_indel=0.3,
                                                   chain_after={int(rng.integers(-1, nv)): int(rng.integers(1, k + 2))})
        g = GraphArrays.from_dicts(seqs, edges, lin, af)
        try:
            crit = oracle.critical_paths(g, k)
            full, flags = oracle.find(g, k, crit, True, 4, return_flags=True)
        except oracle.OracleError:
            continue
        if flags:
            continue
        cp = CriticalGraphPaths(crit[0], crit[1])
        seen_no_crit += len(cp) == 0
        seen_offset0 += bool(np.any(np.asarray(crit[1]) == 0))
        for world in (2, 3, 6):
            cuts = critical_path_cuts(g, cp, world)
            assert len(cuts) == world + 1 and cuts[0] == 0 and all(x <= y for x, y in zip(cuts[:-1], cuts[1:]))
            rows = Counter()
            for a, b in zip(cuts[:-1], cuts[1:]):
                o = oracle.find(g, k, crit, True, 4, start_at_critical_path_number=a, stop_at_critical_path_number=b)
                rows.update(zip(o["kmers"].tolist(), o["start_nodes"].tolist(), o["start_offsets"].tolist(), o["nodes"].tolist()))
            want = Counter(zip(full["kmers"].tolist(), full["start_nodes"].tolist(), full["start_offsets"].tolist(),
                               full["nodes"].tolist()))
            assert rows == want, (it, k, world, cuts)
    assert seen_no_crit > 0 and seen_offset0 > 0


def _socket_worker(rank, world, port, tmpdir):
    """Twin of _worker over parallel.SocketControlPlane (the product's control plane: plain TCP, no torch)."""
    from graph_kmer_index_amd.parallel import SocketControlPlane
    plane = SocketControlPlane(rank, world, "127.0.0.1", port)
    g = synthetic_snp_graph(60000, 700, k=31, seed=21)
    cp = CriticalGraphPaths.from_graph(g, 31)
    a, b = shard_range(g, cp, rank, world)
    mine = oracle.find(g, 31, (cp.nodes, cp.offsets), True, 5, start_at_critical_path_number=a,
                       stop_at_critical_path_number=b)
    counts = plane.allgather_int(len(mine["kmers"]))
    ident = plane.broadcast_bytes(bytes(range(128)) if rank == 0 else None, 0)
    modulo = 100003
    bucket = mine["kmers"].astype(np.uint64) % np.uint64(modulo)
    begins = np.array([modulo * p // world for p in range(world)], dtype=np.uint64)
    send = np.bincount(np.searchsorted(begins, bucket, side="right") - 1, minlength=world)
    matrix = plane.allgather_ints(send.tolist())
    slowest = max(plane.allgather_float(0.5 + rank))
    np.savez(os.path.join(tmpdir, "r%d.npz" % rank), counts=np.array(counts), ident=np.frombuffer(ident, np.uint8),
             send=send, matrix=np.array(matrix), slowest=slowest, n=len(mine["kmers"]))
    plane.barrier()
    plane.close()


def _intruders(port, world, stop):
    """What a shared host throws at rank 0's port while the real ranks join: a connection that says nothing, one that
    sends junk, a worker of another job (wrong token), a rank outside the world, and a second 'rank 1'."""
    import socket
    import struct
    import time
    from graph_kmer_index_amd.parallel import SocketControlPlane as P
    good = P.job_token("127.0.0.1", port, world)
    hellos = [b"GET / HTTP/1.0\r\n\r\n" + b"x" * 40, P.MAGIC + b"\0" * 16 + struct.pack("<qq", world, 1),
              P.MAGIC + good + struct.pack("<qq", world, world + 3), None, P.MAGIC + good + struct.pack("<qq", world, 1)]
    verdicts = []
    for h in hellos:
        deadline = time.time() + 20
        while time.time() < deadline and not stop.is_set():
            try:
                c = socket.create_connection(("127.0.0.1", port), timeout=1.0)
                if c.getsockname() == c.getpeername():       # TCP self-connection: rank 0 is not listening yet
                    c.close()
                    raise OSError("self-connection")
            except OSError:
                time.sleep(0.05)
                continue
            try:
                if h is not None:
                    c.sendall(h)
                c.settimeout(8.0)
                verdicts.append(c.recv(2))
            except OSError:
                verdicts.append(b"")
            c.close()
            break
    return verdicts


@pytest.mark.parametrize("world", [2, 3])
def test_socket_control_plane_ranks(tmp_path, world):
    """World 2 and 3 over the product's control plane, with intruders on the port (a junk client, another
    job's worker, an out-of-range rank, a silent connection must neither be adopted nor stop the job)."""
    import multiprocessing
    import threading
    port = 31000 + os.getpid() % 2000 + 20 * world
    ctx = multiprocessing.get_context("spawn")
    procs = [ctx.Process(target=_socket_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    stop, seen = threading.Event(), []
    intruder = threading.Thread(target=lambda: seen.extend(_intruders(port, world, stop)))
    procs[0].start()
    intruder.start()
    import time
    time.sleep(1.0)                          # the intruders get there first
    for p in procs[1:]:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    stop.set()
    intruder.join(30)
    assert len(seen) >= 3 and b"OK" not in seen[:4]       # junk, foreign token, rank out of range, silence: never adopted
    parts = [np.load(str(tmp_path / ("r%d.npz" % r))) for r in range(world)]
    g = synthetic_snp_graph(60000, 700, k=31, seed=21)
    n_full = len(oracle.find(g, 31, None, True, 5)["kmers"])
    for p in parts:
        assert p["counts"].tolist() == [int(q["n"]) for q in parts] and sum(p["counts"]) == n_full
        assert p["ident"].tolist() == list(range(128))
        assert np.array_equal(p["matrix"], np.stack([q["send"] for q in parts]))
        assert float(p["slowest"]) == world - 0.5


def test_socket_control_plane_reports_a_taken_port():
    import socket
    from graph_kmer_index_amd.parallel import SocketControlPlane
    port = 33500 + os.getpid() % 1000
    squatter = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    squatter.bind(("127.0.0.1", port))
    squatter.listen(1)
    try:
        with pytest.raises(OSError, match="GKI_CONTROL_PORT"):
            SocketControlPlane(0, 2, "127.0.0.1", port, timeout=2.0)
    finally:
        squatter.close()


def test_socket_control_plane_names_the_ranks_that_never_joined():
    from graph_kmer_index_amd.parallel import SocketControlPlane
    with pytest.raises(TimeoutError, match=r"ranks \[1, 2\]"):
        SocketControlPlane(0, 3, "127.0.0.1", 34600 + os.getpid() % 1000, timeout=1.5)
