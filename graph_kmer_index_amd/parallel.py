"""Multi-GPU index build: one process per GPU, enumeration sharded by critical-path ranges, one RCCL
all-gather(v) of the finished FlatKmers columns over xGMI, then the CollisionFreeKmerIndex build.

Mirrors the reference's CLI `index` + `make_from_flat` pair (command_line_interface.py:553-622, 156-174) where a
process pool runs `DenseKmerFinder` per chunk and the results are concatenated in chunk order.

The control plane (exchange of the RCCL id and of the per-rank record counts: 128 bytes and a few integers) is any
object with `rank`, `world`, `broadcast_bytes(b, src)`, `allgather_int(x)` and `allgather_ints(xs)`.
`SocketControlPlane` is the product's: plain TCP to rank 0 at MASTER_ADDR:MASTER_PORT, no third-party package.
`TorchControlPlane` is an optional adapter for callers that already run under torch.distributed.
`LoopbackWorld` runs all ranks of a build in ONE process on ONE GPU (threads; the exchanges become device copies): the
bucket-partitioned build on a single GPU for indexes past the reference's 2^31-record limit, and the way the
multi-rank logic is tested without a multi-GPU node."""
import ctypes as C
import os
import socket
import struct
import threading
import numpy as np

from . import _lib
from .flat_kmers import DeviceFlatKmers
from .sharding import shard_range


class _GatherControlPlane:
    """Everything a control plane offers, on top of one primitive: `_allgather_bytes(payload) -> [payload of rank 0,
    ..., payload of rank world-1]` on every rank."""
    rank = 0
    world = 1

    def _allgather_bytes(self, payload):
        raise NotImplementedError

    def broadcast_bytes(self, b, src=0):
        return self._allgather_bytes(b if self.rank == src and b is not None else b"")[src]

    def allgather_int(self, x):
        return [struct.unpack("<q", p)[0] for p in self._allgather_bytes(struct.pack("<q", int(x)))]

    def allgather_ints(self, xs):
        """Every rank's list of integers -> the matrix (row r = rank r's list)."""
        xs = [int(x) for x in xs]
        rows = self._allgather_bytes(struct.pack("<%dq" % len(xs), *xs))
        return [list(struct.unpack("<%dq" % (len(p) // 8), p)) for p in rows]

    def allgather_float(self, x):
        return [struct.unpack("<d", p)[0] for p in self._allgather_bytes(struct.pack("<d", float(x)))]

    def barrier(self):
        self._allgather_bytes(b"")


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("control plane: peer closed the connection")
        buf += chunk
    return bytes(buf)


def _send_msg(sock, payload):
    sock.sendall(struct.pack("<q", len(payload)) + payload)


def _recv_msg(sock):
    return _recv_exact(sock, struct.unpack("<q", _recv_exact(sock, 8))[0])


class SocketControlPlane(_GatherControlPlane):
    """Star over TCP: rank 0 listens on (addr, port), every other rank keeps one connection to it.  A gather is
    "everyone sends to rank 0, rank 0 sends the list back".  Defaults come from the launcher's environment
    (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT, as torch.distributed.run and mpirun wrappers export them)."""

    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=120.0):
        self.rank = int(os.environ.get("RANK", 0)) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", 1)) if world is None else int(world)
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        # one above the launcher's port: torch.distributed.run's own rendezvous store listens on MASTER_PORT itself
        port = int(port) if port is not None else int(os.environ.get("MASTER_PORT", 29400)) + 1
        self._peers, self._sock, self._server = [], None, None
        if self.world == 1:
            return
        # Rank 0 takes the first free port of port .. port+15 and greets every connection with a magic word; the others
        # try those ports in turn until one answers with it (another service may own a port of the range).
        magic = b"GKI1" + struct.pack("<q", self.world)
        ports = range(port, port + 16)
        if self.rank == 0:
            last = None
            for cand in ports:
                srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    srv.bind((addr, cand))
                    self._server = srv
                    break
                except OSError as e:
                    last = e
                    srv.close()
            if self._server is None:
                raise last
            self._server.listen(self.world)
            self._server.settimeout(timeout)
            peers = {}
            while len(peers) < self.world - 1:
                conn, _ = self._server.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(timeout)
                conn.sendall(magic)
                peers[struct.unpack("<q", _recv_exact(conn, 8))[0]] = conn
            self._peers = [peers[r] for r in range(1, self.world)]
        else:
            import time
            deadline = time.time() + timeout
            while self._sock is None:
                for cand in ports:
                    try:
                        sock = socket.create_connection((addr, cand), timeout=2.0)
                        sock.settimeout(2.0)
                        if _recv_exact(sock, len(magic)) == magic:
                            self._sock = sock
                            break
                        sock.close()
                    except (OSError, ConnectionError):
                        pass
                if self._sock is None:
                    if time.time() > deadline:
                        raise TimeoutError("control plane: rank 0 did not answer on %s:%d..%d" % (addr, port, port + 15))
                    time.sleep(0.05)
            self._sock.settimeout(timeout)
            self._sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            self._sock.sendall(struct.pack("<q", self.rank))

    def _allgather_bytes(self, payload):
        payload = bytes(payload)
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [_recv_msg(c) for c in self._peers]
            blob = b"".join(struct.pack("<q", len(p)) + p for p in parts)
            for c in self._peers:
                _send_msg(c, blob)
            return parts
        _send_msg(self._sock, payload)
        blob, parts, off = _recv_msg(self._sock), [], 0
        for _ in range(self.world):
            n = struct.unpack_from("<q", blob, off)[0]
            parts.append(blob[off + 8:off + 8 + n])
            off += 8 + n
        return parts

    def close(self):
        for c in self._peers + [x for x in (self._sock, self._server) if x is not None]:
            try:
                c.close()
            except OSError:
                pass
        self._peers, self._sock, self._server = [], None, None


class TorchControlPlane:
    """Optional adapter: the same interface over an initialised torch.distributed process group."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist, self._group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def broadcast_bytes(self, b, src=0):
        import torch
        t = torch.tensor(list(b), dtype=torch.uint8) if b is not None else torch.zeros(0, dtype=torch.uint8)
        n = torch.tensor([len(t)], dtype=torch.int64)
        self._dist.broadcast(n, src, group=self._group)
        if self.rank != src:
            t = torch.zeros(int(n[0]), dtype=torch.uint8)
        self._dist.broadcast(t, src, group=self._group)
        return bytes(t.tolist())

    def allgather_int(self, x):
        import torch
        out = [torch.zeros(1, dtype=torch.int64) for _ in range(self.world)]
        self._dist.all_gather(out, torch.tensor([int(x)], dtype=torch.int64), group=self._group)
        return [int(o[0]) for o in out]

    def allgather_ints(self, xs):
        """Every rank's list of `world` integers -> the world x world matrix (row r = rank r's list)."""
        import torch
        mine = torch.tensor([int(x) for x in xs], dtype=torch.int64)
        out = [torch.zeros(len(xs), dtype=torch.int64) for _ in range(self.world)]
        self._dist.all_gather(out, mine, group=self._group)
        return [[int(v) for v in o] for o in out]


class LoopbackWorld:
    """All `world` ranks in this process, on this GPU: `run(fn)` calls fn(comm) once per rank, each in its own thread,
    with a `LoopbackComm` whose collectives rendezvous between the threads and move the data with device copies.  The
    kernels, the partition, the slice builds and the probes are the real ones; only ncclSend/ncclRecv/ncclAllReduce are
    replaced."""

    def __init__(self, world):
        self.world = int(world)
        self._barrier = threading.Barrier(self.world)
        self._slots = [None] * self.world

    def exchange(self, rank, item):
        """Every rank deposits an item; returns the list of all of them (valid until the next exchange)."""
        self._barrier.wait()                      # the previous round's readers are done
        self._slots[rank] = item
        self._barrier.wait()
        return list(self._slots)

    def run(self, fn):
        results, errors = [None] * self.world, []

        def work(r):
            try:
                results[r] = fn(LoopbackComm(self, r))
            except BaseException as e:            # noqa: BLE001 -- re-raised in the caller's thread
                errors.append(e)
                self._barrier.abort()
        threads = [threading.Thread(target=work, args=(r,)) for r in range(self.world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            real = [e for e in errors if not isinstance(e, threading.BrokenBarrierError)]
            raise (real or errors)[0]
        return results


class _LoopbackControl(_GatherControlPlane):
    def __init__(self, world_obj, rank):
        self._w, self.rank, self.world = world_obj, rank, world_obj.world

    def _allgather_bytes(self, payload):
        return self._w.exchange(self.rank, bytes(payload))


class LoopbackComm:
    """`Comm`'s interface for one rank of a LoopbackWorld."""

    def __init__(self, world_obj, rank):
        self._w = world_obj
        self.control = _LoopbackControl(world_obj, rank)

    @staticmethod
    def _copy(dst, dst_off, src, src_off, n):
        if n > 0:
            sz = dst.dtype.itemsize
            _lib.check(_lib.load().gki_memcpy_d2d(C.c_void_p(dst.ptr.value + dst_off * sz),
                                                  C.c_void_p(src.ptr.value + src_off * sz), n * sz))

    def allgather_flat(self, dflat):
        _lib.check(_lib.load().gki_device_synchronize())
        parts = self._w.exchange(self.control.rank, dflat)
        counts = [p.n for p in parts]
        out = DeviceFlatKmers.allocate(sum(counts))
        off = 0
        for p in parts:
            for col in ("hashes", "nodes", "ref_offsets", "allele_frequencies"):
                self._copy(getattr(out, col), off, getattr(p, col), 0, p.n)
            off += p.n
        out.n = off
        _lib.check(_lib.load().gki_device_synchronize())
        self._w.exchange(self.control.rank, None)      # nobody frees a shard another rank is still reading
        return out, counts

    def alltoall_flat(self, dflat, send_start):
        _lib.check(_lib.load().gki_device_synchronize())
        me = self.control.rank
        parts = self._w.exchange(me, (dflat, [int(x) for x in send_start]))
        recv_counts = [ss[me + 1] - ss[me] for _, ss in parts]
        recv_start = np.concatenate([[0], np.cumsum(recv_counts)]).astype(np.int64)
        out = DeviceFlatKmers.allocate(int(recv_start[-1]))
        for r, (src, ss) in enumerate(parts):
            for col in ("hashes", "nodes", "ref_offsets", "allele_frequencies"):
                self._copy(getattr(out, col), int(recv_start[r]), getattr(src, col), ss[me], recv_counts[r])
        out.n = int(recv_start[-1])
        _lib.check(_lib.load().gki_device_synchronize())
        self._w.exchange(me, None)
        return out, [int(x) for x in recv_start]

    def allreduce_counts(self, counts):
        _lib.check(_lib.load().gki_device_synchronize())
        host = self._w.exchange(self.control.rank, counts.to_host())
        total = np.sum(np.stack(host).astype(np.uint64), axis=0).astype(np.uint32)      # uint32 wrap like ncclSum
        _lib.check(_lib.load().gki_memcpy_h2d(counts.ptr, _lib.hptr(total), total.nbytes))
        return counts

    def close(self):
        pass


class Comm:
    """RCCL communicator of libgki_hip.so (gki_comm_*)."""

    def __init__(self, control):
        lib = _lib.load()
        self.control = control
        ident = None
        if control.rank == 0:
            buf = C.create_string_buffer(128)
            _lib.check(lib.gki_comm_get_unique_id(buf))
            ident = buf.raw
        ident = control.broadcast_bytes(ident, 0)
        h = C.c_void_p()
        _lib.check(lib.gki_comm_create(C.byref(h), control.world, control.rank, ident))
        self.handle = h

    def allgather_flat(self, dflat):
        """Every rank's DeviceFlatKmers -> the concatenation in rank order, on every GPU."""
        counts = self.control.allgather_int(dflat.n)
        total = sum(counts)
        out = DeviceFlatKmers.allocate(total)
        carr = (C.c_int64 * len(counts))(*counts)
        _lib.check(_lib.load().gki_comm_allgather_flat(
            self.handle, carr, dflat.hashes.ptr, dflat.nodes.ptr, dflat.ref_offsets.ptr, dflat.allele_frequencies.ptr,
            out.hashes.ptr, out.nodes.ptr, out.ref_offsets.ptr, out.allele_frequencies.ptr))
        out.n = total
        return out, counts

    def alltoall_flat(self, dflat, send_start):
        """dflat partitioned by destination rank (slice r = [send_start[r], send_start[r+1])) -> the records every
        rank sent to this one, in rank order.  Returns (DeviceFlatKmers, recv_start)."""
        world = self.control.world
        matrix = self.control.allgather_ints([send_start[r + 1] - send_start[r] for r in range(world)])
        recv_counts = [matrix[r][self.control.rank] for r in range(world)]
        recv_start = np.concatenate([[0], np.cumsum(recv_counts)]).astype(np.int64)
        out = DeviceFlatKmers.allocate(int(recv_start[-1]))
        ss = (C.c_int64 * (world + 1))(*[int(x) for x in send_start])
        rs = (C.c_int64 * (world + 1))(*[int(x) for x in recv_start])
        _lib.check(_lib.load().gki_comm_alltoall_flat(
            self.handle, ss, dflat.hashes.ptr, dflat.nodes.ptr, dflat.ref_offsets.ptr, dflat.allele_frequencies.ptr,
            rs, out.hashes.ptr, out.nodes.ptr, out.ref_offsets.ptr, out.allele_frequencies.ptr))
        return out, [int(x) for x in recv_start]

    def allreduce_counts(self, counts):
        """In-place sum over ranks of a uint32 DeviceArray."""
        _lib.check(_lib.load().gki_comm_allreduce_u32(self.handle, counts.ptr, counts.n))
        return counts

    def close(self):
        if self.handle is not None:
            _lib.load().gki_comm_destroy(self.handle)
            self.handle = None


def find_sharded(graph_arrays, k, critical_graph_paths, rank, world, **finder_kwargs):
    """This rank's share of DenseKmerFinder.find(): FlatKmers columns in HBM."""
    from .kmer_finder import DenseKmerFinder
    a, b = shard_range(graph_arrays, critical_graph_paths, rank, world)
    finder = DenseKmerFinder(graph_arrays, k, critical_graph_paths=critical_graph_paths,
                             start_at_critical_path_number=a, stop_at_critical_path_number=b, **finder_kwargs)
    out = finder.find_flat_on_device()
    finder.synchronize()
    return out


def build_index_sharded(graph_arrays, k, critical_graph_paths, comm, modulo=452930477, skip_frequencies=False,
                        **finder_kwargs):
    """find (sharded) -> all-gather over RCCL -> CollisionFreeKmerIndex build on every rank (replicated index, as
    KAGE's lookup side wants it).  Returns (DeviceIndex, per-rank record counts)."""
    from .collision_free_kmer_index import DeviceIndex
    mine = find_sharded(graph_arrays, k, critical_graph_paths, comm.control.rank, comm.control.world, **finder_kwargs)
    everything, counts = comm.allgather_flat(mine)
    mine.free()
    index = DeviceIndex.build(everything, modulo, skip_frequencies)
    everything.free()
    return index, counts


def build_index_partitioned(graph_arrays, k, critical_graph_paths, comm, modulo=452930477, skip_frequencies=False,
                            **finder_kwargs):
    """Bucket-range partitioned build: find (sharded by critical paths) -> partition the rank's
    records by owning rank (bucket range) -> one all-to-all(v) over RCCL -> every rank sorts and keeps only its
    1/world of the directory.  Returns this rank's slice as a DeviceIndex (bucket_begin / n_buckets set)."""
    from .collision_free_kmer_index import DeviceIndex, bucket_range, partition_by_bucket_range
    rank, world = comm.control.rank, comm.control.world
    mine = find_sharded(graph_arrays, k, critical_graph_paths, rank, world, **finder_kwargs)
    by_dest, send_start = partition_by_bucket_range(mine, modulo, world)
    mine.free()
    received, _ = comm.alltoall_flat(by_dest, send_start)
    by_dest.free()
    lo, hi = bucket_range(modulo, world, rank)
    index = DeviceIndex.build(received, modulo, skip_frequencies, bucket_begin=lo, n_buckets=hi - lo)
    received.free()
    return index


def map_reads_partitioned(index_slice, comm, letters, read_start, k, n_nodes, strands=3, max_hits=10):
    """Node counts of all k-mers of the reads against a bucket-partitioned index: every rank probes its slice with all
    reads (a k-mer of another slice misses after the modulo, without a memory access), then one all-reduce of the
    uint32 histogram.  Returns the summed counts (DeviceArray) on every rank."""
    counts, _, _ = index_slice.count_nodes_from_reads(letters, read_start, k, n_nodes, strands, max_hits)
    return comm.allreduce_counts(counts)


def map_reads_replicated(index, comm, letters, read_start, k, n_nodes, strands=3, max_hits=10):
    """BASELINE configs[4] on N GPUs, replicas: every rank holds the whole index (DeviceIndex) and maps ITS share of the
    reads; the uint32 node histograms are summed with one all-reduce.  Returns the summed counts on every rank."""
    counts, _, _ = index.count_nodes_from_reads(letters, read_start, k, n_nodes, strands, max_hits)
    return comm.allreduce_counts(counts)


def read_shard(n_reads, rank, world):
    """[begin, end) of the reads rank `rank` maps."""
    return n_reads * rank // world, n_reads * (rank + 1) // world
