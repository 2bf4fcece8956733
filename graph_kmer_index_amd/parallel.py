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
        xs = [((int(x) + (1 << 63)) % (1 << 64)) - (1 << 63) for x in xs]       # uint64 values (checksums) wrap to int64
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
    """Star over TCP: rank 0 listens on ONE port, every other rank keeps one connection to it.  A gather is "everyone
    sends to rank 0, rank 0 sends the list back".  Defaults come from the launcher's environment (RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT, as torch.distributed.run and mpirun wrappers export them); the port is MASTER_PORT + 1
    (torch.distributed.run's own rendezvous store listens on MASTER_PORT itself) unless GKI_CONTROL_PORT names another.

    Handshake: a worker sends magic + job token + world + its rank and waits for rank 0's verdict.  The token is
    derived from MASTER_ADDR:MASTER_PORT and the launcher's run id (TORCHELASTIC_RUN_ID, or GKI_JOB_TOKEN), so a worker
    of another job on the same host is turned away instead of being adopted.  Rank 0 drops a connection that does not
    complete the handshake within a few seconds (a port scanner, a health probe), that carries another token or world
    size, a rank outside 1..world-1 or a rank it already holds, and keeps accepting until every rank has joined or the
    timeout passes."""

    MAGIC = b"GKI2"
    HANDSHAKE_SECONDS = 5.0

    @staticmethod
    def job_token(addr, base_port, world):
        import hashlib
        run = os.environ.get("GKI_JOB_TOKEN") or os.environ.get("TORCHELASTIC_RUN_ID") or ""
        return hashlib.sha256(("%s:%d:%d:%s" % (addr, base_port, world, run)).encode()).digest()[:16]

    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=120.0, token=None):
        import time
        self.rank = int(os.environ.get("RANK", 0)) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", 1)) if world is None else int(world)
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        base_port = int(os.environ.get("MASTER_PORT", 29400))
        if port is None:
            port = int(os.environ["GKI_CONTROL_PORT"]) if os.environ.get("GKI_CONTROL_PORT") else base_port + 1
        port = int(port)
        self._peers, self._sock, self._server = [], None, None
        if self.world == 1:
            return
        if not 0 <= self.rank < self.world:
            raise ValueError("control plane: rank %d outside 0..%d" % (self.rank, self.world - 1))
        token = bytes(token) if token is not None else self.job_token(addr, port, self.world)
        hello = self.MAGIC + token + struct.pack("<qq", self.world, self.rank)
        deadline = time.time() + timeout
        if self.rank == 0:
            # a worker that dials before rank 0 listens can, for an instant, hold the port itself (a TCP self-connection
            # from an ephemeral source port equal to the target): retry the bind briefly before calling the port taken
            bind_deadline = time.time() + min(timeout, 3.0)
            while True:
                srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    srv.bind((addr, port))
                    break
                except OSError as e:
                    srv.close()
                    if time.time() > bind_deadline:
                        raise OSError("control plane: rank 0 cannot listen on %s:%d (%s); set GKI_CONTROL_PORT to a free "
                                      "port on every rank" % (addr, port, e)) from e
                    time.sleep(0.1)
            self._server = srv
            srv.listen(max(self.world, 8))
            peers = {}
            while len(peers) < self.world - 1:
                left = deadline - time.time()
                if left <= 0:
                    missing = sorted(set(range(1, self.world)) - set(peers))
                    self.close()
                    raise TimeoutError("control plane: ranks %s did not join %s:%d within %.0f s" % (missing, addr, port, timeout))
                srv.settimeout(min(left, 1.0))
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    continue
                try:
                    conn.settimeout(self.HANDSHAKE_SECONDS)
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    got = _recv_exact(conn, len(hello))
                    w, r = struct.unpack("<qq", got[-16:])
                    if got[:4] != self.MAGIC or got[4:20] != token or w != self.world or not 1 <= r < self.world or r in peers:
                        try:
                            conn.sendall(b"NO")
                        finally:
                            conn.close()
                        continue
                    conn.sendall(b"OK")
                    conn.settimeout(timeout)
                    peers[r] = conn
                except (OSError, ConnectionError, struct.error):
                    conn.close()                       # junk or a half-open connection: keep accepting
            self._peers = [peers[r] for r in range(1, self.world)]
        else:
            refused = False
            while self._sock is None:
                try:
                    sock = socket.create_connection((addr, port), timeout=2.0)
                    if sock.getsockname() == sock.getpeername():     # connected to itself: nobody listens yet
                        sock.close()
                        raise ConnectionError("self-connection")
                    sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    sock.settimeout(max(self.HANDSHAKE_SECONDS, 2.0) + 5.0)
                    sock.sendall(hello)
                    verdict = _recv_exact(sock, 2)
                    if verdict == b"OK":
                        self._sock = sock
                        break
                    sock.close()
                    refused = verdict == b"NO"
                    if refused:
                        break
                except (OSError, ConnectionError):
                    pass
                if time.time() > deadline:
                    raise TimeoutError("control plane: rank 0 did not answer on %s:%d" % (addr, port))
                time.sleep(0.05)
            if refused:
                raise ConnectionError("control plane: rank 0 at %s:%d turned rank %d away (another job's port, a world-size "
                                      "mismatch, or a duplicate rank)" % (addr, port, self.rank))
            self._sock.settimeout(timeout)

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

    def info(self):
        """What RCCL itself reports about the communicator."""
        w, r = C.c_int(0), C.c_int(-1)
        _lib.check(_lib.load().gki_comm_info(self.handle, C.byref(w), C.byref(r)))
        return {"exchange": "rccl", "rccl_ranks": w.value, "rccl_rank": r.value}

    def close(self):
        if self.handle is not None:
            _lib.load().gki_comm_destroy(self.handle)
            self.handle = None


class SharedDeviceComm:
    """`Comm`'s interface for ranks (processes) that share ONE device: a multi-rank run rehearsed on a single GPU.
    RCCL refuses such a communicator (two ranks on one device: ncclInvalidUsage), so the exchanges go through HIP IPC:
    every rank exports the allocations that hold its columns (gki_ipc_export), the 64-byte handles and the slice tables
    travel over the control plane, every rank maps its peers' columns (gki_ipc_open) and copies what it is owed
    device-to-device.  Same results as `Comm`, no xGMI involved; `make_comm` picks it only when ranks share a device."""

    def __init__(self, control):
        self.control = control
        self.handle = None

    def _exchange(self, arrays, meta):
        """Everyone's (handles of `arrays`, meta) -> per rank (list of mapped DeviceArray-like pointers, meta).  Returns
        (views, metas, close) where views[r][i] is the address of rank r's i-th array in THIS process."""
        lib = _lib.load()
        _lib.check(lib.gki_device_synchronize())
        blob = b""
        for a in arrays:
            h, off = C.create_string_buffer(64), C.c_int64(0)
            _lib.check(lib.gki_ipc_export(a.ptr, h, C.byref(off)))
            blob += h.raw + struct.pack("<q", off.value)
        blob += struct.pack("<%dq" % len(meta), *[int(x) for x in meta])
        everyone = self.control._allgather_bytes(blob)
        me, opened, views, metas = self.control.rank, [], [], []
        for r, b in enumerate(everyone):
            ptrs = []
            for i in range(len(arrays)):
                if r == me:
                    ptrs.append(arrays[i].ptr.value)
                    continue
                h, off = b[72 * i:72 * i + 64], struct.unpack_from("<q", b, 72 * i + 64)[0]
                base = C.c_void_p()
                _lib.check(lib.gki_ipc_open(h, C.byref(base)))
                opened.append(base)
                ptrs.append(base.value + off)
            views.append(ptrs)
            metas.append(list(struct.unpack_from("<%dq" % len(meta), b, 72 * len(arrays))))

        def close():
            _lib.check(lib.gki_device_synchronize())
            for base in opened:
                _lib.check(lib.gki_ipc_close(base))
            self.control.barrier()            # nobody frees a column a peer is still reading
        return views, metas, close

    @staticmethod
    def _cols(d):
        return [d.hashes, d.nodes, d.ref_offsets, d.allele_frequencies]

    def allgather_flat(self, dflat):
        lib = _lib.load()
        views, metas, close = self._exchange(self._cols(dflat), [dflat.n])
        counts = [m[0] for m in metas]
        out = DeviceFlatKmers.allocate(sum(counts))
        off = 0
        for r, n in enumerate(counts):
            for i, dst in enumerate(self._cols(out)):
                sz = dst.dtype.itemsize
                if n > 0:
                    _lib.check(lib.gki_memcpy_d2d(C.c_void_p(dst.ptr.value + off * sz), C.c_void_p(views[r][i]), n * sz))
            off += n
        out.n = off
        close()
        return out, counts

    def alltoall_flat(self, dflat, send_start):
        lib = _lib.load()
        me = self.control.rank
        views, metas, close = self._exchange(self._cols(dflat), send_start)
        recv_counts = [m[me + 1] - m[me] for m in metas]
        recv_start = np.concatenate([[0], np.cumsum(recv_counts)]).astype(np.int64)
        out = DeviceFlatKmers.allocate(int(recv_start[-1]))
        for r, m in enumerate(metas):
            for i, dst in enumerate(self._cols(out)):
                sz = dst.dtype.itemsize
                if recv_counts[r] > 0:
                    _lib.check(lib.gki_memcpy_d2d(C.c_void_p(dst.ptr.value + int(recv_start[r]) * sz),
                                                  C.c_void_p(views[r][i] + m[me] * sz), recv_counts[r] * sz))
        out.n = int(recv_start[-1])
        close()
        return out, [int(x) for x in recv_start]

    def allreduce_counts(self, counts):
        lib = _lib.load()
        views, _, close = self._exchange([counts], [counts.n])
        total = np.zeros(counts.n, dtype=np.uint64)
        tmp = np.empty(counts.n, dtype=np.uint32)
        for r in range(self.control.world):
            _lib.check(lib.gki_memcpy_d2h(_lib.hptr(tmp), C.c_void_p(views[r][0]), tmp.nbytes))
            total += tmp
        self.control.barrier()                # every rank has read every buffer before anyone overwrites its own
        total = total.astype(np.uint32)       # uint32 wrap like ncclSum
        _lib.check(lib.gki_memcpy_h2d(counts.ptr, _lib.hptr(total), total.nbytes))
        close()
        return counts

    def info(self):
        return {"exchange": "hip-ipc", "rccl_ranks": None,
                "why": "the ranks share one device and RCCL refuses a communicator with duplicate devices"}

    def close(self):
        pass


def device_identity():
    """(hostname, PCI bus id of the current device): what tells two ranks on one GPU from two ranks on two GPUs."""
    buf = C.create_string_buffer(64)
    _lib.check(_lib.load().gki_device_bus_id(buf, 64))
    return "%s/%s" % (socket.gethostname(), buf.value.decode())


def make_comm(control):
    """The exchange layer for this job: RCCL (`Comm`) when every rank has a device of its own, HIP IPC
    (`SharedDeviceComm`) when ranks share one."""
    mine = device_identity().encode()
    ids = control._allgather_bytes(mine) if hasattr(control, "_allgather_bytes") else [mine]
    if len(set(ids)) < len(ids):
        return SharedDeviceComm(control)
    return Comm(control)


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
