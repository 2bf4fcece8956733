"""Multi-GPU index build: one process per GPU, enumeration sharded by critical-path ranges, one RCCL
all-gather(v) of the finished FlatKmers columns over xGMI, then the CollisionFreeKmerIndex build.

Mirrors the reference's CLI `index` + `make_from_flat` pair (command_line_interface.py:553-622, 156-174) where a
process pool runs `DenseKmerFinder` per chunk and the results are concatenated in chunk order.

The control plane (exchange of the RCCL id and of the per-rank record counts) is any object with
`broadcast_bytes(b, src)` and `allgather_int(x)`; `TorchControlPlane` implements it over torch.distributed (gloo).
torch is never used for compute or GPU memory."""
import ctypes as C
import numpy as np

from . import _lib
from .flat_kmers import DeviceFlatKmers
from .sharding import shard_range


class TorchControlPlane:
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
