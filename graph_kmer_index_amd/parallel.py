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
