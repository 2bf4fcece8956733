#!/usr/bin/env python3
"""Does RCCL form a communicator when several ranks share ONE device?  (The builder's box has one GPU; the bench's
multi-rank path is rehearsed there.)  Parent: starts W fresh children before touching the GPU.  Child: control plane,
gki_comm_create, one all-gather(v) and one all-to-all(v) of tiny FlatKmers columns, prints what it received."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child():
    import numpy as np
    from graph_kmer_index_amd import _lib
    from graph_kmer_index_amd.parallel import SocketControlPlane, Comm
    from graph_kmer_index_amd.flat_kmers import DeviceFlatKmers, FlatKmers
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    lib = _lib.load()
    _lib.check(lib.gki_set_device(int(os.environ.get("LOCAL_RANK", "0")) % _lib.device_count()))
    plane = SocketControlPlane(rank, world)
    try:
        comm = Comm(plane)
    except Exception as e:           # noqa: BLE001 -- the probe reports whatever RCCL says
        print("rank %d: communicator refused: %s" % (rank, e), flush=True)
        plane.close()
        return 3
    n = 1000 + 10 * rank
    fl = FlatKmers(np.arange(n, dtype=np.uint64) + 10 ** 6 * rank, np.full(n, rank, np.uint32),
                   np.arange(n, dtype=np.uint64), np.ones(n, np.float32))
    d = DeviceFlatKmers.from_flat_kmers(fl)
    out, counts = comm.allgather_flat(d)
    nodes = out.nodes.to_host(out.n)
    ok = counts == [1000 + 10 * r for r in range(world)] and all(
        np.all(nodes[sum(counts[:r]):sum(counts[:r + 1])] == r) for r in range(world))
    print("rank %d: all-gather over RCCL with %d ranks on shared device: %s" % (rank, world, "ok" if ok else "WRONG"), flush=True)
    comm.close()
    plane.close()
    return 0 if ok else 4


def main():
    if "RANK" in os.environ:
        sys.exit(child())
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT="29611", NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "WARN"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
    rc = 0
    for p in procs:
        try:
            rc = max(rc, p.wait(timeout=180))
        except subprocess.TimeoutExpired:
            p.kill()
            rc = max(rc, 9)
    sys.exit(rc)


if __name__ == "__main__":
    main()
