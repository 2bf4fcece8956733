"""Command-line drivers with the reference's sub-command and option names for the in-scope path
(command_line_interface.py:156-174 make_from_flat, :553-638 index, :640-667 find_critical_paths /
add_reverse_complements), writing the reference's .npz formats (flat_kmers.py:65-68,
collision_free_kmer_index.py:393-402).

    python -m graph_kmer_index_amd.command_line_interface index -g graph.npz -k 31 -o flat
    python -m graph_kmer_index_amd.command_line_interface make_from_flat -f flat -o index

`-g` takes an obgraph file when obgraph is installed, else a GraphArrays .npz (GraphArrays.to_file).  `index -t N` runs
one process per GPU (at most N, at most the visible devices), each on its own range of critical-path numbers, and
concatenates the shards in rank order -- the reference's process pool over chunks (:575-614) with GPUs for workers.
"""
import argparse
import logging
import sys
import numpy as np

from .collision_free_kmer_index import CollisionFreeKmerIndex
from .critical_graph_paths import CriticalGraphPaths
from .flat_kmers import FlatKmers
from .graph import GraphArrays
from .kmer_finder import DenseKmerFinder


def load_graph(file_name):
    try:
        return GraphArrays.from_file(file_name)                 # GraphArrays.to_file dump
    except (KeyError, ValueError, OSError):
        from obgraph import Graph                               # the reference's loader (needs obgraph installed)
        return GraphArrays.from_obgraph(Graph.from_file(file_name))


def load_critical_paths(file_name):
    if file_name is None:
        return None
    try:
        d = np.load(file_name if file_name.endswith(".npz") else file_name + ".npz")
        return CriticalGraphPaths(d["nodes"], d["offsets"])
    except (FileNotFoundError, KeyError, ValueError):
        from shared_memory_wrapper import from_file      # the reference's serialisation, when available
        return from_file(file_name)


def _bool(x):
    return bool(x) if not isinstance(x, str) else x.lower() not in ("", "0", "false", "no")


def index(args):
    if args.shard is None and max(1, args.n_threads) > 1:
        ranks = args.ranks or min(args.n_threads, max(1, _visible_devices()))
        if ranks > 1:
            return index_on_ranks(args, ranks)
        logging.info("-t %d: one device visible, one process" % args.n_threads)
    shard = None
    if args.shard is not None:                       # one rank of `index -t N`: its device FIRST -- the critical paths
        from . import _lib                           # below are computed on the device and the graph upload is cached on
        shard = tuple(int(x) for x in args.shard.split("/"))     # the graph object: both must land on this rank's GPU
        _lib.check(_lib.load().gki_set_device(shard[0] % max(1, _lib.device_count())))
    graph = load_graph(args.graph)
    k = args.kmer_size
    cp = load_critical_paths(args.critical_graph_paths) or CriticalGraphPaths.from_graph(graph, k)
    whitelist = None
    if args.whitelist is not None:
        whitelist = CollisionFreeKmerIndex.from_file(args.whitelist)                      # :634 (`kmer in whitelist`)
    chunk = {}
    if shard is not None:                            # its range of critical-path numbers
        from .sharding import shard_range
        r, w = shard
        a, b = shard_range(GraphArrays.from_obgraph(graph), cp, r, w)
        chunk = dict(start_at_critical_path_number=a, stop_at_critical_path_number=b)
    finder = DenseKmerFinder(graph, k, critical_graph_paths=cp, max_variant_nodes=args.max_variant_nodes,
                             only_save_one_node_per_kmer=True, whitelist=whitelist, **chunk)      # :559-565
    dflat = finder.find_flat_on_device(split_layout=False)      # with a whitelist: membership probe + compaction in HBM
    finder.synchronize()
    if args.include_reverse_complement and args.shard is None:                            # :616-620, still in HBM
        from .flat_kmers import DeviceFlatKmers
        both = DeviceFlatKmers.from_multiple_flat_kmers([dflat, dflat.get_reverse_complement_flat_kmers(k)])
        dflat.free()
        dflat = both
    flat = dflat.to_flat_kmers()
    dflat.free()
    logging.info("N kmers in flat kmers: %d" % len(flat._hashes))
    flat.to_file(args.out_file_name)


def _visible_devices():
    from . import _lib
    return _lib.device_count()


def index_on_ranks(args, ranks):
    """`index -t N` (command_line_interface.py:575-614 spreads n_threads * 20 chunks of critical-path numbers over a
    process pool and concatenates the results in chunk order): here one process per GPU, each taking one contiguous
    range of critical-path numbers balanced by bases (sharding.shard_range -- no k-window crosses a critical point, so
    the ranges need no halo), writing its FlatKmers shard; this process concatenates the shards in rank order
    (flat_kmers.py:71-90) and adds the reverse complements afterwards like the reference (:616-620)."""
    import os
    import subprocess
    import tempfile
    k = args.kmer_size
    with tempfile.TemporaryDirectory(prefix="gki_index_") as tmp:
        procs = []
        for r in range(ranks):
            cmd = [sys.executable, "-m", "graph_kmer_index_amd.command_line_interface", "index", "-g", args.graph, "-k", str(k),
                   "-o", os.path.join(tmp, "shard_%d" % r), "-v", str(args.max_variant_nodes), "--shard", "%d/%d" % (r, ranks)]
            if args.critical_graph_paths:
                cmd += ["-c", args.critical_graph_paths]
            if args.whitelist:
                cmd += ["-w", args.whitelist]
            env = dict(os.environ)
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
            procs.append(subprocess.Popen(cmd, env=env))
        codes = [p.wait() for p in procs]
        if any(codes):
            raise RuntimeError("index: rank exit codes %s" % codes)
        flat = FlatKmers.from_multiple_flat_kmers([FlatKmers.from_file(os.path.join(tmp, "shard_%d" % r)) for r in range(ranks)])
    if args.include_reverse_complement:
        flat = FlatKmers.from_multiple_flat_kmers([flat, flat.get_reverse_complement_flat_kmers(k=k)])
    logging.info("N kmers in flat kmers: %d (from %d ranks)" % (len(flat._hashes), ranks))
    flat.to_file(args.out_file_name)


def make_from_flat(args):
    flat = FlatKmers.from_file(args.flat_index)
    if args.add_reverse_complements:
        flat = FlatKmers.from_multiple_flat_kmers([flat, flat.get_reverse_complement_flat_kmers(k=args.kmer_size)])
    if args.make_minimal:
        raise NotImplementedError("MinimalKmerIndex is out of scope (broken on NumPy >= 1.24 in the reference)")
    idx = CollisionFreeKmerIndex.from_flat_kmers(flat, modulo=args.hash_modulo, skip_frequencies=args.skip_frequencies,
                                                 skip_singletons=args.skip_singletons)
    idx.to_file(args.out_file_name)


def find_critical_paths(args):
    cp = CriticalGraphPaths.from_graph(load_graph(args.graph), args.kmer_size)
    cp._make_index()
    np.savez(args.out_file_name, nodes=cp.nodes, offsets=cp.offsets)


def add_reverse_complements(args):
    flat = FlatKmers.from_file(args.flat_kmers)
    FlatKmers.from_multiple_flat_kmers([flat, flat.get_reverse_complement_flat_kmers(k=args.kmer_size)]) \
        .to_file(args.out_file_name)


def build_parser():
    parser = argparse.ArgumentParser(description="graph_kmer_index on MI355X (in-scope sub-commands)")
    sub = parser.add_subparsers()
    p = sub.add_parser("index")
    p.add_argument("-g", "--graph", required=True)
    p.add_argument("-c", "--critical_graph_paths", required=False)
    p.add_argument("-p", "--position_id", required=False)
    p.add_argument("-k", "--kmer-size", type=int, default=31)
    p.add_argument("-o", "--out-file-name", required=True)
    p.add_argument("-t", "--n-threads", type=int, default=1)
    p.add_argument("-w", "--whitelist", required=False)
    p.add_argument("-r", "--include-reverse-complement", type=_bool, default=False)
    p.add_argument("-O", "--only-save-one-node-per-kmer", type=_bool, default=False)
    p.add_argument("-v", "--max-variant-nodes", type=int, default=5)
    p.add_argument("--ranks", type=int, default=0, help="processes of `-t N` (default: min(N, visible devices); more ranks "
                   "than devices share them round-robin)")
    p.add_argument("--shard", default=None, help=argparse.SUPPRESS)       # R/W: this process is rank R of `index -t`
    p.set_defaults(func=index)
    p = sub.add_parser("make_from_flat")
    p.add_argument("-o", "--out_file_name", required=True)
    p.add_argument("-f", "--flat-index", required=True)
    p.add_argument("-m", "--hash_modulo", type=int, default=452930477)
    p.add_argument("-S", "--skip-frequencies", type=_bool, default=False)
    p.add_argument("-M", "--make-minimal", type=_bool, default=False)
    p.add_argument("-r", "--add-reverse-complements", type=_bool, default=False)
    p.add_argument("-k", "--kmer-size", type=int, default=31)
    p.add_argument("-s", "--skip-singletons", type=_bool, default=False)
    p.set_defaults(func=make_from_flat)
    p = sub.add_parser("find_critical_paths")
    p.add_argument("-g", "--graph", required=True)
    p.add_argument("-k", "--kmer-size", type=int, default=31)
    p.add_argument("-o", "--out-file-name", required=True)
    p.set_defaults(func=find_critical_paths)
    p = sub.add_parser("add_reverse_complements")
    p.add_argument("-f", "--flat-kmers", required=True)
    p.add_argument("-o", "--out-file-name", required=True)
    p.add_argument("-k", "--kmer-size", type=int, required=True)
    p.set_defaults(func=add_reverse_complements)
    return parser


def main(argv=None):
    logging.basicConfig(level=logging.INFO)
    parser = build_parser()
    args = parser.parse_args(sys.argv[1:] if argv is None else argv)
    if not hasattr(args, "func"):
        parser.print_help()
        return 1
    args.func(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
