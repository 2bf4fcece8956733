"""Time the REFERENCE's own Python path (build container only; needs /root/reference and the test-only stand-ins)
on a down-scaled graph from the bench generator, next to the oracle on the same input.  Writes
profiles/reference_python_timing.json.  This is the "kind: reference" CPU number that cannot be measured on the GPU
box (the reference does not travel); bench.py reports the oracle ("kind: port") there."""
import json
import logging
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(ROOT, "tests", "standins"), "/root/reference", ROOT]
logging.disable(logging.CRITICAL)

import numpy as np  # noqa: E402
from graph_kmer_index.kmer_finder import DenseKmerFinder  # noqa: E402
from graph_kmer_index.critical_graph_paths import CriticalGraphPaths  # noqa: E402
from graph_kmer_index.flat_kmers import FlatKmers  # noqa: E402
from graph_kmer_index.collision_free_kmer_index import CollisionFreeKmerIndex  # noqa: E402
from graph_kmer_index_amd.graph import synthetic_snp_graph  # noqa: E402
from oracle import oracle  # noqa: E402


_G = {}


def _chunk(args):
    """One worker of the reference's own chunked scheme (command_line_interface.py:588-601)."""
    a, b = args
    f = DenseKmerFinder(_G["g"], 31, critical_graph_paths=_G["cp"], only_save_one_node_per_kmer=True, max_variant_nodes=5,
                        start_at_critical_path_number=a, stop_at_critical_path_number=b)
    f.find()
    return len(f.get_flat_kmers(v="1")._hashes)


def main():
    G, S, k = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3000000, None, 31
    S = G // 600
    g = synthetic_snp_graph(G, S, k=k, seed=1234)
    t = time.perf_counter(); cp = CriticalGraphPaths.from_graph(g, k); t_crit = time.perf_counter() - t
    f = DenseKmerFinder(g, k, critical_graph_paths=cp, only_save_one_node_per_kmer=True, max_variant_nodes=5)
    t = time.perf_counter(); f.find(); fl = f.get_flat_kmers(v="1"); t_find = time.perf_counter() - t
    n = len(fl._hashes)
    flat = FlatKmers(fl._hashes.astype(np.int64), fl._nodes, fl._ref_offsets, fl._allele_frequencies)
    t = time.perf_counter(); idx = CollisionFreeKmerIndex.from_flat_kmers(flat, modulo=20000003, skip_frequencies=True)
    t_build = time.perf_counter() - t
    q = fl._hashes[:: max(1, n // 20000)]
    t = time.perf_counter()
    for x in q:
        idx.get(int(x))
    t_get = time.perf_counter() - t
    # the reference's own multi-process scheme: 8 processes, contiguous ranges of critical paths
    import multiprocessing as mp
    _G["g"], _G["cp"] = g, cp
    n_proc = 8
    cuts = [len(cp.nodes) * i // n_proc for i in range(n_proc + 1)]
    t = time.perf_counter()
    with mp.get_context("fork").Pool(n_proc) as pool:
        counts = pool.map(_chunk, list(zip(cuts[:-1], cuts[1:])))
    t_pool = time.perf_counter() - t
    assert sum(counts) == n
    t = time.perf_counter(); o = oracle.find(g, k, (cp.nodes, cp.offsets), True, 5); t_oracle = time.perf_counter() - t
    assert np.array_equal(o["kmers"], fl._hashes)
    res = {"graph": {"ref_bases": G, "snp_bubbles": S, "k": k, "records": n}, "host": "build container, 1 core of 8 vCPU",
           "reference_python": {"critical_paths_s": t_crit, "find_s": t_find, "find_kmers_per_s": n / t_find,
                                "index_build_skipfreq_s": t_build, "index_build_records_per_s": n / t_build,
                                "get_per_s": len(q) / t_get},
           "reference_python_8_processes": {"scheme": "critical-path chunks, multiprocessing pool (command_line_interface.py:588-614)",
                                            "processes": n_proc, "vcpus": os.cpu_count(), "find_wall_s": t_pool,
                                            "find_kmers_per_s": n / t_pool},
           "oracle_c_port": {"find_s": t_oracle, "find_kmers_per_s": n / t_oracle}}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "reference_python_timing.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
