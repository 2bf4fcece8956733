import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from graph_kmer_index_amd import DenseKmerFinder
from graph_kmer_index_amd.graph import synthetic_indel_graph, synthetic_snp_graph
from gpu_util import assert_same_records, finder_cols
from oracle import oracle
seconds, seed0 = float(sys.argv[1]), int(sys.argv[2])
t_end, it, last = time.time() + seconds, 0, time.time()
while time.time() < t_end:
    seed = seed0 * 1000003 + it; it += 1
    rng = np.random.default_rng(seed)
    k = int(rng.choice([31, 31, 23, 15, 8]))
    G = int(rng.integers(100_000, 1_000_000))
    S = G // int(rng.integers(k // 2 + 3, 120))
    M = int(rng.choice([0, 1, 2, 3, 5, 100])); one = bool(rng.integers(0, 2))
    try:
        g = synthetic_indel_graph(G, S, k=k, seed=seed % (1 << 30), p_del=float(rng.uniform(0, 0.3)), p_ins=float(rng.uniform(0, 0.3)),
                                  max_node_len=int(rng.choice([32767, 2000, 300])))
    except AssertionError:
        continue
    try:
        exp, flags = oracle.find(g, k, None, one, M, return_flags=True)
    except oracle.OracleError:
        continue
    f = DenseKmerFinder(g, k, only_save_one_node_per_kmer=one, max_variant_nodes=M)
    try:
        f.find()
    except (ValueError, NotImplementedError):
        continue
    try:
        assert_same_records(finder_cols(f), exp)
    except AssertionError as e:
        print("MISMATCH mid:", seed, k, G, S, M, one, e); sys.exit(1)
    f.close()
    if time.time() - last > 45:
        last = time.time(); print("...", it, "graphs", flush=True)
print("soak mid ok:", it, "graphs")
