import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import numpy as np
import soak_parity as sp
from graph_kmer_index_amd import DenseKmerFinder, CriticalGraphPaths
from gpu_util import finder_cols
from golden_cases import canonical_order
from oracle import oracle
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
k = int(rng.integers(2, 32)); M = int(rng.choice([0, 1, 2, 3, 4, 5, 100])); one = bool(rng.integers(0, 2))
mode, g = sp.make_graph(rng, k)
crit = oracle.critical_paths(g, k)
kw = {}
if len(crit[0]) > 2 and rng.random() < 0.4:
    a = int(rng.integers(0, len(crit[0]))); b = int(rng.integers(a, len(crit[0]) + 1))
    kw = dict(start_at_critical_path_number=a, stop_at_critical_path_number=b)
print("k", k, "M", M, "one", one, mode, kw)
print("sizes", g.node_size.tolist()); print("is_ref", g.is_ref.tolist())
print("edges", {n: g.edges[g.edge_start[n]:g.edge_start[n+1]].tolist() for n in range(g.n_nodes) if g.edge_start[n+1] > g.edge_start[n]})
print("crit", list(zip(crit[0].tolist(), crit[1].tolist())))
exp = oracle.find(g, k, crit, one, M, **kw)
if "gpu" in sys.argv:
    f = DenseKmerFinder(g, k, critical_graph_paths=CriticalGraphPaths(crit[0], crit[1]), only_save_one_node_per_kmer=one, max_variant_nodes=M, **kw)
    f.find(); got = finder_cols(f)
    def rows(d): return sorted(zip(d["start_nodes"].tolist(), d["start_offsets"].tolist(), d["kmers"].tolist(), d["nodes"].tolist()))
    ge, ee = rows(got), rows(exp)
    from collections import Counter
    cg, ce = Counter(ge), Counter(ee)
    print("extra in gpu:", sorted((cg - ce).elements())); print("missing in gpu:", sorted((ce - cg).elements()))
else:
    print(len(exp["kmers"]))
if "gpu" in sys.argv:
    for n in set(r[0] for r in (cg - ce)) | set(r[0] for r in (ce - cg)):
        print("node", n, "seq", g.seq[g.seq_start[n]:g.seq_start[n + 1]].tolist(), "preds", g.rev_edges[g.rev_start[n]:g.rev_start[n + 1]].tolist())
        print(" gpu records at node:", [r for r in ge if r[0] == n]); print(" exp records at node:", [r for r in ee if r[0] == n])
    f2 = DenseKmerFinder(g, k, critical_graph_paths=CriticalGraphPaths(crit[0], crit[1]), only_save_one_node_per_kmer=one, max_variant_nodes=M)
    f2.find(); full = rows(finder_cols(f2)); print(" full-run gpu records at those nodes:", [r for r in full if r[0] in set(x[0] for x in (cg - ce))])
