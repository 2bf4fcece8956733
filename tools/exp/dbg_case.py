import sys; sys.path[:0]=['.','tests']
import numpy as np
from golden_cases import REFERENCE_TEST_GRAPHS
from graph_kmer_index_amd import GraphArrays, DenseKmerFinder
from oracle import oracle
name = sys.argv[1]
seqs, edges, lin, k, kw = REFERENCE_TEST_GRAPHS[name]
g = GraphArrays.from_dicts(seqs, edges, lin)
o = oracle.find(g, k)
f = DenseKmerFinder(g, k); f.find(); fl = f.get_flat_kmers()
exp = sorted(zip(o["start_nodes"].tolist(), o["start_offsets"].tolist(), o["kmers"].tolist(), o["nodes"].tolist()))
got = sorted(zip(fl._start_nodes.tolist(), fl._start_offsets.tolist(), fl._hashes.tolist(), fl._nodes.tolist()))
print("exp", exp); print("got", got)
