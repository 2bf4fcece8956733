"""Parity at BASELINE.json's full sizes (configs[1] and configs[2]) through the C ABI.

configs[1] (10 Mbp linear) is small enough for the oracle: exact comparison, record by record.
configs[2] (3 Gbp + 5 M SNP bubbles, 3.16e9 records) is checked through size-independent properties:
  * the eight critical-path shards of SURVEY.md 8e partition the full output (counts and per-column
    checksums add up; a checksum of checksums),
  * the split and the by-node layouts hold the same multiset,
  * random chunks of critical paths of the full-size graph equal the oracle's records for the same chunk,
  * interior records re-derived from the sequence on windows of the output.
"""
import numpy as np
import pytest

from graph_kmer_index_amd import CriticalGraphPaths, DenseKmerFinder
from graph_kmer_index_amd.graph import (synthetic_indel_graph, synthetic_linear_graph, synthetic_nested_graph,
                                        synthetic_snp_graph)
from graph_kmer_index_amd.sharding import critical_path_cuts
from gpu_util import assert_same_records, finder_cols
from oracle import oracle

pytestmark = pytest.mark.gpu
K = 31
MASK64 = (1 << 64) - 1


def _checksums(flat):
    cols = (flat.hashes, flat.nodes, flat.ref_offsets, flat.allele_frequencies)
    return [c.checksum(flat.n) for c in cols]


def test_config1_linear_10mbp_equals_oracle_record_by_record():
    g = synthetic_linear_graph(10_000_000, node_len=25_000, seed=1234)
    f = DenseKmerFinder(g, K)
    f.find()
    got = finder_cols(f)
    exp = oracle.find(g, K)
    assert len(got["kmers"]) == len(exp["kmers"]) > 9_900_000
    assert_same_records(got, exp, exact_order=True)


@pytest.mark.parametrize("kind", ["snp", "snp_indel", "nested"])
def test_config2_full_size_properties(kind):
    # "snp_indel": 10 % of the sites 1-bp deletions (empty alt node), 10 % insertions (empty ref-dummy node)
    # "nested": 20 % of the sites an alternative allele that contains a SNP itself (four non-linear nodes, three of them
    #   without a linear-ref predecessor): the GENERAL kernels with gki_classify_nodes' flags, at full size
    M = 5
    if kind == "snp":
        g = synthetic_snp_graph(3_000_000_000, 5_000_000, k=K, seed=1234)
    elif kind == "snp_indel":
        g = synthetic_indel_graph(3_000_000_000, 5_000_000, k=K, seed=1234, p_del=0.1, p_ins=0.1)
    else:
        g = synthetic_nested_graph(3_000_000_000, 5_000_000, k=K, seed=1234, p_nest=0.2)
        M = 8                # <= 2 sites per k bases, <= 3 variant nodes each: the reference's assertion never fires
        from graph_kmer_index_amd.kmer_finder import classify_nodes
        assert classify_nodes(g, K, M)[1]
    cp = CriticalGraphPaths.from_graph(g, K)
    kw = dict(critical_graph_paths=cp, only_save_one_node_per_kmer=True, max_variant_nodes=M)

    # the whole graph, split layout
    full = DenseKmerFinder(g, K, **kw)
    flat = full.find_flat_on_device()
    full.synchronize()
    n, n_int = flat.n, full.interior_records()
    assert n > 3_100_000_000 and 0 < n_int < n
    want = _checksums(flat)

    # interior records are by position: re-derive windows of them from the sequence
    rng = np.random.default_rng(7)
    for a in [0, n_int - 100_000] + rng.integers(0, n_int - 100_000, size=6).tolist():
        a = int(a)
        h = flat.hashes.view(a, 100_000).to_host()
        ro = flat.ref_offsets.view(a, 100_000).to_host().astype(np.int64)      # default position id = global base index
        nodes = flat.nodes.view(a, 100_000).to_host()
        assert np.all(np.diff(ro) > 0)
        node_of = np.searchsorted(g.seq_start, ro, side="right") - 1
        assert np.array_equal(nodes, node_of.astype(np.uint32))
        assert np.all(ro - g.seq_start[node_of] >= K - 1)
        idx = np.arange(0, 100_000, 37)
        win = ro[idx][:, None] - (K - 1) + np.arange(K)[None, :]
        expect = (g.seq[win].astype(np.uint64) << (2 * np.arange(K, dtype=np.uint64))[None, :]).sum(axis=1)
        assert np.array_equal(h[idx], expect)
    # boundary section: by end node, offsets below k-1
    b0 = n_int + int(rng.integers(0, n - n_int - 200_000))
    ro = flat.ref_offsets.view(b0, 200_000).to_host().astype(np.int64)
    node_of = np.searchsorted(g.seq_start, ro, side="right") - 1
    assert np.all(np.diff(node_of) >= 0) and np.all(ro - g.seq_start[node_of] < K - 1)
    flat.free()

    # by-node layout: same multiset
    by_node = full.find_flat_on_device(split_layout=False)
    full.synchronize()
    assert by_node.n == n and _checksums(by_node) == want
    by_node.free()
    full.close()

    # eight shards partition it
    cuts = critical_path_cuts(g, cp, 8)
    total, sums, xors = 0, [0] * 4, [0] * 4
    for a, b in zip(cuts[:-1], cuts[1:]):
        f = DenseKmerFinder(g, K, start_at_critical_path_number=a, stop_at_critical_path_number=b, **kw)
        part = f.find_flat_on_device()
        f.synchronize()
        assert 0.8 * n / 8 < part.n < 1.2 * n / 8            # balanced by bases
        total += part.n
        for i, (s, x) in enumerate(_checksums(part)):
            sums[i] = (sums[i] + s) & MASK64
            xors[i] ^= x
        part.free()
        f.close()
    assert total == n
    assert [(s, x) for s, x in zip(sums, xors)] == want

    # random chunks of the full-size graph against the oracle, record by record (canonical order)
    crit = (cp.nodes, cp.offsets)
    for a in [0, len(cp) - 1500] + rng.integers(0, len(cp) - 1500, size=3).tolist():
        a = int(a)
        f = DenseKmerFinder(g, K, start_at_critical_path_number=a, stop_at_critical_path_number=a + 1500, **kw)
        f.find()
        exp = oracle.find(g, K, crit, True, M, start_at_critical_path_number=a, stop_at_critical_path_number=a + 1500)
        assert len(exp["kmers"]) > 500_000
        assert_same_records(finder_cols(f), exp)
        f.close()
