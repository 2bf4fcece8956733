"""gki_simulate_reads (the device-side read simulator behind bench.py's read_mapping record, BASELINE configs[4] /
SURVEY.md 8d C5) against its NumPy restatement, and the reads it makes mapped on the GPU against the oracle's loop of
read_kmers + CollisionFreeKmerIndex.get (read_kmers.py:14-70, collision_free_kmer_index.py:303-315)."""
import ctypes as C
import numpy as np
import pytest

from graph_kmer_index_amd import _lib, DenseKmerFinder, FlatKmers
from graph_kmer_index_amd.collision_free_kmer_index import CollisionFreeKmerIndex
from graph_kmer_index_amd.graph import synthetic_snp_graph, synthetic_haplotype_sequence
from oracle import oracle

pytestmark = pytest.mark.gpu
M64 = np.uint64


def _mix(x):
    x = (x + M64(0x9E3779B97F4A7C15)).astype(M64)
    x = ((x ^ (x >> M64(30))) * M64(0xBF58476D1CE4E5B9)).astype(M64)
    x = ((x ^ (x >> M64(27))) * M64(0x94D049BB133111EB)).astype(M64)
    return x ^ (x >> M64(31))


def simulate_reads_numpy(hap, n_reads, L, seed, p_sub, p_random, first_read=0):
    """The rule written in csrc/gki_measure.hip above k_simulate_reads."""
    with np.errstate(over="ignore"):
        r = np.arange(first_read, first_read + n_reads, dtype=M64)
        sk = M64(seed) * M64(0xD1342543DE82EF95)
        h0, h1 = _mix(M64(2) * r + sk), _mix(M64(2) * r + M64(1) + sk)
        start = (h0 % M64(len(hap) - L + 1)).astype(np.int64)
        is_random = (h1 & M64(0xFFFF)) < M64(int(p_random * 65536.0 + 0.5))
        rc = ((h1 >> M64(16)) & M64(1)) == 1
        i = np.arange(L, dtype=M64)
        hb = _mix(_mix(np.array([seed], dtype=M64))[0] + r[:, None] * M64(L) + i[None, :])
        code = hap[start[:, None] + i[None, :].astype(np.int64)].astype(np.uint32) & 3
        sub = ((hb >> M64(8)) & M64(0xFFFF)) < M64(int(p_sub * 65536.0 + 0.5))
        code = np.where(sub, (code + 1 + ((hb >> M64(24)) % M64(3)).astype(np.uint32)) & 3, code)
        code = np.where(is_random[:, None], (hb & M64(3)).astype(np.uint32), code)
        code = np.where(rc[:, None], 3 - code[:, ::-1], code)
    return np.frombuffer(b"ACGT", dtype=np.uint8)[code].reshape(-1), start, is_random, rc


def _device_reads(hap, n_reads, L, seed, p_sub, p_random, first_read=0):
    lib = _lib.load()
    d_hap = _lib.DeviceArray.from_host(hap)
    d_letters = _lib.DeviceArray(n_reads * L, np.uint8)
    _lib.check(lib.gki_simulate_reads(d_hap.ptr, len(hap), n_reads, L, seed, p_sub, p_random, first_read, d_letters.ptr))
    out = d_letters.to_host()
    d_hap.free()
    d_letters.free()
    return out


@pytest.mark.parametrize("L,seed,p_sub,p_random", [(150, 99, 0.01, 0.1), (37, 5, 0.25, 0.0), (150, 1, 0.0, 1.0), (1, 7, 0.5, 0.5)])
def test_device_reads_equal_the_numpy_restatement(L, seed, p_sub, p_random):
    rng = np.random.default_rng(seed)
    hap = rng.integers(0, 4, size=100000, dtype=np.uint8)
    n = 20000
    want, start, is_random, rc = simulate_reads_numpy(hap, n, L, seed, p_sub, p_random)
    got = _device_reads(hap, n, L, seed, p_sub, p_random)
    assert np.array_equal(got, want)
    assert set(np.unique(got).tolist()) <= set(b"ACGT")
    assert abs(is_random.mean() - p_random) < 0.02 and (L == 1 or 0.4 < rc.mean() < 0.6)
    # a batch starting at read 7000 is the tail of the full run (counter-based)
    tail = _device_reads(hap, n - 7000, L, seed, p_sub, p_random, first_read=7000)
    assert np.array_equal(tail, want[7000 * L:])


def test_unsubstituted_forward_reads_are_substrings_of_the_haplotype():
    hap = np.random.default_rng(2).integers(0, 4, size=5000, dtype=np.uint8)
    letters, start, is_random, rc = simulate_reads_numpy(hap, 300, 50, 3, 0.0, 0.0)
    got = _device_reads(hap, 300, 50, 3, 0.0, 0.0).reshape(300, 50)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    for r in range(300):
        piece = hap[start[r]:start[r] + 50]
        assert np.array_equal(got[r], lut[3 - piece[::-1]] if rc[r] else lut[piece])


def test_bad_arguments_are_refused():
    lib = _lib.load()
    d = _lib.DeviceArray(64, np.uint8)
    assert lib.gki_simulate_reads(d.ptr, 10, 4, 150, 1, 0.0, 0.0, 0, d.ptr) != 0        # haplotype shorter than a read
    assert lib.gki_simulate_reads(d.ptr, 64, 1, 8, 1, 1.5, 0.0, 0, d.ptr) != 0          # probability out of range
    d.free()


def test_simulated_reads_map_to_the_oracles_node_counts():
    """The bench's read path at 2e4 reads: device reads -> fused k_probe_reads on the variant index == the oracle's loop
    of read_kmers (both strands) + get on those same reads."""
    k = 31
    g = synthetic_snp_graph(400000, 4000, k=k, seed=17)
    f = DenseKmerFinder(g, k, only_save_one_node_per_kmer=True, max_variant_nodes=5)
    f.find()
    fl2 = f.get_flat_kmers()
    bnd = fl2._start_offsets < k - 1                              # the KAGE-like variant index: windows crossing a node boundary
    fl = f.get_flat_kmers(v="1")
    flat = FlatKmers(fl._hashes[bnd], fl._nodes[bnd], fl._ref_offsets[bnd], fl._allele_frequencies[bnd].astype(np.float32))
    modulo = 200003
    idx = CollisionFreeKmerIndex.from_flat_kmers(flat, modulo=modulo)
    o = oracle.index_build(flat._hashes, flat._nodes, flat._ref_offsets, flat._allele_frequencies, modulo=modulo)
    hap = synthetic_haplotype_sequence(g)
    n_reads = 20000
    letters = _device_reads(hap, n_reads, 150, 99, 0.01, 0.1)
    assert np.array_equal(letters, simulate_reads_numpy(hap, n_reads, 150, 99, 0.01, 0.1)[0])
    rs = np.arange(n_reads + 1, dtype=np.int64) * 150
    want, nk, nh = oracle.map_reads(o, letters, rs, k, g.n_nodes, 3, 10)
    dev = idx._device_index()
    counts, n_kmers, hits = dev.count_nodes_from_reads(letters, rs, k, g.n_nodes, 3, 10)
    assert n_kmers == nk == 2 * n_reads * 120 and hits == nh and nh > 1000
    assert np.array_equal(counts.to_host(), want)
