"""2-bit k-mer hash helpers with the reference's names (kmer_hashing.py:4-65,
snp_kmer_finder.py:14-26).  hash = sum_i base[i] * 4^i, a/n/m=0 c=1 g=2 t=3.

Array-sized operations (reverse complement, complement, hashing of sequences) run on the GPU
through libgki_hip.so; the scalar helpers are plain integer arithmetic."""
import numpy as np

from . import _lib
from .flat_kmers import letter_sequence_to_numeric, numeric_to_letter_sequence


def power_array(k):
    return np.power(4, np.arange(k - 1, -1, -1)).astype(np.uint64)


def reverse_power_array(k):
    return np.power(4, np.arange(k)).astype(np.uint64)


def _device_unary(fn_name, hashes, k):
    assert k <= 31
    h = np.ascontiguousarray(np.asarray(hashes)).astype(np.uint64)
    if h.size == 0:
        return h
    _lib.require_device()
    d = _lib.DeviceArray.from_host(h)
    _lib.check(getattr(_lib.load(), fn_name)(d.ptr, h.size, int(k), d.ptr))
    out = d.to_host()
    d.free()
    return out


def kmer_hashes_to_reverse_complement_hash(hashes, k):
    """kmer_hashing.py:24-28: rc = sum_j (3 - d_j) 4^(k-1-j); on device: bit-reverse the 2-bit
    digits of ~x and shift by 64-2k."""
    return _device_unary("gki_reverse_complement", hashes, k)


def kmer_hashes_to_reverse_complement_hash_chunked(hashes, k, chunk_size=1000000):
    return kmer_hashes_to_reverse_complement_hash(hashes, k)


def kmer_hash_to_reverse_complement_hash(hash, k):
    """kmer_hashing.py:12-13 for ONE hash, in plain integer arithmetic (a device round trip per scalar costs more than
    the 2k bits are worth): complement every 2-bit digit, reverse the digit order.  Returns numpy.uint64 like the
    reference's `[...][0]`."""
    assert k <= 31
    x = ~int(hash) & ((1 << (2 * k)) - 1)                                           # digit d -> 3 - d
    x = ((x >> 2) & 0x3333333333333333) | ((x & 0x3333333333333333) << 2)          # reverse the 32 digits of the 64-bit
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0F) | ((x & 0x0F0F0F0F0F0F0F0F) << 4)          # word: digits inside a nibble, nibbles
    x = int.from_bytes(x.to_bytes(8, "little"), "big")                              # inside a byte, then the bytes
    return np.uint64(x >> (64 - 2 * k))


def kmer_hashes_to_complement_hashes(hashes, k):
    """kmer_hashing.py:31-36."""
    return _device_unary("gki_complement", hashes, k)


def kmer_hashes_to_bases(hashes, k):
    """kmer_hashing.py:53-65: digit j of every hash (first base first), uint64[n, k]."""
    h = np.asarray(hashes).astype(np.uint64)
    shifts = (2 * np.arange(k)).astype(np.uint64)
    return (h[:, None] >> shifts[None, :]) & np.uint64(3)


def kmer_hashes_to_complement_bases(hashes, k):
    return np.uint64(3) - kmer_hashes_to_bases(hashes, k)


def kmer_to_hash_fast(kmer, k):
    """snp_kmer_finder.py:24-26."""
    assert kmer.dtype == np.uint64
    return int(np.sum(kmer * reverse_power_array(k)))


def sequence_to_kmer_hash(sequence):
    """snp_kmer_finder.py:19-20."""
    return kmer_to_hash_fast(letter_sequence_to_numeric(sequence).astype(np.uint64), len(sequence))


def kmer_hash_to_sequence(hash, k):
    """snp_kmer_finder.py:14-16."""
    bases = kmer_hashes_to_bases(np.array([hash]), k)[0]
    return ''.join(numeric_to_letter_sequence(bases))
