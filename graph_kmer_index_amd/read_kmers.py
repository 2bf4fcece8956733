"""ReadKmers with the reference's interface (read_kmers.py:9-88); hashing runs on MI355X.

`get_kmers_from_read_dynamic(read, power_vector)` hashes one read; `hash_reads` is the batched
entry point (many reads per launch, both strands) that the lookup benchmark uses."""
import ctypes as C
import itertools
import numpy as np

from . import _lib
from .flat_kmers import letter_sequence_to_numeric
from .kmer_hashing import kmer_to_hash_fast, power_array


def hash_reads(reads, k, strand=0, return_device=False):
    """k-mer hashes of every read.  reads: list of str/bytes, or (uint8 letters, int64 read_start).
    strand 0: forward; 1: reverse complement of the read (read_kmers.py:21-26).
    Returns (hashes uint64[], out_start int64[n_reads+1])."""
    _lib.require_device()
    lib = _lib.load()
    if isinstance(reads, tuple):
        letters, read_start = reads
        letters = np.ascontiguousarray(letters, dtype=np.uint8)
        read_start = np.ascontiguousarray(read_start, dtype=np.int64)
    else:
        enc = [r.encode("ascii") if isinstance(r, str) else bytes(r) for r in reads]
        read_start = np.zeros(len(enc) + 1, dtype=np.int64)
        np.cumsum([len(e) for e in enc], out=read_start[1:])
        letters = np.frombuffer(b"".join(enc), dtype=np.uint8)
    n_reads = len(read_start) - 1
    d_letters = _lib.DeviceArray.from_host(letters if len(letters) else np.zeros(1, np.uint8))
    d_start = _lib.DeviceArray.from_host(read_start)
    d_out_start = _lib.DeviceArray(n_reads + 1, np.int64)
    n_out = C.c_int64(0)
    _lib.check(lib.gki_hash_reads(d_letters.ptr, d_start.ptr, n_reads, int(k), int(strand), d_out_start.ptr, None, 0,
                                  C.byref(n_out)))
    d_out = _lib.DeviceArray(max(n_out.value, 1), np.uint64)
    _lib.check(lib.gki_hash_reads(d_letters.ptr, d_start.ptr, n_reads, int(k), int(strand), d_out_start.ptr, d_out.ptr,
                                  d_out.n, C.byref(n_out)))
    out_start = d_out_start.to_host()
    d_letters.free(); d_start.free(); d_out_start.free()
    if return_device:
        return d_out, n_out.value, out_start
    out = d_out.to_host(n_out.value)
    d_out.free()
    return out, out_start


class ReadKmers:
    def __init__(self, kmers):
        self.kmers = kmers
        self._power_vector = None

    @classmethod
    def from_fasta_file(cls, fasta_file_name, k, small_k=None, smallest_k=8):
        """read_kmers.py:14-49: per read, forward k-mers of every line, then reverse-complement ones."""
        with open(fasta_file_name) as f:
            lines = [l.strip() for l in f.readlines() if not l.startswith(">")]

        def per_read(kk, strand):
            hashes, start = hash_reads(lines, kk, strand)
            return [hashes[a:b] for a, b in zip(start[:-1], start[1:])]

        if small_k is None:
            return cls(itertools.chain(per_read(k, 0), per_read(k, 1)))
        parts = []
        for kk in (k, small_k, smallest_k):
            fwd, rev = per_read(kk, 0), per_read(kk, 1)
            parts.append([itertools.chain(a, b) for a, b in zip(fwd, rev)])
        return cls(zip(*parts))

    @classmethod
    def from_list_of_string_kmers(cls, string_kmers):
        return cls([[kmer_to_hash_fast(letter_sequence_to_numeric(km), len(km)) for km in read_kmers]
                    for read_kmers in string_kmers])

    @staticmethod
    def get_kmers_from_read(read, k):
        # read_kmers.py:51-57 (note the reference's range(len-k): the last k-mer is not produced)
        return [kmer_to_hash_fast(letter_sequence_to_numeric(read[i:i + k]), k) for i in range(len(read) - k)]

    @staticmethod
    def get_kmers_from_read_dynamic(read, power_vector):
        """read_kmers.py:67-70: np.convolve(codes, power_array(k), 'valid')."""
        k = len(power_vector)
        if not np.array_equal(np.asarray(power_vector, dtype=np.uint64), power_array(k)):
            raise ValueError("power_vector must be power_array(k)")
        if isinstance(read, np.ndarray):
            read = "".join("acgt"[int(c) & 3] for c in read) if read.dtype.kind in "iu" else "".join(read.tolist())
        return hash_reads([read], k, 0)[0]

    @staticmethod
    def get_kmers_from_read_dynamic_slow(read, k):
        """read_kmers.py:72-74: the reference raises before its rolling-hash loop runs; so does this (use
        `get_kmers_from_read_dynamic`)."""
        raise NotImplementedError()

    def __iter__(self):
        return self.kmers.__iter__()

    def __next__(self):
        return self.kmers.__next__()
