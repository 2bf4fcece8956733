"""FlatKmers / FlatKmers2 record containers with the reference's attributes and .npz layout
(flat_kmers.py:7-154).  Columns are NumPy arrays owned by Python; the device-resident variant used
between GPU stages is `DeviceFlatKmers`."""
import logging
import numpy as np

from . import _lib
from .graph import encode_letters


class FlatKmers2:
    def __init__(self, hashes, start_nodes, start_offsets, nodes, allele_frequencies):
        assert len(hashes) == len(nodes)
        assert len(start_nodes) == len(nodes)
        assert len(start_offsets) == len(start_nodes)
        self._hashes = hashes
        self._nodes = nodes
        self._start_nodes = start_nodes
        self._start_offsets = start_offsets
        if allele_frequencies is None:
            self._allele_frequencies = np.zeros(len(self._hashes), dtype=np.single) + 1.0
        else:
            self._allele_frequencies = allele_frequencies

    def __str__(self):
        return '\n'.join(str(data) for data in [self._hashes, self._nodes])

    __repr__ = __str__


class FlatKmers:
    def __init__(self, hashes, nodes, ref_offsets=None, allele_frequencies=None):
        assert len(hashes) == len(nodes)
        self._hashes = hashes
        self._nodes = nodes
        self._ref_offsets = np.zeros(len(self._nodes)) if ref_offsets is None else ref_offsets
        if allele_frequencies is None:
            self._allele_frequencies = np.zeros(len(self._hashes), dtype=np.single) + 1.0
        else:
            self._allele_frequencies = allele_frequencies

    def describtion(self):
        return "".join("%d: %d\n" % (kmer, node) for kmer, node in zip(self._hashes, self._nodes))

    @classmethod
    def from_file(cls, file_name):
        try:
            data = np.load(file_name)
        except FileNotFoundError:
            data = np.load(file_name + ".npz")
        return cls(data["hashes"], data["nodes"], data["ref_offsets"], data["allele_frequencies"])

    def to_file(self, file_name):
        np.savez(file_name, hashes=self._hashes, nodes=self._nodes, ref_offsets=self._ref_offsets,
                 allele_frequencies=self._allele_frequencies)
        logging.info("Save to %s.npz" % file_name)

    @classmethod
    def from_multiple_flat_kmers(cls, flat_kmers_list):
        """Concatenate and cast to uint64/uint32/uint64/float32 (flat_kmers.py:71-90)."""
        flats = list(flat_kmers_list)
        hashes = np.concatenate([np.asarray(f._hashes) for f in flats]).astype(np.uint64) if flats \
            else np.zeros(0, np.uint64)
        nodes = np.concatenate([np.asarray(f._nodes) for f in flats]).astype(np.uint32) if flats \
            else np.zeros(0, np.uint32)
        refs = [np.asarray(f._ref_offsets) for f in flats if f._ref_offsets is not None]
        ref_offsets = np.concatenate(refs).astype(np.uint64) if refs and sum(len(r) for r in refs) else None
        af = np.concatenate([np.asarray(f._allele_frequencies) for f in flats]).astype(np.single) if flats \
            else np.zeros(0, np.single)
        return FlatKmers(hashes, nodes, ref_offsets, af)

    def _frequencies_in(self, index):
        """get_frequency(kmer) of every hash (flat_kmers.py:92-96): batched on an index that offers it."""
        if hasattr(index, "get_frequencies"):
            return [int(x) for x in index.get_frequencies(np.asarray(self._hashes))]
        return [index.get_frequency(int(kmer)) for kmer in self._hashes]

    def sum_of_kmer_frequencies(self, kmer_index_with_frequencies):
        return sum([0] + [max(1, f) for f in self._frequencies_in(kmer_index_with_frequencies)])

    def maximum_kmer_frequency(self, kmer_index_with_frequencies):
        return max([0] + self._frequencies_in(kmer_index_with_frequencies))

    def get_new_without_singletons(self):
        """Keep the 2nd and later occurrences of every hash, original order (flat_kmers.py:98-125)."""
        h = np.asarray(self._hashes)
        order = np.argsort(h, kind="stable")
        first = np.ones(len(h), dtype=bool)
        first[1:] = h[order][1:] != h[order][:-1]
        keep = np.ones(len(h), dtype=bool)
        keep[order[first]] = False
        return FlatKmers(h[keep], np.asarray(self._nodes)[keep], np.asarray(self._ref_offsets)[keep],
                         np.asarray(self._allele_frequencies)[keep])

    def get_reverse_complement_flat_kmers(self, k):
        from .kmer_hashing import kmer_hashes_to_reverse_complement_hash
        return FlatKmers(kmer_hashes_to_reverse_complement_hash(self._hashes, k), self._nodes, self._ref_offsets,
                         self._allele_frequencies)


class DeviceFlatKmers:
    """FlatKmers columns resident in HBM, in the merged layout of flat_kmers.py:90
    (hashes uint64, nodes uint32, ref_offsets uint64, allele_frequencies float32)."""

    def __init__(self, n, hashes, nodes, ref_offsets, allele_frequencies):
        self.n = int(n)
        self.hashes, self.nodes, self.ref_offsets, self.allele_frequencies = hashes, nodes, ref_offsets, allele_frequencies

    @classmethod
    def allocate(cls, n):
        n_alloc = max(int(n), 1)
        return cls(n, _lib.DeviceArray(n_alloc, np.uint64), _lib.DeviceArray(n_alloc, np.uint32),
                   _lib.DeviceArray(n_alloc, np.uint64), _lib.DeviceArray(n_alloc, np.float32))

    @classmethod
    def from_flat_kmers(cls, flat):
        n = len(flat._hashes)
        return cls(n, _lib.DeviceArray.from_host(np.asarray(flat._hashes).astype(np.uint64)),
                   _lib.DeviceArray.from_host(np.asarray(flat._nodes).astype(np.uint32)),
                   _lib.DeviceArray.from_host(np.asarray(flat._ref_offsets).astype(np.uint64)),
                   _lib.DeviceArray.from_host(np.asarray(flat._allele_frequencies).astype(np.float32)))

    def get_reverse_complement_flat_kmers(self, k):
        """flat_kmers.py:127-131 on the device: reverse-complemented hashes, the other columns shared (views)."""
        out = _lib.DeviceArray(max(self.n, 1), np.uint64)
        _lib.check(_lib.load().gki_reverse_complement(self.hashes.ptr, self.n, int(k), out.ptr))
        return DeviceFlatKmers(self.n, out, self.nodes.view(0, self.n), self.ref_offsets.view(0, self.n),
                               self.allele_frequencies.view(0, self.n))

    @classmethod
    def from_multiple_flat_kmers(cls, parts):
        """flat_kmers.py:71-90: concatenation, in HBM."""
        total = sum(p.n for p in parts)
        out = cls.allocate(total)
        at = 0
        lib = _lib.load()
        for p in parts:
            for src, dst in ((p.hashes, out.hashes), (p.nodes, out.nodes), (p.ref_offsets, out.ref_offsets),
                             (p.allele_frequencies, out.allele_frequencies)):
                if p.n:
                    _lib.check(lib.gki_memcpy_d2d(dst.view(at, p.n).ptr, src.ptr, p.n * src.dtype.itemsize))
            at += p.n
        return out

    def get_new_without_singletons(self):
        """flat_kmers.py:98-125 on the device: the 2nd and later occurrences of every hash, original order."""
        flags = _lib.DeviceArray(max(self.n, 1), np.uint8)
        _lib.check(_lib.load().gki_flag_repeated_kmers(self.hashes.ptr, self.n, flags.ptr))
        out = self.compacted(flags)
        flags.free()
        return out

    def compacted(self, flags):
        """Records whose flag (DeviceArray uint8, one per record) is set, order kept (gki_compact_flat)."""
        import ctypes as C
        keep, _ = flags.checksum(self.n)
        out = DeviceFlatKmers.allocate(keep)
        n_out = C.c_int64(0)
        _lib.check(_lib.load().gki_compact_flat(flags.ptr, self.n, self.hashes.ptr, self.nodes.ptr, self.ref_offsets.ptr,
                                                self.allele_frequencies.ptr, out.hashes.ptr, out.nodes.ptr,
                                                out.ref_offsets.ptr, out.allele_frequencies.ptr, max(keep, 1),
                                                C.byref(n_out)))
        assert n_out.value == keep
        return out

    def to_flat_kmers(self):
        return FlatKmers(self.hashes.to_host(self.n), self.nodes.to_host(self.n), self.ref_offsets.to_host(self.n),
                         self.allele_frequencies.to_host(self.n))

    def free(self):
        for a in (self.hashes, self.nodes, self.ref_offsets, self.allele_frequencies):
            a.free()


def letter_sequence_to_numeric(sequence):
    """a/n/m (and anything else) -> 0, c -> 1, g -> 2, t -> 3, case-insensitive, uint64
    (flat_kmers.py:134-145)."""
    if isinstance(sequence, np.ndarray):
        if sequence.dtype.kind in "US":
            sequence = "".join(str(x) for x in sequence.tolist())
        else:
            return sequence.astype(np.uint64)
    return encode_letters(sequence).astype(np.uint64)


def numeric_to_letter_sequence(sequence):
    lut = np.array(["a", "c", "g", "t"], dtype=object)
    return lut[np.asarray(sequence).astype(np.int64) & 3]
