"""graph_kmer_index hot path on MI355X (gfx950): k-mer enumeration over a variation graph, 2-bit
hashing, FlatKmers, CollisionFreeKmerIndex build and batched lookup.

Public names mirror /root/reference/graph_kmer_index/__init__.py:1-12 for the in-scope path.
Host code is Python/NumPy over a ctypes C ABI (include/gki.h) to hand-written HIP kernels; there
is no CPU fallback -- without libgki_hip.so and a GPU the operations raise.
"""
from .flat_kmers import letter_sequence_to_numeric, numeric_to_letter_sequence  # noqa: F401
from .kmer_hashing import kmer_to_hash_fast, sequence_to_kmer_hash, kmer_hash_to_sequence  # noqa: F401
from .flat_kmers import FlatKmers, FlatKmers2, DeviceFlatKmers  # noqa: F401
from .graph import GraphArrays  # noqa: F401
from .device_graph import DeviceGraph  # noqa: F401
from .critical_graph_paths import CriticalGraphPaths  # noqa: F401
from .kmer_finder import DenseKmerFinder  # noqa: F401
from .collision_free_kmer_index import CollisionFreeKmerIndex  # noqa: F401
from .collision_free_kmer_index import CollisionFreeKmerIndex as KmerIndex  # noqa: F401
from .collision_free_kmer_index import CounterKmerIndex  # noqa: F401
from .reverse_kmer_index import ReverseKmerIndex  # noqa: F401
from .read_kmers import ReadKmers  # noqa: F401
from .nplist import NpList  # noqa: F401

__all__ = ["letter_sequence_to_numeric", "numeric_to_letter_sequence", "kmer_to_hash_fast",
           "sequence_to_kmer_hash", "kmer_hash_to_sequence", "FlatKmers", "FlatKmers2", "DeviceFlatKmers",
           "GraphArrays", "DeviceGraph", "CriticalGraphPaths", "DenseKmerFinder", "CollisionFreeKmerIndex", "KmerIndex", "CounterKmerIndex", "ReverseKmerIndex", "ReadKmers", "NpList"]
