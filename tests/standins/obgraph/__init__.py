"""TEST-ONLY stand-in for the third-party `obgraph` package (absent from this image).

It exists so that the read-only reference at /root/reference can be imported in
THIS container to generate golden vectors (tests/golden/make_golden.py) and to
cross-check the oracle.  It is not part of the product and never travels into
`graph_kmer_index_amd`.  Only the accessors the reference's hot path calls are
provided (SURVEY.md section 8b lists them).
"""
from .graph import Graph, VariantNotFoundException  # noqa: F401
