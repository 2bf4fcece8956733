"""Helpers shared by the -m gpu parity tests."""
import numpy as np
from golden_cases import canonical_order


def finder_cols(finder):
    fl = finder.get_flat_kmers()           # v="2"
    return dict(kmers=fl._hashes, nodes=fl._nodes, start_nodes=fl._start_nodes, start_offsets=fl._start_offsets,
                allele_frequencies=fl._allele_frequencies)


def assert_same_records(got, exp, exact_order=False):
    keys = ("kmers", "nodes", "start_nodes", "start_offsets", "allele_frequencies")
    for k in keys:
        assert len(got[k]) == len(exp[k]), "%s: %d records vs %d expected" % (k, len(got[k]), len(exp[k]))
    if not exact_order:
        og, oe = canonical_order(got), canonical_order(exp)
    for k in keys:
        a, b = np.asarray(got[k]), np.asarray(exp[k])
        if not exact_order:
            a, b = a[og], b[oe]
        assert np.array_equal(a, b), "column %s differs" % k
