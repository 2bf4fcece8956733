import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

REFERENCE = "/root/reference"
HAVE_REFERENCE = os.path.isdir(os.path.join(REFERENCE, "graph_kmer_index"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: drives the read-only reference at /root/reference "
                                       "(build container only; skipped elsewhere)")


def pytest_collection_modifyitems(config, items):
    import pytest
    skip_ref = pytest.mark.skip(reason="/root/reference not present")
    for item in items:
        if "reference" in item.keywords and not HAVE_REFERENCE:
            item.add_marker(skip_ref)
