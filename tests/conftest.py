import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    from oracle import ref_harness
    if ref_harness.available():
        return
    skip = pytest.mark.skip(reason="/root/reference not present (GPU box)")
    for item in items:
        if "reference" in item.keywords:
            item.add_marker(skip)
