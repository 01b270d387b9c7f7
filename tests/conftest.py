import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "duckdb-faiss-ext_amd", "pyhost"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU: skip rather than crash inside HIP
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
