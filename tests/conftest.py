import os
import sys

import pytest

# The oracle (oracle/liborc.so) is OpenMP code and takes every hardware thread it is given: on a GPU box with 256 of them and other
# tenants' load the fork/join of its many small parallel regions alone cost seconds per call (k-means on 12 000 x 32 rows: 4.6 s with
# 256 threads, 0.003 s with 32 -- profiles/r6_oracle_threads.txt; tests/test_ivf_gpu.py took 541 s on such a box, a quarter of that on an
# idle one).  Unless the caller says otherwise the suite runs it on 32.  (Set before any test module loads the library.)
os.environ.setdefault("OMP_NUM_THREADS", "32")

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
