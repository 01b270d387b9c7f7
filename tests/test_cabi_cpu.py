"""CPU-side checks of the drop-in boundary: libmi355faiss.so loads, exports every symbol that
include/mi355_faiss.h declares, fails loudly without a GPU, and the host k-way merge
(the step after the RCCL all-gather) equals the oracle's merge."""
import os
import re

import numpy as np
import pytest

from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mi355_faiss.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mvs_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import ctypes

    import mi355_faiss as mf

    names = _declared_functions()
    assert len(names) >= 25
    L = ctypes.CDLL(mf.LIB_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(mf.DECLARED_SYMBOLS) == names
    assert "mi355" in mf.lib().mvs_version().decode()


def test_header_cites_reference_call_sites():
    src = open(HEADER).read()
    for cite in ("faiss_extension.cpp:154", "faiss_extension.cpp:631", "faiss_extension.cpp:510", "gpu.cpp:48"):
        assert cite in src


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful on a box without a GPU")
def test_fails_loudly_without_gpu():
    import mi355_faiss as mf

    with pytest.raises(mf.FaissException, match="no CPU fallback"):
        mf.index_factory(8, "Flat")


@pytest.mark.parametrize("metric", [orc.METRIC_L2, orc.METRIC_INNER_PRODUCT])
def test_host_merge_equals_oracle_merge(metric):
    import mi355_faiss as mf

    rng = np.random.default_rng(4)
    xb = rng.random((6000, 24), dtype=np.float32)
    xb[rng.integers(0, 6000, 500)] = xb[rng.integers(0, 6000, 500)]  # duplicates -> ties
    xq = rng.random((50, 24), dtype=np.float32)
    parts = np.array_split(np.arange(6000), 5)
    Ds, Is = [], []
    for p in parts:
        D, I = orc.flat_search(metric, xb[p], xq, 10, force_path=orc.PATH_BLAS)
        Ds.append(D)
        Is.append(np.where(I >= 0, I + p[0], -1))
    Dm, Im = mf.merge_shards(metric, np.stack(Ds), np.stack(Is))
    Do, Io = orc.merge_shards(metric, np.stack(Ds), np.stack(Is))
    assert np.array_equal(Dm, Do) and np.array_equal(Im, Io)
    if metric == orc.METRIC_L2:
        Dr, Ir = orc.flat_search(metric, xb, xq, 10)
        assert np.array_equal(Im, Ir) and np.array_equal(Dm, Dr)


def test_host_merge_pads_short_shards():
    import mi355_faiss as mf

    D = np.array([[[1.0, np.finfo(np.float32).max]], [[0.5, 2.0]]], np.float32)
    I = np.array([[[7, -1]], [[3, 9]]], np.int64)
    Dm, Im = mf.merge_shards(mf.METRIC_L2, D, I)
    assert Im.tolist() == [[3, 7]] and Dm.tolist() == [[0.5, 1.0]]
    Dm, Im = mf.merge_shards(mf.METRIC_L2, D[:1], I[:1])
    assert Im.tolist() == [[7, -1]]
