"""oracle/orc_core.c search_openblas: FAISS's BLAS branch (utils/distances.cpp exhaustive_*_blas: 4096 x 1024 sgemm blocks, norms
formula, heaps) on the REAL OpenBLAS sgemm -- the library the reference links (/root/reference/CMakeLists.txt:78-90,
vcpkg_ports/openblas/vcpkg.json:3 pins 0.3.29; numpy's wheel bundles that version).  It is the independent reference for label
stability: its summation order is OpenBLAS's, not the k-ordered chain the oracle shares with the device."""
import numpy as np
import pytest

from oracle import oracle as orc

L2, IP = orc.METRIC_L2, orc.METRIC_INNER_PRODUCT

pytestmark = pytest.mark.skipif(orc.openblas_path() is None, reason="no libscipy_openblas64_ under numpy.libs")


def test_loads_the_pinned_version():
    cfg = orc.openblas_load()
    assert "OpenBLAS 0.3.29" in cfg and "USE64BITINT" in cfg, cfg


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("d,n,nq", [(8, 1000, 25), (128, 30000, 300), (100, 5000, 64), (768, 6000, 40)])
def test_matches_float64_truth_and_the_port(metric, d, n, nq):
    orc.openblas_load()
    rng = np.random.default_rng(d + n)
    xb = (rng.random((n, d), dtype=np.float32) - 0.3).astype(np.float32)
    xq = (rng.random((nq, d), dtype=np.float32) - 0.3).astype(np.float32)
    k = 10
    Db, Ib = orc.flat_search(metric, xb, xq, k, force_path=orc.PATH_OPENBLAS)
    Dp, Ip = orc.flat_search(metric, xb, xq, k, force_path=orc.PATH_BLAS)
    # values: both are f32 evaluations of the same formula -> 1e-5 relative of each other, 1e-4 of float64 truth
    x64, y64 = xq.astype(np.float64), xb.astype(np.float64)
    for q in range(nq):
        rows = y64[Ib[q]]
        truth = ((rows - x64[q]) ** 2).sum(1) if metric == L2 else rows @ x64[q]
        np.testing.assert_allclose(Db[q], truth, rtol=1e-4, atol=1e-4)
    # labels: the port's result against OpenBLAS's -- every differing slot inside the rounding band
    cen = orc.openblas_census(metric, xb, xq, k, Dp, Ip)
    assert cen["differing_slots_inside_band"] == cen["slots_label_differs"], cen
    assert cen["slots_label_differs"] <= cen["fragile_adjacent_pairs"], cen


def test_blocking_and_order_contract():
    """more than one 4096-query block and more than one 1024-row block; FAISS's output order; k > N padding"""
    orc.openblas_load()
    rng = np.random.default_rng(7)
    xb = rng.random((2500, 16), dtype=np.float32)
    xq = rng.random((4100, 16), dtype=np.float32)
    D, I = orc.flat_search(L2, xb, xq, 5, force_path=orc.PATH_OPENBLAS)
    assert (np.diff(D, axis=1) >= 0).all() and I.min() >= 0 and I.max() < 2500
    Dn, In = orc.flat_search_naive(L2, xb[:300], xq[:30], 5, orc.PATH_BLAS)
    Ds, Is = orc.flat_search(L2, xb[:300], xq[:30], 5, force_path=orc.PATH_OPENBLAS)
    np.testing.assert_allclose(Ds, Dn, rtol=1e-5, atol=1e-6)
    Dk, Ik = orc.flat_search(IP, xb[:7], xq[:20], 10, force_path=orc.PATH_OPENBLAS)
    assert (Ik[:, 7:] == -1).all() and (Dk[:, 7:] == -np.finfo(np.float32).max).all()


def test_census_flags_a_wrong_label():
    """a slot whose label is outside the rounding band must NOT be excused"""
    orc.openblas_load()
    rng = np.random.default_rng(3)
    xb = rng.random((4000, 32), dtype=np.float32)
    xq = rng.random((40, 32), dtype=np.float32)
    D, I = orc.flat_search(L2, xb, xq, 10, force_path=orc.PATH_BLAS)
    I = I.copy()
    far = int(np.argmax(((xb - xq[0]) ** 2).sum(1)))
    I[0, 0] = far
    cen = orc.openblas_census(L2, xb, xq, 10, D, I)
    assert cen["slots_label_differs"] >= 1 and cen["differing_slots_inside_band"] < cen["slots_label_differs"], cen
