"""The metrics beyond L2 / inner product that the glue registers (src/faiss_extension.cpp:58-68 maps the names L1, Linf,
Lp, Canberra, BrayCurtis, JensenShannon, Jaccard onto faiss::MetricType): IndexFlat::search -> knn_extra_metrics
[faiss/utils/extra_distances.cpp], restated in oracle/orc_core.c extra_distance().

Bar: labels and distances bit-identical for the metrics built from +, -, |.|, min, max and IEEE division (L1, Linf,
Canberra, BrayCurtis, Jaccard, and Lp with the exponent the glue leaves at its default 0).  Lp with a real exponent and
JensenShannon go through powf / logf, whose last bits differ between glibc and the device math library: distances
within 2e-6 relative, labels equal wherever the oracle's neighbouring distances are further apart than that."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

L1, LINF, LP, CANBERRA, BRAYCURTIS, JS, JACCARD = 2, 3, 4, 20, 21, 22, 23
EXACT = [L1, LINF, CANBERRA, BRAYCURTIS, JACCARD]


@pytest.fixture(scope="module")
def mf():
    import mi355_faiss

    return mi355_faiss


def _data(seed, n, nq, d):
    rs = np.random.RandomState(seed)
    # strictly positive rows: Canberra / JensenShannon / Jaccard are defined there (0/0 and log(0) otherwise, in FAISS too)
    xb = (rs.rand(n, d) * 0.95 + 0.05).astype(np.float32)
    xq = (rs.rand(nq, d) * 0.95 + 0.05).astype(np.float32)
    return xb, xq


def _pair(mf, d, desc, metric):
    return mf.index_factory(d, desc, metric), orc.Index(d, desc, metric)


@pytest.mark.parametrize("metric", EXACT)
@pytest.mark.parametrize("d,n,nq,k", [(7, 300, 1, 1), (8, 2049, 5, 10), (16, 1000, 33, 4), (32, 5000, 21, 16),
                                      (100, 3000, 19, 40), (128, 4097, 70, 10), (130, 777, 3, 100)])
def test_exact_metrics_bitwise(mf, metric, d, n, nq, k):
    xb, xq = _data(100 * metric + d, n, nq, d)
    g, o = _pair(mf, d, "Flat", metric)
    g.add(xb)
    o.add(xb)
    D, I = g.search(xq, k)
    Do, Io = o.search(xq, k)
    assert np.array_equal(I, Io)
    assert np.array_equal(D.view(np.uint32), Do.view(np.uint32))


@pytest.mark.parametrize("metric", EXACT)
def test_signed_rows_l1_linf_braycurtis(mf, metric):
    if metric in (CANBERRA, JACCARD):
        pytest.skip("defined for non-negative data")
    rs = np.random.RandomState(metric)
    xb = (rs.rand(4000, 48).astype(np.float32) - 0.5) * 8
    xq = (rs.rand(25, 48).astype(np.float32) - 0.5) * 8
    xb[2000] = xb[3]  # an exact duplicate: equal distances, the smaller id first
    g, o = _pair(mf, 48, "Flat", metric)
    g.add(xb)
    o.add(xb)
    D, I = g.search(xq, 12)
    Do, Io = o.search(xq, 12)
    assert np.array_equal(I, Io) and np.array_equal(D.view(np.uint32), Do.view(np.uint32))


def test_lp_with_the_glue_default_exponent(mf):
    """metric_arg stays 0 behind the glue: powf(|x-y|, 0) = 1, every distance is d and the first k rows win the ties"""
    xb, xq = _data(4, 1000, 7, 24)
    g, o = _pair(mf, 24, "Flat", LP)
    g.add(xb)
    o.add(xb)
    D, I = g.search(xq, 5)
    Do, Io = o.search(xq, 5)
    assert np.array_equal(I, Io) and np.array_equal(D, Do)
    assert np.all(D == 24.0)


def _close_with_gaps(D, I, Do, Io, rtol):
    assert np.allclose(D, Do, rtol=rtol, atol=0)
    gap_next = np.abs(np.diff(Do, axis=1)) > 4 * rtol * np.abs(Do[:, 1:])
    safe = np.ones_like(Io, dtype=bool)
    safe[:, 1:] &= gap_next
    safe[:, :-1] &= gap_next
    safe[:, -1] = False  # the (k+1)-th distance is not in the list: its gap is unknown
    assert safe.mean() > 0.5
    assert np.array_equal(I[safe], Io[safe])


@pytest.mark.parametrize("d,n,nq,k", [(8, 2000, 5, 10), (40, 3000, 33, 8), (128, 2500, 20, 10)])
def test_lp_exponent_3(mf, d, n, nq, k):
    xb, xq = _data(11 + d, n, nq, d)
    g, o = _pair(mf, d, "Flat", LP)
    g.set_option("metric_arg_bits", int(np.float32(3.0).view(np.uint32)))
    orc.set_metric_arg(3.0)
    try:
        g.add(xb)
        o.add(xb)
        D, I = g.search(xq, k)
        Do, Io = o.search(xq, k)
    finally:
        orc.set_metric_arg(0.0)
    _close_with_gaps(D, I, Do, Io, 2e-6)


@pytest.mark.parametrize("d,n,nq,k", [(8, 2000, 5, 10), (40, 3000, 33, 8), (100, 2500, 20, 10)])
def test_jensen_shannon(mf, d, n, nq, k):
    xb, xq = _data(31 + d, n, nq, d)
    xb /= xb.sum(1, keepdims=True)  # distributions
    xq /= xq.sum(1, keepdims=True)
    g, o = _pair(mf, d, "Flat", JS)
    g.add(xb)
    o.add(xb)
    D, I = g.search(xq, k)
    Do, Io = o.search(xq, k)
    # the summands -x log(m/x) - y log(m/y) cancel to ~1e-3 of their size: an ulp of logf shows up 1e3 times larger
    _close_with_gaps(D, I, Do, Io, 2e-4)


@pytest.mark.parametrize("metric", [L1, JACCARD])
def test_idmap_selector_and_roundtrip(mf, metric, tmp_path):
    d, n = 20, 3000
    xb, xq = _data(5 * metric, n, 9, d)
    ids = (np.random.RandomState(3).permutation(4 * n)[:n] + 7).astype(np.int64)
    g, o = _pair(mf, d, "IDMap,Flat", metric)
    g.add_with_ids(xb, ids)
    o.add_with_ids(xb, ids)
    keep = ids[::3].copy()
    for sel in (None, ("batch", keep)):
        D, I = g.search(xq, 10, sel=sel)
        Do, Io = o.search(xq, 10, sel=sel)
        assert np.array_equal(I, Io) and np.array_equal(D.view(np.uint32), Do.view(np.uint32))
    path = str(tmp_path / "m.index")
    mf.write_index(g, path)
    g2 = mf.read_index(path)
    assert g2.metric_type == metric
    D2, I2 = g2.search(xq, 10)
    Do, Io = o.search(xq, 10)
    assert np.array_equal(I2, Io) and np.array_equal(D2.view(np.uint32), Do.view(np.uint32))


def test_large_batch_and_many_splits(mf):
    """enough rows for several row splits and enough queries for several 20-query groups"""
    d, n, nq, k = 64, 60000, 130, 10
    xb, xq = _data(77, n, nq, d)
    for metric in (L1, BRAYCURTIS):
        g, o = _pair(mf, d, "Flat", metric)
        g.add(xb)
        o.add(xb)
        D, I = g.search(xq, k)
        Do, Io = o.search(xq, k)
        assert np.array_equal(I, Io) and np.array_equal(D.view(np.uint32), Do.view(np.uint32))


def test_ivf_and_hnsw_reject_the_extra_metrics(mf):
    for desc in ("IVF16,Flat", "HNSW16"):
        with pytest.raises(mf.FaissException):
            mf.index_factory(16, desc, L1)


@pytest.mark.parametrize("metric", [L1, JACCARD])
def test_row_shards(mf, metric):
    """the in-library row sharding (csrc/sharded.hip) merges the shards' lists under the metric's order"""
    d, n, k = 24, 20000, 10
    xb, xq = _data(13 * metric, n, 40, d)
    sh, o = _pair(mf, d, "Flat", metric)
    sh.shard_to_gpus([0, 0, 0])
    for i0 in range(0, n, 2048):
        sh.add(xb[i0 : i0 + 2048])
        o.add(xb[i0 : i0 + 2048])
    assert sum(sh.shard_info()["rows_per_shard"]) == n
    D, I = sh.search(xq, k)
    Do, Io = o.search(xq, k)
    assert np.array_equal(I, Io) and np.array_equal(D.view(np.uint32), Do.view(np.uint32))
