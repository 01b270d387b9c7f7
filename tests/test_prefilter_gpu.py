"""bf16x3 prefilter + exact f32 re-scoring (csrc/flat_bf16.hip) behind IndexFlat::search (src/faiss_extension.cpp:631):
the answers must be those of the exact f32 kernel and of the oracle's BLAS branch BIT FOR BIT -- labels and distances --
on friendly data, on duplicate-heavy data (where the proof fails and queries fall back to the exact kernel), on
non-finite input, for both metrics, with inner-product boundary ties."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
L2, IP = orc.METRIC_L2, orc.METRIC_INNER_PRODUCT


@pytest.fixture(scope="module")
def mf():
    import mi355_faiss

    return mi355_faiss


def _pair(mf, d, metric, xb, desc="Flat", ids=None):
    pf, ex = mf.index_factory(d, desc, metric), mf.index_factory(d, desc, metric)
    pf.set_option("prefilter", 1)
    ex.set_option("prefilter", 0)
    for ix in (pf, ex):
        for i0 in range(0, len(xb), 1 << 16):
            if ids is None:
                ix.add(xb[i0 : i0 + (1 << 16)])
            else:
                ix.add_with_ids(xb[i0 : i0 + (1 << 16)], ids[i0 : i0 + (1 << 16)])
    return pf, ex


def _check(pf, ex, xq, k, metric, xb=None, oracle_rows=0):
    D1, I1 = pf.search(xq, k)
    assert pf.last_kernel_info()["name"] == "flat_bf16x3_kernel"
    D0, I0 = ex.search(xq, k)
    assert ex.last_kernel_info()["name"] == "flat_mfma_kernel"
    assert np.array_equal(I1, I0), "labels differ from the exact f32 kernel"
    assert np.array_equal(D1.view(np.uint32), D0.view(np.uint32)), "distances differ from the exact f32 kernel"
    if oracle_rows:
        Do, Io = orc.flat_search(metric, xb, xq[:oracle_rows], k, force_path=orc.PATH_BLAS)
        assert np.array_equal(I1[:oracle_rows], Io) and np.array_equal(D1[:oracle_rows].view(np.uint32), Do.view(np.uint32))
    return D1, I1


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("d,nb,nq,k", [(128, 200_000, 700, 10), (64, 150_000, 300, 5), (100, 99_991, 257, 1), (128, 70_000, 1000, 32), (40, 50_000, 64, 10)])
def test_prefilter_equals_exact_kernel_and_oracle(mf, metric, d, nb, nq, k):
    rs = np.random.RandomState(d + nb)
    xb = rs.rand(nb, d).astype(np.float32) - (0.5 if metric == IP else 0.0)
    xq = rs.rand(nq, d).astype(np.float32) - (0.5 if metric == IP else 0.0)
    pf, ex = _pair(mf, d, metric, xb)
    _check(pf, ex, xq, k, metric, xb, oracle_rows=64)
    st = pf.prefilter_stats()
    assert st["queries"] == nq and st["fallback_queries"] <= nq // 10, st  # friendly data: the proof almost always holds
    assert 0 < st["max_rel_err"] < st["err_bound"] / 5, st  # the bound has >= 5x room over what the device really does


@pytest.mark.parametrize("metric", [L2, IP])
def test_duplicates_and_ties_fall_back_and_stay_exact(mf, metric):
    """40 distinct vectors repeated 100k times: no candidate margin can be proven for any query, so every query is re-run on
    the exact kernel; mixed data: only the queries that land on duplicated rows fall back"""
    rs = np.random.RandomState(5)
    base = rs.rand(40, 128).astype(np.float32)
    xb = base[rs.randint(0, 40, 100_000)]
    xq = rs.rand(300, 128).astype(np.float32)
    pf, ex = _pair(mf, 128, metric, xb)
    _check(pf, ex, xq, 10, metric, xb, oracle_rows=32)
    assert pf.prefilter_stats()["fallback_queries"] == len(xq)  # nothing provable: everything re-run, still exact
    xb2 = rs.rand(120_000, 128).astype(np.float32)
    xb2[rs.randint(0, 120_000, 30_000)] = xb2[rs.randint(0, 120_000, 30_000)]
    xq2 = np.concatenate([rs.rand(200, 128).astype(np.float32), xb2[rs.randint(0, 120_000, 200)]])
    pf, ex = _pair(mf, 128, metric, xb2)
    _check(pf, ex, xq2, 10, metric, xb2, oracle_rows=400)


def test_idmap_and_integer_inner_product_ties(mf):
    rs = np.random.RandomState(9)
    xb = rs.randint(-3, 4, size=(80_000, 64)).astype(np.float32)
    xq = rs.randint(-3, 4, size=(256, 64)).astype(np.float32)
    ids = (rs.permutation(400_000)[:80_000] + 11).astype(np.int64)
    pf, ex = _pair(mf, 64, IP, xb, desc="IDMap,Flat", ids=ids)
    D, I = _check(pf, ex, xq, 10, IP)
    o = orc.Index(64, "IDMap,Flat", IP)
    o.add_with_ids(xb, ids)
    Do, Io = o.search(xq, 10)
    assert np.array_equal(I, Io) and np.array_equal(D, Do)


@pytest.mark.parametrize("metric", [L2, IP])
def test_non_finite_and_huge_values(mf, metric):
    rs = np.random.RandomState(2)
    xb = rs.rand(60_000, 128).astype(np.float32)
    xb[100, 5] = np.nan
    xb[200, 7] = np.inf
    xb[300] *= 1e18
    xb[400] *= 1e-20
    xq = rs.rand(128, 128).astype(np.float32)
    xq[3, 1] = np.nan
    xq[4] *= 1e19
    pf, ex = _pair(mf, 128, metric, xb)
    D1, I1 = pf.search(xq, 10)
    D0, I0 = ex.search(xq, 10)
    assert np.array_equal(I1, I0) and np.array_equal(D1.view(np.uint32), D0.view(np.uint32))


def test_error_model_is_an_upper_bound_with_room(mf):
    """the device's bf16x3 inner product against the exact chain: measured on real kernel output (approximate values are
    not exposed, so compare through the proof's own quantity: every query of a friendly batch must be provable, and
    on 200k x 128 uniform rows the observed candidate margin must dwarf the bound)"""
    rs = np.random.RandomState(77)
    xb = rs.rand(200_000, 128).astype(np.float32)
    xq = rs.rand(2048, 128).astype(np.float32)
    pf, ex = _pair(mf, 128, L2, xb)
    _check(pf, ex, xq, 10, L2)
    # float64 truth vs the returned (exact-chain) distances: 1e-4 relative, north_star's tolerance
    D, I = pf.search(xq[:64], 10)
    truth = ((xb[I].astype(np.float64) - xq[:64, None, :].astype(np.float64)) ** 2).sum(-1)
    np.testing.assert_allclose(D, truth, rtol=1e-4)


def test_selector_searches_never_take_the_prefilter(mf):
    """inner product + IDSelector: the bf16x3 prefilter has no selector instances -- such a search rides the coarse filter
    (since round 3 also at d = 64: rows zero-padded in the 128-dim bf16 store) or, with the coarse filter off, the fused f32
    kernel's SEL instances (FAISS's per-pair branch); the selector is honoured on every route"""
    rs = np.random.RandomState(4)
    xb = rs.rand(300_000, 64).astype(np.float32) - 0.5
    xq = rs.rand(600, 64).astype(np.float32) - 0.5
    ix = mf.index_factory(64, "Flat", IP)
    ix.add(xb)
    keep = np.arange(300_000)[rs.rand(300_000) < 0.3]
    Do, Io = orc.flat_search(IP, xb, xq[:64], 10, sel=("batch", keep))
    D, I = ix.search(xq, 10, sel=("batch", keep))
    assert ix.last_kernel_info()["name"] == "flat_bf16_collect_kernel"
    assert np.isin(I, keep).all()
    assert np.array_equal(I[:64], Io) and np.array_equal(D[:64], Do)
    ix.set_option("prefilter", 1)  # the bf16x3 prefilter wherever ITS kernel serves the shape: never under a selector
    D1, I1 = ix.search(xq, 10, sel=("batch", keep))
    assert ix.last_kernel_info()["name"] == "flat_mfma_kernel"
    assert np.array_equal(I1, I) and np.array_equal(D1, D)
    D2, I2 = ix.search(xq, 10)  # and without the selector the same index does take it
    assert ix.last_kernel_info()["name"] == "flat_bf16x3_kernel"


@pytest.mark.parametrize("seed", range(16))
def test_prefilter_fuzz(mf, seed):
    """seeded differential fuzz with the prefilter FORCED at small, ragged shapes: random d in (32, 128], N, nq, k, metric,
    IDMap, duplicates, scaled / shifted data -- always the exact kernel's answer, bit for bit, and the oracle's"""
    rs = np.random.RandomState(7000 + seed)
    d = int(rs.choice([33, 40, 64, 65, 96, 100, 127, 128]))
    n = int(rs.choice([4096, 5000, 8191, 20000, 33333]))
    nq = int(rs.choice([20, 21, 64, 255, 257, 300]))
    k = int(rs.choice([1, 2, 7, 10, 11, 16, 27, 39]))
    metric = [L2, IP][rs.randint(2)]
    idmap = bool(rs.randint(2))
    scale = float(rs.choice([1.0, 1e-3, 300.0]))
    xb = ((rs.rand(n, d).astype(np.float32) - (0.5 if rs.randint(2) else 0.0)) * scale).astype(np.float32)
    xq = ((rs.rand(nq, d).astype(np.float32) - 0.5) * scale).astype(np.float32)
    if rs.randint(2):
        xb[rs.randint(0, n, n // 10)] = xb[rs.randint(0, n, n // 10)]  # duplicates: ties, proof failures
        xq[: nq // 4] = xb[rs.randint(0, n, nq // 4)]
    ids = (rs.permutation(3 * n)[:n] + 5).astype(np.int64) if idmap else None
    desc = "IDMap,Flat" if idmap else "Flat"
    pf, ex = _pair(mf, d, metric, xb, desc=desc, ids=ids)
    D1, I1 = pf.search(xq, k)
    assert pf.last_kernel_info()["name"] == "flat_bf16x3_kernel"
    D0, I0 = ex.search(xq, k)
    what = f"seed={seed} d={d} n={n} nq={nq} k={k} metric={metric} idmap={idmap} scale={scale}"
    assert np.array_equal(I1, I0), what
    assert np.array_equal(D1.view(np.uint32), D0.view(np.uint32)), what
    o = orc.Index(d, desc, metric)
    o.add_with_ids(xb, ids) if idmap else o.add(xb)
    Do, Io = o.search(xq, k)
    assert np.array_equal(I1, Io) and np.array_equal(D1.view(np.uint32), Do.view(np.uint32)), what
