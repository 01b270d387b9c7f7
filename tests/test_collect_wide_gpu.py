"""bf16 coarse filter for 128 < d <= 1024 (csrc/flat_collect_wide.hip, option prefilter = 2) behind IndexFlat::search
(src/faiss_extension.cpp:631): same contract as tests/test_collect_gpu.py -- labels and distances BIT FOR BIT those of the
exact f32 kernel and of the oracle's BLAS branch -- at the store widths 256 / 384 / 512 / 768 / 1024 (the last two: k split over a wave pair), with ragged dimensions, both
metrics, through IDMap, with duplicates, offset data, non-finite queries and small (per-pair branch) batches."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
L2, IP = orc.METRIC_L2, orc.METRIC_INNER_PRODUCT
KERNEL = ("flat_bf16_wide_kernel", "flat_bf16_big_kernel")  # the family: the big kernel serves the 768 / 1024 / 1536-dim stores


@pytest.fixture(scope="module")
def mf():
    import mi355_faiss

    return mi355_faiss


def _pair(mf, d, metric, xb, desc="Flat", ids=None):
    cl, ex = mf.index_factory(d, desc, metric), mf.index_factory(d, desc, metric)
    cl.set_option("prefilter", 2)
    ex.set_option("prefilter", 0)
    for ix in (cl, ex):
        for i0 in range(0, len(xb), 1 << 15):
            if ids is None:
                ix.add(xb[i0 : i0 + (1 << 15)])
            else:
                ix.add_with_ids(xb[i0 : i0 + (1 << 15)], ids[i0 : i0 + (1 << 15)])
    return cl, ex


def _check(cl, ex, xq, k, metric, xb=None, oracle_rows=0, path=orc.PATH_BLAS, kernel=KERNEL):
    D1, I1 = cl.search(xq, k)
    assert cl.last_kernel_info()["name"] in ((kernel,) if isinstance(kernel, str) else kernel), cl.last_kernel_info()
    D0, I0 = ex.search(xq, k)
    assert ex.last_kernel_info()["name"] not in KERNEL
    assert np.array_equal(I1, I0), "labels differ from the exact f32 kernel"
    assert np.array_equal(D1.view(np.uint32), D0.view(np.uint32)), "distances differ from the exact f32 kernel"
    if oracle_rows:
        Do, Io = orc.flat_search(metric, xb, xq[:oracle_rows], k, force_path=path)
        assert np.array_equal(I1[:oracle_rows], Io) and np.array_equal(D1[:oracle_rows].view(np.uint32), Do.view(np.uint32))
    return D1, I1


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("d,nb,nq,k", [(256, 60_000, 600, 10), (192, 40_001, 257, 1), (129, 30_000, 300, 5), (384, 50_000, 333, 10),
                                       (300, 35_000, 130, 15), (512, 40_000, 260, 10), (400, 20_000, 64, 3),
                                       (768, 40_000, 300, 10), (600, 25_000, 65, 15), (513, 20_000, 200, 1), (700, 30_000, 1000, 4),
                                       (1024, 30_000, 300, 10), (900, 20_000, 129, 16), (769, 20_000, 40, 2),
                                       # 1024 < d <= 1536: two 768-dim parts per row on the one-wave-per-SIMD kernel
                                       (1536, 30_000, 300, 10), (1100, 20_000, 140, 5), (1025, 16_000, 33, 16), (1536, 12_345, 1000, 1),
                                       # 16 < k <= 32 (31 for inner product: k + 1 with the tie detection): 32 row classes per query
                                       (256, 50_000, 300, 20), (384, 30_000, 200, 31), (512, 30_000, 150, 24), (768, 40_000, 300, 20),
                                       (1024, 25_000, 100, 31), (1536, 20_000, 120, 20)])
def test_wide_collect_equals_exact_kernel_and_oracle(mf, metric, d, nb, nq, k):
    rs = np.random.RandomState(d + nb)
    xb = rs.rand(nb, d).astype(np.float32) - (0.5 if metric == IP else 0.0)
    xq = rs.rand(nq, d).astype(np.float32) - (0.5 if metric == IP else 0.0)
    cl, ex = _pair(mf, d, metric, xb)
    _check(cl, ex, xq, k, metric, xb, oracle_rows=48)  # (inner product at k = 16 searches 17: 32 row classes)
    st = cl.collect_stats()
    assert st["queries"] == nq and st["overflows"] == 0 and st["candidates"] >= nq * k, st
    assert cl.prefilter_stats()["fallback_queries"] == 0
    # the filter does filter: far fewer candidates than rows
    assert st["candidates"] < nq * nb * 0.2, st


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("d,nb,nq,k", [(256, 60_000, 300, 64), (384, 50_000, 200, 100), (512, 40_000, 150, 33), (768, 50_000, 300, 100),
                                       (1024, 30_000, 100, 127), (1536, 25_000, 120, 50), (700, 30_000, 40, 128), (768, 40_000, 19, 100)])
def test_wide_k_up_to_128_on_four_subsets_of_32_classes(mf, metric, d, nb, nq, k):
    """32 < k <= 128 at 128 < d <= 1536 (round 6; VERDICT r5 missing #3: these fell to the f32 kernel, ~10x slower at batch sizes
    that fill the matrix pipe): 128 class slots per query (row & 127) in four subsets of 32, as tests/test_collect_gpu.py's d <= 128
    case -- the bound is the WORST of the subsets' ceil(k / 4)-th best class values.  Same answers as the exact f32 kernel and the
    oracle (k >= 100: FAISS's reservoir; inner product searches k + 1 for the tie detection, so k = 128 stays on the exact kernels there)."""
    rs = np.random.RandomState(k * 1000 + d)
    xb = rs.rand(nb, d).astype(np.float32) - (0.5 if metric == IP else 0.0)
    xq = rs.rand(nq, d).astype(np.float32) - (0.5 if metric == IP else 0.0)
    xb[::53] = xb[11]  # duplicates: ties inside the result lists
    cl, ex = _pair(mf, d, metric, xb)
    if metric == IP and k + 1 > 128:
        D, I = cl.search(xq, k)
        assert cl.last_kernel_info()["name"] not in KERNEL
        De, Ie = ex.search(xq, k)
        assert np.array_equal(I, Ie) and np.array_equal(D.view(np.uint32), De.view(np.uint32))
        return
    _check(cl, ex, xq, k, metric, xb, oracle_rows=24, path=orc.PATH_BLAS if nq >= 20 else orc.PATH_PAIR)
    st = cl.collect_stats()
    assert st["queries"] == nq and st["overflows"] == 0 and st["candidates"] >= nq * k, st
    assert cl.prefilter_stats()["fallback_queries"] == 0
    assert st["candidates"] < nq * nb * 0.3, st


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("d,nb,nq,k", [(256, 270_000, 200, 129), (384, 270_000, 100, 300), (768, 265_000, 150, 1000), (1536, 263_000, 40, 200),
                                       (512, 265_000, 12, 400)])
def test_wide_lists_beyond_128_entries(mf, metric, d, nb, nq, k):
    """round 6: k > 128 on the wide stores -- bounds from row ranges' class slots, the scan against them frozen, segmented-sort
    selection (tests/test_collect_gpu.py::test_lists_beyond_128_entries_stay_on_the_coarse_filter); same answers as the exact kernels
    and the oracle"""
    rs = np.random.RandomState(k + d)
    xb = rs.rand(nb, d).astype(np.float32) - (0.5 if metric == IP else 0.0)
    xq = rs.rand(nq, d).astype(np.float32) - (0.5 if metric == IP else 0.0)
    xb[::53] = xb[11]
    cl, ex = _pair(mf, d, metric, xb)
    D1, I1 = cl.search(xq, k)
    assert cl.last_kernel_info()["name"] in KERNEL, cl.last_kernel_info()
    D0, I0 = ex.search(xq, k)
    assert ex.last_kernel_info()["name"] not in KERNEL
    assert np.array_equal(I1, I0), "labels differ from the exact kernels"
    assert np.array_equal(D1.view(np.uint32), D0.view(np.uint32)), "distances differ from the exact kernels"
    Do, Io = orc.flat_search(metric, xb, xq[:3], k, force_path=orc.PATH_BLAS if nq >= 20 else orc.PATH_PAIR)
    assert np.array_equal(I1[:3], Io) and np.array_equal(D1[:3].view(np.uint32), Do.view(np.uint32))


@pytest.mark.parametrize("metric", [L2, IP])
def test_wide_k_100_with_selector_and_idmap(mf, metric):
    rs = np.random.RandomState(78)
    d, nb, k = 768, 40_000, 100
    xb = rs.randint(-2, 3, size=(nb, d)).astype(np.float32)  # integer rows: exact ties everywhere, also at the k-th value
    xq = rs.randint(-2, 3, size=(64, d)).astype(np.float32)
    ids = (rs.permutation(3 * nb)[:nb] + 3).astype(np.int64)
    g, o = mf.index_factory(d, "IDMap,Flat", metric), orc.Index(d, "IDMap,Flat", metric)
    g.set_option("prefilter", 2)
    for a in (g, o):
        a.add_with_ids(xb, ids)
    keep = ids[rs.rand(nb) < 0.5]
    for sel in (None, ("batch", keep)):
        D, I = g.search(xq, k, sel=sel)
        assert g.last_kernel_info()["name"] in KERNEL
        Do, Io = o.search(xq, k, sel=sel)
        assert np.array_equal(D.view(np.uint32), Do.view(np.uint32)), sel and sel[0]
        assert np.array_equal(I, Io), sel and sel[0]


@pytest.mark.parametrize("d", [384, 768, 1024, 1536])
@pytest.mark.parametrize("metric", [L2, IP])
def test_wide_normalised_embeddings_and_added_rows(mf, metric, d):
    """unit vectors (the C4 shape at a width the kernel serves), rows added after the first search (the store grows, the
    centre stays), lists up to the filter's 16"""
    rs = np.random.RandomState(11)
    xb = rs.randn(45_000, d).astype(np.float32)
    xb /= np.linalg.norm(xb, axis=1, keepdims=True)
    xq = rs.randn(400, d).astype(np.float32)
    xq /= np.linalg.norm(xq, axis=1, keepdims=True)
    cl, ex = _pair(mf, d, metric, xb[:30_000])
    _check(cl, ex, xq, 10, metric, xb[:30_000], oracle_rows=32)
    cl.add(xb[30_000:])
    ex.add(xb[30_000:])
    _check(cl, ex, xq, 16 if metric == L2 else 15, metric, xb, oracle_rows=32)  # (inner product keeps one rank for its tie detection)


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("d", [256, 768])
def test_wide_duplicates_offsets_and_idmap(mf, metric, d):
    rs = np.random.RandomState(3)
    xb = (rs.rand(50_000, d) * 2.0 + 3.0).astype(np.float32)  # far from the origin: the centring keeps the bound tight
    xb[rs.randint(0, 50_000, 8_000)] = xb[rs.randint(0, 50_000, 8_000)]
    xq = np.concatenate([(rs.rand(150, d) * 2.0 + 3.0).astype(np.float32), xb[rs.randint(0, 50_000, 150)]])
    ids = rs.permutation(1 << 20)[:50_000].astype(np.int64)
    cl, ex = _pair(mf, d, metric, xb, desc="IDMap,Flat", ids=ids)
    D1, I1 = _check(cl, ex, xq, 10, metric)
    Do, Io = orc.flat_search(metric, xb, xq[:64], 10, force_path=orc.PATH_BLAS)
    assert np.array_equal(I1[:64], ids[Io]) and np.array_equal(D1[:64].view(np.uint32), Do.view(np.uint32))


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("d", [320, 640])
def test_wide_non_finite_queries_go_to_the_exact_kernel(mf, metric, d):
    rs = np.random.RandomState(8)
    xb = rs.rand(30_000, d).astype(np.float32)
    xq = rs.rand(200, d).astype(np.float32)
    xq[5, 7] = np.inf
    xq[17, 300] = np.nan
    xq[99] *= 1e30
    cl, ex = _pair(mf, d, metric, xb)
    D1, I1 = cl.search(xq, 5)
    D0, I0 = ex.search(xq, 5)
    assert cl.last_kernel_info()["name"] in KERNEL
    assert cl.prefilter_stats()["fallback_queries"] >= 3
    assert np.array_equal(I1, I0) and np.array_equal(D1.view(np.uint32), D0.view(np.uint32))


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("d", [256, 768])
def test_wide_small_batch_uses_the_per_pair_arithmetic(mf, metric, d):
    """fewer than 20 queries: FAISS's per-pair branch (L2 = sum (x - y)^2); the candidates are re-scored in that arithmetic"""
    rs = np.random.RandomState(21)
    xb = rs.rand(40_000, d).astype(np.float32)
    xq = rs.rand(7, d).astype(np.float32)
    cl, ex = _pair(mf, d, metric, xb)
    D1, I1 = cl.search(xq, 10)
    assert cl.last_kernel_info()["name"] in KERNEL
    D0, I0 = ex.search(xq, 10)
    assert np.array_equal(I1, I0) and np.array_equal(D1.view(np.uint32), D0.view(np.uint32))
    Do, Io = orc.flat_search(metric, xb, xq, 10)  # the oracle picks the branch by the batch size, as FAISS does
    assert np.array_equal(I1, Io) and np.array_equal(D1.view(np.uint32), Do.view(np.uint32))


def test_wide_is_not_used_where_it_has_no_instance(mf):
    """d > 1536 stays on the exact kernels (prefilter = 2 only forces the filter where it exists)"""
    rs = np.random.RandomState(2)
    xb = rs.rand(20_000, 1600).astype(np.float32)
    ix = mf.index_factory(1600, "Flat", L2)
    ix.set_option("prefilter", 2)
    ix.add(xb)
    D, I = ix.search(xb[:40], 3)
    assert ix.last_kernel_info()["name"] == "flat_mfma_kernel" and np.array_equal(I[:, 0], np.arange(40))


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("d,idmap,frac", [(256, False, 0.5), (384, True, 0.05), (768, False, 0.3), (640, True, 0.9), (512, False, 0.01), (1024, True, 0.2),
                                          (1536, True, 0.3)])
def test_wide_selector_searches(mf, metric, d, idmap, frac):
    """IDSelectorBitmap / IDSelectorBatch in front of the wide kernels: one selector bit per row, rejected rows are neither
    candidates nor evidence for the bound; FAISS's per-pair arithmetic under a selector (tests/test_collect_gpu.py)"""
    rs = np.random.RandomState(d + int(frac * 100))
    nb, nq, k = 40_000, 150, 8
    xb = rs.rand(nb, d).astype(np.float32) - (0.5 if metric == IP else 0.0)
    xq = rs.rand(nq, d).astype(np.float32) - (0.5 if metric == IP else 0.0)
    ids = (rs.permutation(1 << 18)[:nb] + 5).astype(np.int64) if idmap else None
    universe = ids if idmap else np.arange(nb, dtype=np.int64)
    keep = np.sort(universe[rs.rand(nb) < frac])
    if d % 128 == 0:
        bm = np.zeros((int(universe.max()) + 8) // 8, dtype=np.uint8)
        np.bitwise_or.at(bm, keep >> 3, (1 << (keep & 7)).astype(np.uint8))
        sel = ("bitmap", bm)
    else:
        sel = ("batch", keep)
    cl, ex = _pair(mf, d, metric, xb, desc="IDMap,Flat" if idmap else "Flat", ids=ids)
    D1, I1 = cl.search(xq, k, sel=sel)
    assert cl.last_kernel_info()["name"] in KERNEL, cl.last_kernel_info()
    D0, I0 = ex.search(xq, k, sel=sel)
    assert ex.last_kernel_info()["name"] not in KERNEL
    assert np.array_equal(I1, I0) and np.array_equal(D1.view(np.uint32), D0.view(np.uint32))
    o = orc.Index(d, "IDMap,Flat" if idmap else "Flat", metric)
    if idmap:
        o.add_with_ids(xb, ids)
    else:
        o.add(xb)
    Do, Io = o.search(xq[:48], k, sel=sel)
    assert np.array_equal(I1[:48], Io) and np.array_equal(D1[:48].view(np.uint32), Do.view(np.uint32))
    assert np.isin(I1[I1 >= 0], keep).all()


@pytest.mark.parametrize("d", [256, 768, 1000, 1536])
def test_wide_inner_product_boundary_ties(mf, d):
    """integer data: many rows share the k-th score; FAISS's heap outcome from the candidate list (tests/test_collect_gpu.py)"""
    rs = np.random.RandomState(d)
    xb = rs.randint(-1, 2, size=(40_000, d)).astype(np.float32)
    xq = rs.randint(-1, 2, size=(150, d)).astype(np.float32)
    cl, ex = _pair(mf, d, IP, xb)
    D1, I1 = _check(cl, ex, xq, 6, IP)
    o = orc.Index(d, "Flat", IP)
    o.add(xb)
    Do, Io = o.search(xq, 6)
    assert np.array_equal(I1, Io) and np.array_equal(D1, Do)
