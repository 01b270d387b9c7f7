"""GPU parity tests proper: the HIP path, called through the C ABI (ctypes), against the CPU oracle on
the same seeded inputs and against the reference's golden vectors.  Bar: labels AND distances bit-exact
(the kernels and the oracle share one k-ordered fma arithmetic; see oracle/orc.h)."""
import numpy as np
import pytest

from helpers import assert_same_results, bitmap_from_ids, goldens, load_csv_fixtures
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

L2, IP = orc.METRIC_L2, orc.METRIC_INNER_PRODUCT


@pytest.fixture(scope="module")
def mf():
    import mi355_faiss

    assert mi355_faiss.device_count() >= 1
    return mi355_faiss


def _data(nb, nq, d, seed=0, dup=0, scale=1.0, center=False):
    rng = np.random.default_rng(seed)
    xb = rng.random((nb, d), dtype=np.float32) * scale
    xq = rng.random((nq, d), dtype=np.float32) * scale
    if center:
        xb -= 0.5 * scale
        xq -= 0.5 * scale
    if dup:
        xb[rng.integers(0, nb, dup)] = xb[rng.integers(0, nb, dup)]
    return xb, xq


# ------------------------------------------------------------------ reference goldens through the C ABI


def test_golden_flat_ip(mf):
    """test/sql/faiss.test:7-38 (BASELINE config C1 shape: Flat d=8 N=1000 nq=10)"""
    ids, xb, _, xq = load_csv_fixtures()
    ix = mf.index_factory(8, "Flat")
    for i0 in range(0, 1000, 300):
        ix.add(xb[i0 : i0 + 300])
    assert ix.ntotal == 1000
    D, I = ix.search(xq, 2)
    np.testing.assert_allclose(D.reshape(-1), goldens()["flat_ip_k2_distances"], rtol=1e-6)
    # and bit-exact against the oracle, also at BASELINE C1's k=10
    for k in (2, 10):
        o = orc.Index(8, "Flat")
        o.add(xb)
        Do, Io = o.search(xq, k)
        D, I = ix.search(xq, k)
        assert_same_results(D, I, Do, Io, False, what=f"C1 k={k}")


def test_golden_idmap_and_filter(mf):
    """test/sql/faiss2.test:17-42, faiss3.test:22-68"""
    ids, xb, _, xq = load_csv_fixtures()
    ix = mf.index_factory(8, "IDMap,Flat")
    ix.add_with_ids(xb, ids)
    D, I = ix.search(xq, 2)
    g = goldens()
    assert sorted(I.reshape(-1).tolist()) == g["idmap_flat_ip_k2_labels_multiset"]
    assert [int(I[q, j]) for q in range(10) for j in range(2)] == [r[1] for r in g["idmap_flat_ip_k2"]]
    np.testing.assert_allclose(D.reshape(-1), [r[2] for r in g["idmap_flat_ip_k2"]], rtol=1e-6)
    bm = bitmap_from_ids(ids, ids > 100)
    D, I = ix.search(xq, 2, sel=("bitmap", bm))
    assert [int(I[q, j]) for q in range(10) for j in range(2)] == [r[1] for r in g["idmap_flat_ip_k2_filter_id_gt_100"]]
    np.testing.assert_allclose(D.reshape(-1), [r[2] for r in g["idmap_flat_ip_k2_filter_id_gt_100"]], atol=1e-5)
    D2, I2 = ix.search(xq, 2, sel=("batch", ids[ids > 100]))
    assert np.array_equal(I, I2) and np.array_equal(D, D2)


def test_error_strings_and_lifecycle(mf):
    """faiss4.test:19-25, faiss6.test:27-33, faiss7.test:15-25"""
    ids, xb, _, _ = load_csv_fixtures()
    ix = mf.index_factory(8, "Flat", L2)
    with pytest.raises(mf.FaissException, match="add_with_ids not implemented for this type of index"):
        ix.add_with_ids(xb, ids)
    assert ix.ntotal == 0
    ix.add(xb)
    assert ix.ntotal == 1000
    with pytest.raises(mf.FaissException, match="could not parse index string"):
        mf.index_factory(8, "Bogus")
    with pytest.raises(mf.FaissException, match="k > 0"):
        ix.search(xb[:2], 0)
    small = mf.index_factory(2, "IDMap,Flat")
    small.add_with_ids(np.array([[0.0040321066, 0.023423655]], np.float32), np.array([231]))
    q = np.array([[-0.04529257, 0.024853613]], np.float32)
    D, I = small.search(q, 2)
    assert I.tolist() == [[231, -1]] and D[0, 1] == -np.finfo(np.float32).max
    D, I = small.search(q, 2, sel=("bitmap", bitmap_from_ids(np.array([231]), np.array([False]))))
    assert I.tolist() == [[-1, -1]]
    empty = mf.index_factory(4, "Flat", L2)
    D, I = empty.search(np.zeros((25, 4), np.float32), 3)
    assert (I == -1).all() and (D == np.finfo(np.float32).max).all()


# ------------------------------------------------------------------ seeded parity vs the oracle


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize(
    "nb,nq,d,k",
    [
        (1000, 10, 8, 10),  # BASELINE C1
        (5000, 64, 128, 10),  # MFMA path, resident queries, KC=128
        (4096, 64, 128, 10),
        (3001, 33, 100, 7),  # ragged rows / padded d
        (2000, 20, 16, 1),  # k = 1
        (9000, 130, 64, 32),  # two query blocks, k = nprobe-like
        (700, 21, 96, 5),
        (50, 40, 32, 60),  # k > N
        (6000, 25, 8, 3),
    ],
)
def test_flat_matches_oracle_bit_exact(mf, metric, nb, nq, d, k):
    xb, xq = _data(nb, nq, d, seed=nb + d, dup=nb // 20)
    ix = mf.index_factory(d, "Flat", metric)
    for i0 in range(0, nb, 2048):  # DataChunk-sized adds (STANDARD_VECTOR_SIZE)
        ix.add(xb[i0 : i0 + 2048])
    D, I = ix.search(xq, k)
    Do, Io = orc.flat_search(metric, xb, xq, k)
    # inner product included: boundary ties at the k-th score follow FAISS's CMin heap (tie pass)
    assert_same_results(D, I, Do, Io, metric == L2, what=f"flat m={metric} nb={nb} nq={nq} d={d} k={k}")


@pytest.mark.parametrize("staged", [0, 1])
@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("nq", [1, 5, 19])
def test_small_batches_take_the_per_pair_path(mf, metric, nq, staged):
    """nq < 20: FAISS computes sum((x-y)^2) directly (distance_compute_blas_threshold).  Two kernels implement the
    per-pair arithmetic: the packed-fp32 scan kernel (default) and the LDS-staged flat_direct kernel"""
    xb, xq = _data(7000, nq, 128, seed=nq, center=True)
    ix = mf.index_factory(128, "Flat", metric)
    ix.set_option("force_staged", staged)
    ix.add(xb)
    D, I = ix.search(xq, 10)
    Do, Io = orc.flat_search(metric, xb, xq, 10)
    assert_same_results(D, I, Do, Io, metric == L2, what=f"pair path nq={nq}")
    # 1-4 queries always stream through the LDS-staged kernel; inner-product batches of >= 8 ride the MFMA kernel
    # (fvec_inner_product is the same k-ordered chain)
    want = "flat_direct_kernel" if staged or nq <= 4 else "flat_pair_scan (ivf_scan_kernel)"
    if metric == IP and nq >= 8 and not staged:
        want = "flat_mfma_kernel"
    assert ix.last_kernel_info()["name"] == want


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("d", [16, 40, 100, 200])
def test_pair_scan_dims_and_selector(mf, metric, d):
    """per-pair scan kernel across row formats (dp = 16 / 64 / 128 pair-interleaved, 256 streaming format), ragged N,
    with and without a selector, nq on both sides of one 20-query group"""
    nb = 5003
    xb, xq = _data(nb, 45, d, seed=d, center=True)
    ix = mf.index_factory(d, "Flat", metric)
    ix.add(xb)
    keep = np.arange(nb)[np.arange(nb) % 3 != 1]
    Dg, Ig = ix.search(xq, 7, sel=("batch", keep))
    Do, Io = orc.flat_search(metric, xb, xq, 7, sel=("batch", keep))
    # L2 + selector = per-pair arithmetic (scan kernel); inner product + selector rides the fused MFMA kernel
    assert ix.last_kernel_info()["name"].startswith("flat_pair_scan" if metric == L2 else "flat_mfma_kernel")
    assert_same_results(Dg, Ig, Do, Io, metric == L2, what=f"pair scan + selector d={d}")
    D1, I1 = ix.search(xq[:7], 7)
    Do1, Io1 = orc.flat_search(metric, xb, xq[:7], 7)
    assert ix.last_kernel_info()["name"].startswith("flat_pair_scan")  # 5-7 queries: scan kernel for both metrics
    assert_same_results(D1, I1, Do1, Io1, metric == L2, what=f"pair scan nq=7 d={d}")


def test_l2_blas_vs_pair_arithmetic_differ_but_both_match(mf):
    """the two FAISS branches give different low-order bits; the device follows the same dispatch"""
    xb, xq = _data(3000, 20, 64, seed=77)
    ix = mf.index_factory(64, "Flat", L2)
    ix.add(xb)
    D20, I20 = ix.search(xq, 5)
    D19, I19 = ix.search(xq[:19], 5)
    Do20, Io20 = orc.flat_search(L2, xb, xq, 5)
    Do19, Io19 = orc.flat_search(L2, xb, xq[:19], 5)
    assert_same_results(D20, I20, Do20, Io20, True, what="nq=20")
    assert_same_results(D19, I19, Do19, Io19, True, what="nq=19")
    # float64 truth within north_star's 1e-4 relative
    S = ((xq[:, None, :].astype(np.float64) - xb[None].astype(np.float64)) ** 2).sum(-1)
    np.testing.assert_allclose(D20, np.sort(S, 1)[:, :5], rtol=1e-4)


@pytest.mark.parametrize("metric", [L2, IP])
def test_d768_streaming_geometry(mf, metric):
    """d > 128: 256-row tiles, k streamed in units of 32 (BASELINE C4/C5 dimension)"""
    xb, xq = _data(3000, 40, 768, seed=5, center=True)
    ix = mf.index_factory(768, "Flat", metric)
    ix.add(xb)
    D, I = ix.search(xq, 10)
    Do, Io = orc.flat_search(metric, xb, xq, 10)
    assert_same_results(D, I, Do, Io, metric == L2, what="d=768")


@pytest.mark.parametrize("metric", [L2, IP])
def test_selectors_with_large_batches(mf, metric):
    """selector forces the per-pair branch for any nq (utils/distances.cpp)"""
    xb, xq = _data(5000, 45, 32, seed=9)
    ids = (np.arange(5000, dtype=np.int64) * 7 + 3) % 100003
    ix = mf.index_factory(32, "IDMap,Flat", metric)
    ix.add_with_ids(xb, ids)
    o = orc.Index(32, "IDMap,Flat", metric)
    o.add_with_ids(xb, ids)
    keep = ids % 3 == 0
    for sel in (("bitmap", bitmap_from_ids(ids, keep)), ("batch", ids[keep])):
        D, I = ix.search(xq, 8, sel=sel)
        Do, Io = o.search(xq, 8, sel=sel)
        assert_same_results(D, I, Do, Io, metric == L2, what=f"selector {sel[0]}")
        assert np.isin(I, ids[keep]).all()


def test_large_k_goes_through_the_list_kernel(mf):
    """k beyond the fused kernel's LDS lists (the Go harness reaches k ~ 2000: main_test.go:26-32)"""
    xb, xq = _data(20000, 24, 64, seed=3)
    ix = mf.index_factory(64, "Flat", L2)
    ix.add(xb)
    D, I = ix.search(xq, 500)
    Do, Io = orc.flat_search(L2, xb, xq, 500)
    assert_same_results(D, I, Do, Io, True, what="k=500")


def test_synth_generator_matches_host(mf):
    import torch

    a = mf.synth_uniform_torch(1000, 128, 1234, row0=17).cpu().numpy()
    assert np.array_equal(a, orc.synth_uniform(1000, 128, 1234, row0=17))
    c = mf.synth_clustered_torch(500, 64, 99, row0=5, n_centers=16, sigma=0.1).cpu().numpy()
    assert np.array_equal(c, orc.synth_clustered(500, 64, 99, row0=5, n_centers=16, sigma=0.1))


def test_device_resident_api_and_label_offset(mf):
    import torch

    xb = mf.synth_uniform_torch(30000, 128, 1234)
    xq = mf.synth_uniform_torch(256, 128, 4321)
    ix = mf.index_factory(128, "Flat", L2)
    ix.add_torch(xb)
    ix.set_label_offset(1000000)
    D, I = ix.search_torch(xq, 10)
    torch.cuda.synchronize()
    Do, Io = orc.flat_search(L2, xb.cpu().numpy(), xq.cpu().numpy(), 10)
    assert_same_results(D.cpu().numpy(), I.cpu().numpy() - 1000000, Do, Io, True, what="device API")


def test_medium_size_properties(mf):
    """N = 1M (BASELINE C2's N), nq = 2048 (one DuckDB chunk): checked through size-independent properties --
    sorted output, self-query returns itself at distance ~0, sharded == unsharded, oracle on a query subsample."""
    import torch

    n, d, nq, k = 1_000_000, 128, 2048, 10
    xb = mf.synth_uniform_torch(n, d, 1234)
    xq = mf.synth_uniform_torch(nq, d, 4321)
    xq[:64] = xb[torch.arange(64, device=xb.device) * 15625]  # self queries
    ix = mf.index_factory(d, "Flat", L2)
    ix.add_torch(xb)
    D, I = ix.search_torch(xq, k)
    torch.cuda.synchronize()
    Dn, In = D.cpu().numpy(), I.cpu().numpy()
    assert (np.diff(Dn, axis=1) >= 0).all()
    assert (In[:64, 0] == np.arange(64) * 15625).all() and (Dn[:64, 0] <= 1e-4).all()
    assert (In >= 0).all() and (In < n).all()
    # sharded (4 row shards + host merge) == unsharded
    Ds, Is = [], []
    for s in range(4):
        sh = mf.index_factory(d, "Flat", L2)
        sh.add_torch(xb[s * 250000 : (s + 1) * 250000].contiguous())
        sh.set_label_offset(s * 250000)
        a, b = sh.search_torch(xq, k)
        Ds.append(a.cpu().numpy())
        Is.append(b.cpu().numpy())
    Dm, Im = mf.merge_shards(L2, np.stack(Ds), np.stack(Is))
    assert np.array_equal(Im, In) and np.array_equal(Dm, Dn)
    # oracle on a subsample of queries against the full database
    sub = np.arange(0, nq, 8)
    Do, Io = orc.flat_search(L2, xb.cpu().numpy(), xq[sub].cpu().numpy(), k, force_path=orc.PATH_BLAS)
    assert_same_results(Dn[sub], In[sub], Do, Io, True, what="1M subsample")


@pytest.mark.parametrize("idmap", [False, True])
@pytest.mark.parametrize("d,nb,nq,k", [(128, 20000, 64, 10), (24, 5003, 8, 3), (200, 9001, 33, 40), (768, 3000, 20, 10)])
def test_filtered_inner_product_runs_on_the_fused_kernel(mf, d, nb, nq, k, idmap):
    """inner product + IDSelector: FAISS's per-pair fvec_inner_product is the same k-ordered chain as the MFMA, so the
    filtered search (faiss_search_filter on the default metric) stays on the fused kernel and must still match the
    oracle's per-pair path bit for bit -- bitmap and batch selectors, through IDMap (external ids) or not"""
    xb, xq = _data(nb, nq, d, seed=nb + d, center=True)
    ids = (np.random.RandomState(d).permutation(3 * nb)[:nb] + 17).astype(np.int64) if idmap else np.arange(nb, dtype=np.int64)
    desc = "IDMap,Flat" if idmap else "Flat"
    ix, o = mf.index_factory(d, desc, IP), orc.Index(d, desc, IP)
    for a in (ix, o):
        a.add_with_ids(xb, ids) if idmap else a.add(xb)
    keep = ids[np.random.RandomState(nb).rand(nb) < 0.3]
    bm = bitmap_from_ids(ids, np.isin(ids, keep))
    for sel in (("batch", keep), ("bitmap", bm)):
        D, I = ix.search(xq, k, sel=sel)
        assert ix.last_kernel_info()["name"] == "flat_mfma_kernel"
        Do, Io = o.search(xq, k, sel=sel)
        assert np.all(np.isin(I[I >= 0], keep))
        assert_same_results(D, I, Do, Io, False, what=f"filtered IP on MFMA d={d} {sel[0]} idmap={idmap}")


def test_small_batch_large_k_on_the_fused_kernel(mf):
    """few query blocks make the planner split the rows many ways; the split count must still respect the merge
    kernel's LDS budget when k is large"""
    xb, xq = _data(300000, 24, 64, seed=5, center=True)
    for metric, k in ((L2, 80), (IP, 64)):
        ix = mf.index_factory(64, "Flat", metric)
        ix.set_option("prefilter", 0)  # (since round 4 the coarse filter serves k <= 128 at this size: keep the fused kernel under test)
        ix.add(xb)
        D, I = ix.search(xq, k)
        assert ix.last_kernel_info()["name"] == "flat_mfma_kernel"
        Do, Io = orc.flat_search(metric, xb, xq, k)
        assert_same_results(D, I, Do, Io, metric == L2, what=f"small batch large k m={metric}")
        ix.set_option("prefilter", -1)  # ... and the default route (24 queries, k = 64 / 80: the coarse filter) agrees
        D2, I2 = ix.search(xq, k)
        assert_same_results(D2, I2, Do, Io, metric == L2, what=f"small batch large k m={metric}, default route")


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("d,k", [(128, 13), (128, 100), (64, 200), (200, 150)])
def test_k_beyond_12_keeps_lists_in_global_memory(mf, metric, d, k):
    """k > 12: the per-query k-lists of the fused kernel live in the partial-result buffers (global memory) so that two
    workgroups still fit a CU; the answers must not change"""
    xb, xq = _data(40000, 150, d, seed=k, center=True)
    ix = mf.index_factory(d, "Flat", metric)
    ix.add(xb)
    D, I = ix.search(xq, k)
    assert ix.last_kernel_info()["name"] == "flat_mfma_kernel"
    Do, Io = orc.flat_search(metric, xb, xq, k)
    assert_same_results(D, I, Do, Io, metric == L2, what=f"global k-lists d={d} k={k} m={metric}")


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("desc", ["Flat", "IDMap,Flat", "IDMap,IVF4,Flat", "IDMap,HNSW8"])
def test_selectors_that_reject_everything(mf, metric, desc):
    """an empty id batch / an all-zero or zero-length bitmap: every slot comes back (-1, neutral) on every path"""
    d, n = 32, 3000
    xb, xq = _data(n, 30, d, seed=3, center=True)
    ids = np.arange(n, dtype=np.int64) + 1000
    ix = mf.index_factory(d, desc, metric)
    ix.train(xb)
    ix.add_with_ids(xb, ids) if desc.startswith("IDMap") else ix.add(xb)
    neutral = np.finfo(np.float32).max * (1 if metric == L2 else -1)
    for sel in (("batch", np.zeros(0, dtype=np.int64)), ("bitmap", np.zeros(1, dtype=np.uint8)), ("bitmap", np.zeros(0, dtype=np.uint8)),
                ("batch", np.array([10**9], dtype=np.int64))):  # nobody has this id
        for q in (xq, xq[:3]):
            D, I = ix.search(q, 4, sel=sel, nprobe=4, efSearch=32)
            assert np.all(I == -1), (desc, sel[0], len(q))
            assert np.all(D == neutral)


@pytest.mark.parametrize("metric", [L2, IP])
def test_result_does_not_depend_on_the_row_split_count(mf, metric):
    """The planner's split count (8...256, chosen for grid efficiency) is a pure performance knob: partial lists are
    merged with the exact (value, id) rule, so any forced count gives the default's result and the oracle's."""
    xb, xq = _data(300_000, 200, 128, seed=21, dup=500)
    ix = mf.index_factory(128, "Flat", metric)
    ix.set_option("prefilter", 0)  # this test is about the exact f32 kernel (the coarse filter would take a batch of this size)
    ix.add(xb)
    try:
        D0, I0 = ix.search(xq, 10)
        assert ix.last_kernel_info()["name"] == "flat_mfma_kernel"
        Do, Io = orc.flat_search(metric, xb, xq, 10)
        assert np.array_equal(D0, Do)
        assert np.array_equal(I0, Io)
        for ns in (1, 8, 40, 136, 256):
            ix.set_option("mfma_nsplit", ns)
            D, I = ix.search(xq, 10)
            assert ix.last_kernel_info()["nsplit"] == ns
            assert np.array_equal(D, D0) and np.array_equal(I, I0), ns
    finally:
        ix.set_option("mfma_nsplit", 0)


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("k", [10, 33, 100])
def test_heavy_ties_large_k_heap_lists(mf, metric, k):
    """The fused kernel's k-lists are heaps on (value, id): a database made of 40 distinct vectors repeated over and
    over (almost every comparison is an exact value tie decided by the id) must still give the oracle's result."""
    rng = np.random.default_rng(33)
    base = rng.random((40, 96), dtype=np.float32)
    xb = base[rng.integers(0, 40, 30_000)]
    xq = rng.random((64, 96), dtype=np.float32)
    ix = mf.index_factory(96, "Flat", metric)
    ix.add(xb)
    D, I = ix.search(xq, k)
    assert ix.last_kernel_info()["name"] == "flat_mfma_kernel"
    Do, Io = orc.flat_search(metric, xb, xq, k)
    assert np.array_equal(D, Do)
    # both metrics, every slot: for inner product almost every query has an exact tie at the k-th score, resolved by
    # the tie pass exactly as FAISS's CMin heap would (which rows of the tied run survive depends on arrival order)
    assert np.array_equal(I, Io)


@pytest.mark.parametrize("idmap", [False, True])
@pytest.mark.parametrize("nq", [3, 12, 19, 20, 150])
@pytest.mark.parametrize("d,nb,k", [(8, 4000, 10), (24, 30000, 3), (130, 9000, 17), (64, 200000, 10), (16, 50, 49)])
def test_inner_product_boundary_ties_follow_the_cmin_heap(mf, d, nb, k, nq, idmap):
    """Small-integer coordinates: scores are exact small integers, so nearly every query has MANY rows tied at its
    k-th score, rows above the boundary arriving before and after the tied ones.  FAISS's CMin heap (strict insert,
    root = smallest (score, id)) keeps an arrival-order dependent subset of the tied rows; every
    kernel that serves inner product -- per-pair scan, LDS-staged, fused MFMA, with and without a selector, through
    IDMap -- must return exactly that subset, in heap_reorder's order."""
    rs = np.random.RandomState(d * 1000 + nb + k + nq)
    xb = rs.randint(-2, 3, size=(nb, d)).astype(np.float32)
    xq = rs.randint(-2, 3, size=(nq, d)).astype(np.float32)
    ids = (rs.permutation(3 * nb)[:nb] + 5).astype(np.int64) if idmap else np.arange(nb, dtype=np.int64)
    desc = "IDMap,Flat" if idmap else "Flat"
    g, o = mf.index_factory(d, desc, IP), orc.Index(d, desc, IP)
    for a in (g, o):
        for i0 in range(0, nb, 2048):
            a.add_with_ids(xb[i0 : i0 + 2048], ids[i0 : i0 + 2048]) if idmap else a.add(xb[i0 : i0 + 2048])
    keep = ids[rs.rand(nb) < 0.6]
    for sel in (None, ("batch", keep), ("bitmap", bitmap_from_ids(ids, np.isin(ids, keep)))):
        D, I = g.search(xq, k, sel=sel)
        Do, Io = o.search(xq, k, sel=sel)
        Dn, _ = o.search(xq, min(k + 1, nb), sel=sel)
        if nb >= 1000 and sel is None and nq >= 12:
            assert (Dn[:, k - 1] == Dn[:, k]).mean() > 0.3  # the case under test really occurs
        assert_same_results(D, I, Do, Io, False, what=f"IP ties d={d} nb={nb} k={k} nq={nq} idmap={idmap} sel={sel and sel[0]}")


@pytest.mark.parametrize("idmap", [False, True])
@pytest.mark.parametrize("nq", [7, 40])
@pytest.mark.parametrize("d,nb,k", [(8, 6000, 100), (16, 20000, 250), (1, 5000, 128), (32, 9000, 1000)])
def test_inner_product_k_from_100_follows_the_reservoir(mf, d, nb, k, nq, idmap):
    """From k = 100 on FAISS collects results in a ReservoirTopN instead of a heap (utils/distances.cpp,
    distance_compute_min_k_reservoir; impl/ResultHandler.h, utils/partitioning.cpp partition_fuzzy_median3): which of the rows TIED
    at the k-th score survive depends on where the sampled thresholds fell while the stream went by.  Small-integer data puts
    many rows on every boundary; the device replays the reservoir for the queries it flags (csrc/flat_reservoir.hip) and must
    return the oracle's rows (oracle/orc_core.c reservoir_t) on every query -- fused kernel (k + 1 <= its k-lists) and
    flat_direct (beyond), both FAISS branches, selectors, IDMap.  d = 1: scores = a handful of integers, thousands of ties."""
    rs = np.random.RandomState(d * 77 + nb + k + nq)
    if d > 1:
        xb = rs.randint(-2, 3, size=(nb, d)).astype(np.float32)
        xq = rs.randint(-2, 3, size=(nq, d)).astype(np.float32)
    else:  # 1.5 k rows tied at the boundary, k / 2 better ones, k more tied, then worse ones: a shrink lands ON the boundary
        v = np.concatenate([np.full(k + k // 2, 5), np.full(k // 2, 9), np.full(k, 5), rs.randint(0, 5, size=nb - 3 * k)])
        xb = v.astype(np.float32).reshape(nb, 1)
        xq = np.ones((nq, 1), dtype=np.float32)
    ids = (rs.permutation(3 * nb)[:nb] + 5).astype(np.int64) if idmap else np.arange(nb, dtype=np.int64)
    desc = "IDMap,Flat" if idmap else "Flat"
    g, o = mf.index_factory(d, desc, IP), orc.Index(d, desc, IP)
    for a in (g, o):
        for i0 in range(0, nb, 2048):
            a.add_with_ids(xb[i0 : i0 + 2048], ids[i0 : i0 + 2048]) if idmap else a.add(xb[i0 : i0 + 2048])
    keep = ids[rs.rand(nb) < 0.7]
    differs_from_heap = 0
    for sel in (None, ("batch", keep)) if d > 1 else (None,):
        D, I = g.search(xq, k, sel=sel)
        Do, Io = o.search(xq, k, sel=sel)
        assert_same_results(D, I, Do, Io, False, what=f"IP reservoir d={d} nb={nb} k={k} nq={nq} idmap={idmap} sel={sel and sel[0]}")
        orc.set_reservoir(False)
        try:
            _, Ih = o.search(xq, k, sel=sel)
        finally:
            orc.set_reservoir(True)
        differs_from_heap += int((Ih != Io).any(1).sum())
    if d == 1:
        assert differs_from_heap > 0  # the reservoir's outcome really is not the heap's on such data


def test_l2_k_from_100_reservoir_equals_heap_on_the_device_too(mf):
    """L2: the reservoir keeps the k smallest (distance, id) like the heap -- duplicate-heavy rows, k = 100 ... 1500"""
    rs = np.random.RandomState(5)
    d, nb = 16, 12000
    xb = rs.randint(0, 4, size=(nb, d)).astype(np.float32)
    xq = rs.randint(0, 4, size=(33, d)).astype(np.float32)
    g, o = mf.index_factory(d, "Flat", L2), orc.Index(d, "Flat", L2)
    g.add(xb)
    o.add(xb)
    for k in (100, 257, 1500):
        assert_same_results(*g.search(xq, k), *o.search(xq, k), True, what=f"L2 k={k}")



def test_concurrent_ingest_into_several_indexes_shares_the_copy_helpers(mf):
    """round 6: the staging copy of a DataChunk-sized add runs on three threads (csrc/index.hip stage_copy_mt).  The two helpers belong
    to the library, not to an index: threads feeding DIFFERENT indexes at once either get them or copy alone -- every row of every index
    arrives where it belongs (each row finds itself at distance 0 under its own number)."""
    import threading

    d, n = 128, 60_000
    rs = np.random.RandomState(3)
    data = [rs.rand(n, d).astype(np.float32) + t for t in range(3)]  # (three clouds apart: a row copied into the wrong index would show)
    idx = [mf.index_factory(d, "Flat", L2) for _ in range(3)]
    errs = []

    def feed(t):
        try:
            for i0 in range(0, n, 2048):
                idx[t].add(data[t][i0 : i0 + 2048])
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=feed, args=(t,)) for t in range(3)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    for t in range(3):
        assert idx[t].ntotal == n
        probe = np.arange(0, n, 997)
        D, I = idx[t].search(data[t][probe], 1)
        assert np.array_equal(I[:, 0], probe), t
        assert float(D.max()) <= 1e-3, (t, float(D.max()))



@pytest.mark.parametrize("desc,d", [("Flat", 128), ("IDMap,Flat", 13), ("IDMap,Flat", 128), ("Flat", 770)])
def test_staged_adds_reach_every_reader(mf, tmp_path, desc, d):
    """round 6: DataChunk-sized add() calls (src/faiss_extension.cpp:510,512: <= 2048 rows each) are collected in a pinned
    slot and sent to the device once per 8 MB -- ntotal counts them at once, and search / write_index / clone / a growth of the row store
    in between all see every row.  Same results as an index that got its rows in one call and as one with the staging switched off."""
    rs = np.random.RandomState(d)
    n, nq, k = 30_000, 40, 5
    xb = rs.rand(n, d).astype(np.float32)
    xq = rs.rand(nq, d).astype(np.float32)
    ids = (np.arange(n, dtype=np.int64) * 7 + 3) if desc.startswith("IDMap") else None

    def feed(ix, lo, hi, chunk):
        for i0 in range(lo, hi, chunk):
            i1 = min(hi, i0 + chunk)
            if ids is None:
                ix.add(xb[i0:i1])
            else:
                ix.add_with_ids(xb[i0:i1], ids[i0:i1])

    whole = mf.index_factory(d, desc, L2)
    feed(whole, 0, n, n)
    eager = mf.index_factory(d, desc, L2)
    eager.set_option("lazy_adds", 0)
    lazy = mf.index_factory(d, desc, L2)
    checkpoints = [1, 2048 + 77, 9_000, 20_001, n]
    lo = 0
    for hi in checkpoints:
        feed(lazy, lo, hi, 777 if hi < 20_000 else 2048)  # (odd chunks: unaligned staging offsets at d = 13)
        feed(eager, lo, hi, 2048)
        lo = hi
        assert lazy.ntotal == hi
        kk = min(k, hi)
        a, b = lazy.search(xq, kk), eager.search(xq, kk)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), hi
    a, b = lazy.search(xq, k), whole.search(xq, k)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
    # a write with rows still staged: the file holds them all
    more = rs.rand(300, d).astype(np.float32)
    if ids is None:
        lazy.add(more)
        whole.add(more)
    else:
        mid = np.arange(300, dtype=np.int64) + 10_000_000
        lazy.add_with_ids(more, mid)
        whole.add_with_ids(more, mid)
    path = str(tmp_path / "staged.index")
    mf.write_index(lazy, path)
    back = mf.read_index(path)
    assert back.ntotal == n + 300
    a, b = back.search(xq, k), whole.search(xq, k)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
