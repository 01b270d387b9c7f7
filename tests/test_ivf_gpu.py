"""IVFFlat on device vs the CPU oracle's restatement of IndexIVFFlat (no golden values exist in the reference for IVF
results: parity unpinned by the reference).  Training assigns on the fused MFMA kernel and updates centroids in
FAISS's summation order, so centroids are expected to be bit-identical to the oracle's; list scans use the per-pair
arithmetic (IVFFlatScanner) and must match bit for bit, exact distance ties included (arrival order = probe rank, then list
position; csrc/ivf_ties.hip)."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
L2, IP = orc.METRIC_L2, orc.METRIC_INNER_PRODUCT


@pytest.fixture(scope="module")
def mf():
    import mi355_faiss

    return mi355_faiss


def _clustered(n, d, seed, ncent=64, sigma=0.15):
    return orc.synth_clustered(n, d, seed, n_centers=ncent, sigma=sigma)


def _no_tie_rows(D):
    """queries whose k results have pairwise distinct distances -- only the two opt-in L2 modes that evaluate the norms formula
    instead of the scanner's arithmetic (ivf_mfma = 1 / 2) still need it; every default path reproduces FAISS's heap under
    exact ties (csrc/ivf_ties.hip)"""
    return np.array([len(np.unique(r)) == len(r) for r in D])


@pytest.mark.parametrize("metric", [L2, IP])
def test_kmeans_centroids_bit_identical_to_oracle(mf, metric):
    xb = _clustered(6000, 32, 7)
    o = orc.Index(32, "IVF16,Flat", metric)
    o.train(xb)
    g = mf.index_factory(32, "IVF16,Flat", metric)
    assert not g.is_trained and g.kind == mf.KIND_IVFFLAT and g.nlist == 16
    g.train(xb)
    assert g.is_trained
    assert g.quantizer.ntotal == 16
    assert np.array_equal(g.ivf_centroids().view(np.uint32), o.ivf_centroids().view(np.uint32))


def test_subsampled_training_matches_oracle(mf):
    """nx > 256*nlist: rand_perm(mt19937) subsample (Clustering.cpp subsample_training_set)"""
    xb = _clustered(3000, 16, 3, ncent=8)
    o = orc.Index(16, "IVF8,Flat", L2)
    o.train(xb)  # 3000 > 8*256 = 2048
    g = mf.index_factory(16, "IVF8,Flat", L2)
    g.train(xb)
    assert np.array_equal(g.ivf_centroids(), o.ivf_centroids())


@pytest.mark.parametrize("fast_scan", [1, 0])
@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("nprobe", [1, 4, 16])
def test_ivf_search_matches_oracle(mf, metric, nprobe, fast_scan):
    d, nlist = 64, 16
    xb = _clustered(20000, d, 11)
    xq = _clustered(300, d, 12)
    o = orc.Index(d, f"IVF{nlist},Flat", metric)
    o.train(xb)
    o.add(xb)
    g = mf.index_factory(d, f"IVF{nlist},Flat", metric)
    g.ivf_set_centroids(o.ivf_centroids())  # share the centroids: this test is about add + search
    g.set_option("ivf_fast_scan", fast_scan)  # 1: csrc/ivf_scan.hip, 0: the LDS-staged flat_direct item kernel
    g.set_option("ivf_mfma", 0)  # inner product would default to the MFMA variant (tested separately below)
    for i0 in range(0, 20000, 7000):
        g.add(xb[i0 : i0 + 7000])
    assert g.ntotal == 20000
    for k in (10, 40):  # 40: threshold classes wider than one 16-slot row
        Do, Io = o.search(xq, k, nprobe=nprobe)
        D, I = g.search(xq, k, nprobe=nprobe)
        ok = np.ones(len(Do), dtype=bool)  # exact ties included (csrc/ivf_ties.hip)
        assert ok.sum() > 250
        assert np.array_equal(I[ok], Io[ok])
        assert np.array_equal(D[ok].view(np.uint32), Do[ok].view(np.uint32))
    assert g.last_kernel_info()["name"].startswith("ivf_scan_kernel" if fast_scan else "ivf_list_scan")


def test_ivf_train_add_search_end_to_end_and_recall(mf):
    """BASELINE config C3's shape in miniature: IVF64,Flat on a Gaussian mixture, recall@10 vs exact Flat"""
    d, n, nq = 128, 60000, 500
    xb = _clustered(n, d, 21, ncent=256, sigma=0.1)
    xq = _clustered(nq, d, 22, ncent=256, sigma=0.1)
    g = mf.index_factory(d, "IVF64,Flat", L2)
    g.train(xb)
    g.add(xb)
    # ground truth in the same per-pair arithmetic the list scan uses (the BLAS-branch formula ranks near-ties differently)
    _, Igt = orc.flat_search(L2, xb, xq, 10, force_path=orc.PATH_PAIR)
    recalls = {}
    for nprobe in (1, 8, 64):
        _, I = g.search(xq, 10, nprobe=nprobe)
        recalls[nprobe] = np.mean([len(set(a) & set(b)) / 10 for a, b in zip(I, Igt)])
    assert recalls[64] == 1.0  # probing every list is exhaustive
    assert recalls[8] > 0.9 and recalls[1] > 0.4 and recalls[1] <= recalls[8] <= recalls[64]


def test_idmap_ivf_with_ids_selector_and_small_cases(mf):
    rng = np.random.default_rng(5)
    xb = rng.random((3000, 8), dtype=np.float32)
    ids = np.arange(3000, dtype=np.int64) * 3 + 7
    o = orc.Index(8, "IDMap,IVF4,Flat", L2)
    o.train(xb)
    o.add_with_ids(xb, ids)
    g = mf.index_factory(8, "IDMap,IVF4,Flat", L2)
    assert not g.is_trained
    g.train(xb)
    assert g.is_trained
    g.add_with_ids(xb, ids)
    D, I = g.search(xb[:40], 3, nprobe=4)
    Do, Io = o.search(xb[:40], 3, nprobe=4)
    assert np.array_equal(I, Io) and np.array_equal(D, Do)
    assert I[:, 0].tolist() == ids[:40].tolist()
    keep = ids[ids % 2 == 0]
    D, I = g.search(xb[:40], 3, nprobe=4, sel=("batch", keep))
    Do, Io = o.search(xb[:40], 3, nprobe=4, sel=("batch", keep))
    assert np.array_equal(I, Io) and np.array_equal(D, Do)
    # 'faiss_add_ids_with_train copy.test': IDMap,IVF1,Flat with a single vector (train(1) -> centroids = points)
    s = mf.index_factory(2, "IDMap,IVF1,Flat")
    x = np.array([[0.0040321066, 0.023423655]], np.float32)
    s.train(x)
    s.add_with_ids(x, np.array([231]))
    D, I = s.search(x, 2)
    assert I.tolist() == [[231, -1]]
    # too few training points: substring the glue matches (src/faiss_extension.cpp:400,592)
    t = mf.index_factory(4, "IVF8,Flat")
    with pytest.raises(mf.FaissException, match="should be at least as large as number of clusters"):
        t.train(np.zeros((3, 4), np.float32))
    with pytest.raises(mf.FaissException, match="is_trained"):
        t.add(np.zeros((3, 4), np.float32))


@pytest.mark.parametrize("metric", [L2, IP])
def test_ivf_with_hnsw_coarse_quantizer(mf, tmp_path, metric):
    """"IVF<n>_HNSW<m>,Flat" (reference Makefile:93): index_factory sets quantizer_trains_alone = 2 -- k-means on an
    IndexFlatL2, centroids inserted into an IndexHNSWFlat that then serves assign (k=1) and the coarse search"""
    d, nlist = 32, 32
    xb = _clustered(20000, d, 41, ncent=64, sigma=0.2)
    xq = _clustered(200, d, 42, ncent=64, sigma=0.2)
    desc = f"IVF{nlist}_HNSW8,Flat"
    o = orc.Index(d, desc, metric)
    g = mf.index_factory(d, desc, metric)
    assert g.kind == mf.KIND_IVFFLAT and not g.is_trained and g.quantizer.kind == mf.KIND_HNSW
    g.set_option("hnsw_build_waves", 1)  # reaches the quantizer: FAISS's single-thread graph = the oracle's
    o.train(xb)
    g.train(xb)
    assert g.is_trained and g.quantizer.ntotal == nlist
    assert np.array_equal(g.ivf_centroids().view(np.uint32), o.ivf_centroids().view(np.uint32))
    go, gg = o_graph(o), g.quantizer.hnsw_graph()
    assert np.array_equal(go["neighbors"], gg["neighbors"]) and go["entry_point"] == gg["entry_point"]
    o.add(xb)
    g.add(xb)
    for nprobe, efs in ((1, 0), (4, 16), (8, 64)):
        Do, Io = o.search(xq, 10, nprobe=nprobe, efSearch=efs)
        D, I = g.search(xq, 10, nprobe=nprobe, efSearch=efs)
        ok = np.ones(len(Do), dtype=bool)  # exact ties included (csrc/ivf_ties.hip)
        assert ok.sum() > 150
        assert np.array_equal(I[ok], Io[ok]) and np.array_equal(D[ok].view(np.uint32), Do[ok].view(np.uint32))
    # write_index / read_index keep the graph of the quantizer
    p = str(tmp_path / "ivf_hnsw.index")
    mf.write_index(g, p)
    ld = mf.read_index(p)
    assert ld.quantizer.kind == mf.KIND_HNSW and ld.ntotal == g.ntotal
    D0, I0 = g.search(xq, 10, nprobe=4, efSearch=32)
    D1, I1 = ld.search(xq, 10, nprobe=4, efSearch=32)
    assert np.array_equal(I0, I1) and np.array_equal(D0, D1)


def o_graph(o):
    """HNSW graph of the oracle's coarse quantizer"""
    return o.quantizer_hnsw_graph()


@pytest.mark.parametrize("d,nlist", [(64, 16), (128, 8), (200, 16)])
def test_ivf_inner_product_scans_on_the_mfma_kernel_bit_exact(mf, d, nlist):
    """inner product: IVFFlatScanner's fvec_inner_product is the k-ordered chain the MFMA computes, so the list scan
    runs as a segmented variant of the fused Flat kernel (items of <= 128 queries per list, lists padded to 64 rows
    in the pair-interleaved format) and still matches the oracle bit for bit; selectors through IDMap included"""
    n = 20000
    xb = _clustered(n, d, 61, ncent=64, sigma=0.2)
    xq = _clustered(300, d, 62, ncent=64, sigma=0.2)
    ids = np.arange(n, dtype=np.int64) * 2 + 9
    o = orc.Index(d, f"IDMap,IVF{nlist},Flat", IP)
    g = mf.index_factory(d, f"IDMap,IVF{nlist},Flat", IP)
    o.train(xb)
    g.ivf_set_centroids(o.ivf_centroids())
    for i0 in range(0, n, 6000):
        o.add_with_ids(xb[i0 : i0 + 6000], ids[i0 : i0 + 6000])
        g.add_with_ids(xb[i0 : i0 + 6000], ids[i0 : i0 + 6000])
    g.set_option("ivf_collect", 0)  # (large batches default to the bf16 coarse filter: tested below)
    keep = ids[np.arange(n) % 3 == 0]
    for nprobe, k, sel in ((1, 10, None), (4, 10, None), (nlist, 40, None), (4, 10, ("batch", keep))):
        Do, Io = o.search(xq, k, nprobe=nprobe, sel=sel)
        D, I = g.search(xq, k, nprobe=nprobe, sel=sel)
        assert g.last_kernel_info()["name"].startswith("ivf_mfma_scan")
        ok = np.ones(len(Do), dtype=bool)  # exact ties included (csrc/ivf_ties.hip)
        assert ok.sum() > 250
        assert np.array_equal(I[ok], Io[ok]), (nprobe, k, sel and sel[0])
        assert np.array_equal(D[ok].view(np.uint32), Do[ok].view(np.uint32))
    g.set_option("ivf_mfma", 0)  # the per-pair scan kernel gives the same answers
    D2, I2 = g.search(xq, 10, nprobe=4)
    assert g.last_kernel_info()["name"].startswith("ivf_scan_kernel")
    D1, I1 = o.search(xq, 10, nprobe=4)
    ok = np.ones(len(D1), dtype=bool)  # exact ties included (csrc/ivf_ties.hip)
    assert np.array_equal(I2[ok], I1[ok])


def test_ivf_l2_mfma_mode_is_recall_equivalent(mf):
    """L2 on the MFMA variant (option ivf_mfma = 1) evaluates ||x||^2 + ||y||^2 - 2<x,y> instead of the scanner's
    sum (x-y)^2: the same neighbours up to rounding-level near-ties, distances within 1e-4 relative"""
    d, nlist, n = 128, 32, 40000
    xb = _clustered(n, d, 71, ncent=128, sigma=0.15)
    xq = _clustered(500, d, 72, ncent=128, sigma=0.15)
    g = mf.index_factory(d, f"IVF{nlist},Flat", L2)
    g.train(xb)
    g.add(xb)
    De, Ie = g.search(xq, 10, nprobe=8)
    # default for L2: the scanner's arithmetic -- directly (small batches) or as the exact re-scoring behind the MFMA
    # prefilter (batches of >= 64 queries, csrc/ivf.hip mfma_prefilter_search); same bits either way
    assert g.last_kernel_info()["name"].startswith(("ivf_scan_kernel", "ivf_mfma_prefilter", "ivf_bf16_collect"))
    g.set_option("ivf_mfma", 1)
    Dm, Im = g.search(xq, 10, nprobe=8)
    assert g.last_kernel_info()["name"].startswith("ivf_mfma_scan")
    overlap = np.mean([len(set(a) & set(b)) / 10 for a, b in zip(Im, Ie)])
    assert overlap >= 0.999, overlap
    same = Im == Ie
    np.testing.assert_allclose(Dm[same], De[same], rtol=1e-4, atol=1e-6)
    assert np.all(np.diff(Dm, axis=1) >= 0)


@pytest.mark.parametrize("metric", [L2, IP])
def test_every_list_probed_large_nprobe_times_k(mf, metric):
    """nprobe = nlist (FAISS clamps larger values): nprobe*k candidates per query exceed one LDS pass of the merge,
    which then runs in chunks; probing every list is exhaustive, so the result also equals the Flat per-pair search"""
    d, nlist, n, k = 16, 256, 30000, 40
    xb = _clustered(n, d, 81, ncent=300, sigma=0.3)
    xq = _clustered(60, d, 82, ncent=300, sigma=0.3)
    o = orc.Index(d, f"IVF{nlist},Flat", metric)
    g = mf.index_factory(d, f"IVF{nlist},Flat", metric)
    o.train(xb)
    g.ivf_set_centroids(o.ivf_centroids())
    o.add(xb)
    g.add(xb)
    D, I = g.search(xq, k, nprobe=nlist + 7)
    Do, Io = o.search(xq, k, nprobe=nlist + 7)
    ok = np.ones(len(Do), dtype=bool)  # exact ties included (csrc/ivf_ties.hip)
    assert ok.sum() > 40
    assert np.array_equal(I[ok], Io[ok]) and np.array_equal(D[ok].view(np.uint32), Do[ok].view(np.uint32))
    Df, If = orc.flat_search(metric, xb, xq, k, force_path=orc.PATH_PAIR)
    assert np.array_equal(I[ok], If[ok])


@pytest.mark.parametrize("metric", [L2, IP])
def test_ivf_edge_cases_empty_and_tiny(mf, metric):
    """trained but empty index, a single stored row, k larger than what the probed lists hold, one query"""
    d, nlist = 64, 8
    xb = _clustered(2000, d, 91, ncent=8, sigma=0.2)
    o = orc.Index(d, f"IVF{nlist},Flat", metric)
    g = mf.index_factory(d, f"IVF{nlist},Flat", metric)
    o.train(xb)
    g.ivf_set_centroids(o.ivf_centroids())
    neutral = np.finfo(np.float32).max * (1 if metric == L2 else -1)
    D, I = g.search(xb[:3], 5, nprobe=4)
    assert np.all(I == -1) and np.all(D == neutral)
    o.add(xb[:1])
    g.add(xb[:1])
    for a in (o, g):
        D, I = a.search(xb[:2], 5, nprobe=nlist)
        assert np.all(I[:, 0] == 0) and np.all(I[:, 1:] == -1) and np.all(D[:, 1:] == neutral)
    o.add(xb[1:40])
    g.add(xb[1:40])
    Do, Io = o.search(xb[5:6], 30, nprobe=2)  # one query, k beyond the rows of the two probed lists
    D, I = g.search(xb[5:6], 30, nprobe=2)
    assert np.array_equal(I, Io) and np.array_equal(D.view(np.uint32), Do.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("metric", [L2, IP])
def test_row_sharded_lists_with_replicated_centroids_equal_one_index(mf, metric):
    """one set of centroids on every GPU, each inverted list row-sharded; the host merge of the shards'
    results must be the single-index result bit for bit (here: three shard indexes on one device)."""
    n, d, nq, k, nprobe = 60000, 64, 300, 10, 6
    xb = orc.synth_clustered(n, d, 1234, n_centers=64, sigma=0.2)
    xq = orc.synth_clustered(nq, d, 4321, n_centers=64, sigma=0.2)
    one = mf.index_factory(d, "IVF64,Flat", metric)
    one.train(xb)
    one.add(xb)
    Dr, Ir = one.search(xq, k, nprobe=nprobe)
    cent = one.ivf_centroids()
    bounds = [0, 17000, 17001, 41000, n]  # ragged shards, one of a single row
    Ds, Is = [], []
    for r0, r1 in zip(bounds[:-1], bounds[1:]):
        sh = mf.index_factory(d, "IVF64,Flat", metric)
        sh.ivf_set_centroids(cent)
        assert sh.is_trained
        sh.set_label_offset(r0)  # implicit ids of the shard = global row numbers (stored in the lists at add time)
        sh.add(xb[r0:r1])
        with pytest.raises(mf.FaissException, match="before rows are added"):
            sh.set_label_offset(r0 + 1)
        D, I = sh.search(xq, k, nprobe=nprobe)
        Ds.append(D)
        Is.append(I)
    Dm, Im = mf.merge_shards(metric, np.stack(Ds), np.stack(Is))
    assert np.array_equal(Dm, Dr)
    assert np.array_equal(Im, Ir)


@pytest.mark.parametrize("idmap", [False, True])
@pytest.mark.parametrize("d,nlist,n,nq,k,nprobe", [(128, 64, 60000, 500, 10, 8), (64, 16, 20000, 200, 5, 16), (96, 32, 30000, 64, 20, 4)])
def test_l2_prefilter_equals_scanner_and_oracle(mf, d, nlist, n, nq, k, nprobe, idmap):
    """option ivf_mfma = 2, L2 batches of >= 64 queries: MFMA list scan on residual rows as a prefilter (k + 4 candidates) + exact re-scoring
    in IVFFlatScanner's arithmetic + per-query proof; must equal the plain scanner kernel and the oracle bit for bit, also
    on duplicate-heavy data where queries are re-run (same coarse assignment) and with a selector"""
    xb = orc.synth_clustered(n, d, 31, n_centers=nlist, sigma=0.2)
    xq = orc.synth_clustered(nq, d, 32, n_centers=nlist, sigma=0.2)
    xb[n // 2 :: 7] = xb[: len(xb[n // 2 :: 7])]  # duplicates: exact distance ties near the top
    xq[: nq // 4] = xb[5 : 5 + nq // 4]
    ids = (np.arange(n, dtype=np.int64) * 3 + 11)
    desc = ("IDMap," if idmap else "") + f"IVF{nlist},Flat"
    g, o = mf.index_factory(d, desc, L2), orc.Index(d, desc, L2)
    o.train(xb)
    g.ivf_set_centroids(o.ivf_centroids())
    for a in (g, o):
        a.add_with_ids(xb, ids)
    keep = ids[::2]
    for sel in (None, ("batch", keep)):
        g.set_option("ivf_mfma", 2)
        D1, I1 = g.search(xq, k, nprobe=nprobe, sel=sel)
        assert g.last_kernel_info()["name"].startswith("ivf_mfma_prefilter")
        g.set_option("ivf_mfma", 0)
        D0, I0 = g.search(xq, k, nprobe=nprobe, sel=sel)
        assert g.last_kernel_info()["name"].startswith("ivf_scan_kernel")
        g.set_option("ivf_mfma", -1)
        Do, Io = o.search(xq, k, nprobe=nprobe, sel=sel)
        ok = _no_tie_rows(Do)
        assert ok.sum() >= 1
        assert np.array_equal(D1, D0) and np.array_equal(D1.view(np.uint32), Do.view(np.uint32)), (sel and sel[0])
        assert np.array_equal(I1[ok], I0[ok]) and np.array_equal(I1[ok], Io[ok]), (sel and sel[0])


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("idmap", [False, True])
def test_ivf_large_k_select_path(mf, metric, idmap):
    """k beyond the k-list kernels (> 256) -- the harness's post-filter runs ask an IVF index for ~2 000 rows
    (go/main_test.go:17-45): all distances of the probed lists + one segmented sort (csrc/ivf_select.hip)"""
    d, nlist, n = 48, 32, 40000
    xb = _clustered(n, d, 31)
    xq = _clustered(70, d, 32)
    desc = f"IDMap,IVF{nlist},Flat" if idmap else f"IVF{nlist},Flat"
    o = orc.Index(d, desc, metric)
    o.train(xb)
    g = mf.index_factory(d, desc, metric)
    g.ivf_set_centroids(o.ivf_centroids())
    ids = (np.random.RandomState(1).permutation(3 * n)[:n] + 11).astype(np.int64)
    for a in (o, g):
        a.add_with_ids(xb, ids) if idmap else a.add(xb)
    all_ids = ids if idmap else np.arange(n)
    keep = all_ids[np.random.RandomState(2).rand(n) < 0.3].astype(np.int64)
    for k, nprobe, q, sel in [(300, 4, xq, None), (1000, 8, xq[:1], None), (2048, 8, xq, None), (2048, 1, xq[:5], None),
                              (1500, 8, xq[:20], ("batch", keep)), (3000, 32, xq[:3], None)]:
        Do, Io = o.search(q, k, nprobe=nprobe, sel=sel)
        D, I = g.search(q, k, nprobe=nprobe, sel=sel)
        # (round 6: up to k = 2048 the bf16 filter against a frozen bound serves these -- collect_search_big; beyond, the select path)
        assert g.last_kernel_info()["name"].startswith("ivf_select") == (k > 2048), (k, g.last_kernel_info()["name"])
        # same candidates, same per-pair arithmetic: the sorted distance lists agree bit for bit (incl. the -1 / neutral
        # padding when fewer than k rows were probed); labels wherever a distance is unique within its list
        assert np.array_equal(D.view(np.uint32), Do.view(np.uint32)), (k, nprobe)
        assert np.array_equal(I == -1, Io == -1)
        uniq = np.ones_like(I, dtype=bool)
        uniq[:, 1:] &= Do[:, 1:] != Do[:, :-1]
        uniq[:, :-1] &= Do[:, 1:] != Do[:, :-1]
        assert uniq[Io >= 0].mean() > 0.9
        assert np.array_equal(I[uniq], Io[uniq]), (k, nprobe)


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("d,nlist,n,nq,k,nprobe,idmap", [(128, 32, 40000, 200, 33, 8, False), (128, 32, 40000, 100, 100, 16, True),
                                                          (64, 16, 30000, 40, 257, 4, True), (128, 32, 40000, 12, 1000, 32, False)])
def test_ivf_lists_beyond_32_entries_on_the_coarse_filter(mf, metric, d, nlist, n, nq, k, nprobe, idmap):
    """round 6: k > 32 on an IVF index -- B(q) = the k-th best exact value among the rows of the query's nearest lists, the grouped bf16
    scan of every probed list against that bound FROZEN, the candidates re-scored in the scanner's arithmetic and sorted
    (IVFFlatIndex::collect_search_big).  Same lists as the scanner / select kernels (option ivf_cl_big = 0) and as the oracle, exact
    ties included (the wrapper around every path replays FAISS's heap for the tied boundaries)."""
    xb = _clustered(n, d, 61 + k)
    xq = _clustered(nq, d, 62 + k)
    xb[::41] = xb[7]  # duplicates: exact ties inside the lists and at boundaries
    desc = f"IDMap,IVF{nlist},Flat" if idmap else f"IVF{nlist},Flat"
    o = orc.Index(d, desc, metric)
    o.train(xb)
    g = mf.index_factory(d, desc, metric)
    g.ivf_set_centroids(o.ivf_centroids())
    ids = (np.random.RandomState(3).permutation(3 * n)[:n] + 5).astype(np.int64)
    for a in (o, g):
        a.add_with_ids(xb, ids) if idmap else a.add(xb)
    keep = (ids if idmap else np.arange(n))[np.random.RandomState(4).rand(n) < 0.4].astype(np.int64)
    for sel in (None, ("batch", keep)):
        D1, I1 = g.search(xq, k, nprobe=nprobe, sel=sel)
        assert g.last_kernel_info()["name"] == "ivf_bf16_collect_kernel", g.last_kernel_info()
        g.set_option("ivf_cl_big", 0)
        D0, I0 = g.search(xq, k, nprobe=nprobe, sel=sel)
        assert g.last_kernel_info()["name"] != "ivf_bf16_collect_kernel"
        g.set_option("ivf_cl_big", 2)  # (the bound from every row of the nearest lists, through the select path)
        D2, I2 = g.search(xq, k, nprobe=nprobe, sel=sel)
        assert g.last_kernel_info()["name"] == "ivf_bf16_collect_kernel"
        assert np.array_equal(D2.view(np.uint32), D0.view(np.uint32)) and np.array_equal(I2, I0), (sel and sel[0])
        g.set_option("ivf_cl_big", 1)
        assert np.array_equal(D1.view(np.uint32), D0.view(np.uint32)), (sel and sel[0])
        assert np.array_equal(I1, I0), (sel and sel[0])
        no = min(nq, 8)
        Do, Io = o.search(xq[:no], k, nprobe=nprobe, sel=sel)
        assert np.array_equal(D1[:no].view(np.uint32), Do.view(np.uint32)), (sel and sel[0])
        assert np.array_equal(I1[:no], Io), (sel and sel[0])


def test_ivf_k_beyond_the_tie_pass_lds_keeps_working(mf):
    """the exact-tie wrapper was admitted for every k < 16 384 although the tie pass keeps A_k in LDS (fits up to
    ~12 700 at d = 128): k in between threw "IVF tie pass: k too large" AFTER the whole k + 1 search.  Such k now keep the pure
    order, as k >= 16 384 always did."""
    d, nlist, n = 128, 16, 20000
    xb = _clustered(n, d, 51)
    xq = _clustered(6, d, 52)
    o = orc.Index(d, f"IVF{nlist},Flat", L2)
    o.train(xb)
    g = mf.index_factory(d, f"IVF{nlist},Flat", L2)
    g.ivf_set_centroids(o.ivf_centroids())
    o.add(xb)
    g.add(xb)
    for k in (12000, 13000, 16383, 16384):
        Do, Io = o.search(xq, k, nprobe=nlist)
        D, I = g.search(xq, k, nprobe=nlist)
        assert np.array_equal(D.view(np.uint32), Do.view(np.uint32)), k
        uniq = np.ones_like(I, dtype=bool)
        uniq[:, 1:] &= Do[:, 1:] != Do[:, :-1]
        uniq[:, :-1] &= Do[:, 1:] != Do[:, :-1]
        assert np.array_equal(I[uniq], Io[uniq]), k


def test_ivf_select_path_equals_k_list_path_at_small_k(mf):
    d, nlist, n = 64, 16, 20000
    xb = _clustered(n, d, 41)
    xq = _clustered(100, d, 42)
    g = mf.index_factory(d, f"IVF{nlist},Flat", L2)
    g.train(xb)
    g.add(xb)
    D0, I0 = g.search(xq, 50, nprobe=4)
    g.set_option("ivf_select", 1)
    D1, I1 = g.search(xq, 50, nprobe=4)
    assert g.last_kernel_info()["name"].startswith("ivf_select")
    assert np.array_equal(I0, I1) and np.array_equal(D0.view(np.uint32), D1.view(np.uint32))


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("idmap", [False, True])
@pytest.mark.parametrize("d,nlist,n,nq,k,nprobe", [(128, 64, 60000, 500, 10, 8), (64, 16, 20000, 200, 5, 16), (96, 32, 30000, 64, 15, 4),  # (15: k + 1 = 16 entries with the tie detection of csrc/ivf_ties.hip)
                                                   (128, 32, 50000, 300, 1, 1), (100, 48, 40000, 777, 10, 48),
                                                   # 16 < k + 1 <= 32: 32 row classes per query (ivf_bf16_collect_kernel<32>)
                                                   (128, 32, 50000, 300, 20, 6), (64, 16, 30000, 150, 31, 8), (96, 64, 60000, 90, 16, 12)])
def test_l2_coarse_filter_equals_scanner_and_oracle(mf, d, nlist, n, nq, k, nprobe, idmap, metric):
    """default for L2 batches of >= 64 queries, k <= 16 (option ivf_collect): bf16 coarse filter on residual rows with a
    proven bound + exact re-scoring in IVFFlatScanner's arithmetic (csrc/ivf_collect.hip); must equal the plain scanner
    kernel and the oracle bit for bit, also on duplicate-heavy data (all tied rows are candidates) and with an id map"""
    xb = orc.synth_clustered(n, d, 31, n_centers=nlist, sigma=0.2)
    xq = orc.synth_clustered(nq, d, 32, n_centers=nlist, sigma=0.2)
    xb[n // 2 :: 7] = xb[: len(xb[n // 2 :: 7])]  # duplicates: exact distance ties near the top
    xq[: nq // 4] = xb[5 : 5 + nq // 4]
    ids = (np.arange(n, dtype=np.int64) * 3 + 11)
    desc = ("IDMap," if idmap else "") + f"IVF{nlist},Flat"
    g, o = mf.index_factory(d, desc, metric), orc.Index(d, desc, metric)
    o.train(xb)
    g.ivf_set_centroids(o.ivf_centroids())
    for a in (g, o):
        a.add_with_ids(xb, ids)
    D1, I1 = g.search(xq, k, nprobe=nprobe)
    assert g.last_kernel_info()["name"].startswith("ivf_bf16_collect")
    g.set_option("ivf_collect", 0)
    D0, I0 = g.search(xq, k, nprobe=nprobe)
    assert g.last_kernel_info()["name"].startswith(("ivf_scan_kernel", "ivf_mfma_scan"))
    Do, Io = o.search(xq, k, nprobe=nprobe)
    ok = np.ones(len(Do), dtype=bool)  # exact ties included (csrc/ivf_ties.hip)
    assert ok.sum() >= 1
    # (values of tied rows agree by definition, so whole distance rows agree; labels away from exact ties)
    assert np.array_equal(D1.view(np.uint32), D0.view(np.uint32)) and np.array_equal(D1.view(np.uint32), Do.view(np.uint32))
    assert np.array_equal(I1[ok], I0[ok]) and np.array_equal(I1[ok], Io[ok])
    # IDSelector: one selector bit per row in front of the same kernel; the scanner kernel and the oracle agree
    keep = ids[::2] if idmap else np.arange(n, dtype=np.int64)[::2]
    for sel in (("batch", keep), ("bitmap", np.packbits(np.isin(np.arange(int(keep.max()) + 8), keep), bitorder="little"))):
        g.set_option("ivf_collect", -1)
        D2, I2 = g.search(xq, k, nprobe=nprobe, sel=sel)
        assert g.last_kernel_info()["name"].startswith("ivf_bf16_collect")
        g.set_option("ivf_collect", 0)
        D3, I3 = g.search(xq, k, nprobe=nprobe, sel=sel)
        assert g.last_kernel_info()["name"].startswith(("ivf_scan_kernel", "ivf_mfma_scan"))
        Do2, Io2 = o.search(xq, k, nprobe=nprobe, sel=sel)
        assert np.isin(I2[I2 >= 0], keep).all()
        assert np.array_equal(D2.view(np.uint32), D3.view(np.uint32)) and np.array_equal(D2.view(np.uint32), Do2.view(np.uint32)), sel[0]
        ok2 = np.ones(len(Do2), dtype=bool)  # exact ties included (csrc/ivf_ties.hip)
        assert np.array_equal(I2[ok2], I3[ok2]) and np.array_equal(I2[ok2], Io2[ok2]), sel[0]


@pytest.mark.parametrize("metric", [L2, IP])
def test_coarse_filter_stream_overflow_grows_the_stream(mf, metric):
    """a candidate stream that is too small (here: 64 entries per query by option; in the field: duplicate-heavy lists) is grown
    once and the main pass repeated -- the batch stays on the coarse filter instead of falling to the scanner kernel; a stream
    that would need more than 16 384 entries per query still hands the batch over.  Same results either way."""
    d, nlist, n, nq, k, nprobe = 64, 16, 40_000, 300, 10, 8
    xb = orc.synth_clustered(n, d, 41, n_centers=nlist, sigma=0.2)
    xq = orc.synth_clustered(nq, d, 42, n_centers=nlist, sigma=0.2)
    g, o = mf.index_factory(d, f"IVF{nlist},Flat", metric), orc.Index(d, f"IVF{nlist},Flat", metric)
    o.train(xb)
    g.ivf_set_centroids(o.ivf_centroids())
    g.add(xb)
    o.add(xb)
    Do, Io = o.search(xq, k, nprobe=nprobe)
    D0, I0 = g.search(xq, k, nprobe=nprobe)
    assert g.last_kernel_info()["name"].startswith("ivf_bf16_collect")
    g.set_option("ivf_cl_stream_cap", 8)  # the first main pass overflows, the grown stream holds the candidates
    D1, I1 = g.search(xq, k, nprobe=nprobe)
    assert g.last_kernel_info()["name"].startswith("ivf_bf16_collect")
    g.set_option("ivf_cl_stream_cap", 0)
    for D, I in ((D0, I0), (D1, I1)):
        assert np.array_equal(I, Io) and np.array_equal(D.view(np.uint32), Do.view(np.uint32))
    # 6 000 copies of one vector in one list, every query on top of it: 6 000 tied candidates per query at k = 10 ... the
    # stream would need > 16 384 x nq / ... entries only with more copies; here the grown stream must still serve it
    xb2 = xb.copy()
    xb2[:6000] = xb[7]
    g2, o2 = mf.index_factory(d, f"IVF{nlist},Flat", metric), orc.Index(d, f"IVF{nlist},Flat", metric)
    o2.train(xb2)
    g2.ivf_set_centroids(o2.ivf_centroids())
    g2.add(xb2)
    o2.add(xb2)
    xq2 = np.repeat(xb[7:8], 200, axis=0)
    D2, I2 = g2.search(xq2, k, nprobe=nprobe)
    Do2, Io2 = o2.search(xq2, k, nprobe=nprobe)
    assert np.array_equal(I2, Io2) and np.array_equal(D2.view(np.uint32), Do2.view(np.uint32))


def test_l2_coarse_filter_non_finite_queries_fall_back(mf):
    d, nlist, n = 128, 16, 20000
    xb = orc.synth_clustered(n, d, 5, n_centers=nlist, sigma=0.2)
    xq = orc.synth_clustered(100, d, 6, n_centers=nlist, sigma=0.2)
    xq[3, 7] = np.nan
    xq[9] *= 1e19
    g, o = mf.index_factory(d, f"IVF{nlist},Flat", L2), orc.Index(d, f"IVF{nlist},Flat", L2)
    o.train(xb)
    g.ivf_set_centroids(o.ivf_centroids())
    g.add(xb)
    o.add(xb)
    D1, I1 = g.search(xq, 10, nprobe=4)
    assert g.last_kernel_info()["name"].startswith("ivf_bf16_collect")
    g.set_option("ivf_collect", 0)
    D0, I0 = g.search(xq, 10, nprobe=4)
    assert np.array_equal(I1, I0) and np.array_equal(D1.view(np.uint32), D0.view(np.uint32))


@pytest.mark.parametrize("nlist,nprobe,d,nq", [(256, 32, 128, 300), (1000, 7, 100, 77), (4096, 32, 64, 500), (512, 255, 32, 64),
                                               (2048, 1, 128, 20), (8192, 64, 16, 130)])
@pytest.mark.parametrize("metric", [L2, IP])
def test_coarse_quantiser_by_distance_matrix(mf, metric, nlist, nprobe, d, nq):
    """IVF coarse quantisation over a few thousand centroids (csrc/coarse_select.hip): the whole [nq][nlist] distance matrix in
    the BLAS-branch arithmetic + one wavefront per query selecting the nprobe smallest (dis, id).  Same probes, same order as the
    k-list kernels (option ivf_coarse_select = 0) and as the oracle -- checked through the final result of the IVF search, with
    DUPLICATE centroids (exact ties at the nprobe-th distance decide which list is probed), queries sitting on centroids and a
    non-finite query."""
    rs = np.random.RandomState(nlist + nprobe)
    n = max(20_000, 6 * nlist)
    xb = _clustered(n, d, nlist)
    cent = xb[rs.permutation(n)[:nlist]].copy()
    dup = rs.permutation(nlist)[: nlist // 8]
    cent[dup] = cent[rs.permutation(nlist)[: nlist // 8]]  # exact ties between centroids
    xq = _clustered(nq, d, nlist + 1)
    xq[: nq // 4] = cent[rs.randint(0, nlist, nq // 4)]  # distance 0 to a (possibly duplicated) centroid
    xq[nq // 2, 0] = np.nan
    g = mf.index_factory(d, f"IVF{nlist},Flat", metric)
    g.ivf_set_centroids(cent)
    g.add(xb)
    D1, I1 = g.search(xq, 10, nprobe=nprobe)
    g.set_option("ivf_coarse_select", 0)
    D0, I0 = g.search(xq, 10, nprobe=nprobe)
    g.set_option("ivf_coarse_select", 1)
    # (inner product: boundary ties between duplicate centroids follow FAISS's CMin heap through resolve_ip_ties on both paths)
    assert np.array_equal(I1, I0) and np.array_equal(D1.view(np.uint32), D0.view(np.uint32))
    o = orc.Index(d, f"IVF{nlist},Flat", metric)
    o.ivf_set_centroids(cent)
    o.add(xb)
    m = min(nq, 96)
    Do, Io = o.search(xq[:m], 10, nprobe=nprobe)
    assert np.array_equal(D1[:m].view(np.uint32), Do.view(np.uint32))
    ok = np.ones_like(Io, dtype=bool)
    if metric == IP:  # (equal scores inside a result list: FAISS's order depends on the probe order; tests above)
        ok[:, 1:] &= Do[:, 1:] != Do[:, :-1]
        ok[:, :-1] &= Do[:, 1:] != Do[:, :-1]
    assert np.array_equal(I1[:m][ok], Io[ok])


@pytest.mark.parametrize("metric", [L2, IP])
def test_coarse_selection_when_one_lane_owns_the_nearest_centroids(mf, metric):
    """The selection's fast path bounds the nprobe-th distance by the nprobe-th smallest LANE minimum; when the nearest centroids
    all sit in one lane's ids (4 (64 i + lane) + e) that bound admits most of the row and the kernel falls back to the full
    bitwise search (csrc/coarse_select.hip).  Same result either way."""
    rs = np.random.RandomState(4)
    d, nlist, nprobe, nq = 32, 4096, 40, 60
    sign = 1.0 if metric == L2 else -1.0
    cent = (rs.rand(nlist, d) + 50.0 * sign).astype(np.float32)  # far (L2) / low score (inner product)
    near = np.array([4 * 64 * i + e for i in range(16) for e in range(4)])  # the 64 ids lane 0 holds
    cent[near] = (rs.rand(64, d) * 0.1).astype(np.float32) * (1.0 if metric == L2 else 30.0)
    xb = np.concatenate([cent[near] + rs.rand(64, d).astype(np.float32) * 0.01 for _ in range(40)] +
                        [cent[rs.randint(0, nlist, 3000)] + rs.rand(3000, d).astype(np.float32) * 0.01]).astype(np.float32)
    xq = (rs.rand(nq, d) * 0.1).astype(np.float32)
    g = mf.index_factory(d, f"IVF{nlist},Flat", metric)
    g.ivf_set_centroids(cent)
    g.add(xb)
    D1, I1 = g.search(xq, 10, nprobe=nprobe)
    g.set_option("ivf_coarse_select", 0)
    D0, I0 = g.search(xq, 10, nprobe=nprobe)
    g.set_option("ivf_coarse_select", 1)
    assert np.array_equal(I1, I0) and np.array_equal(D1.view(np.uint32), D0.view(np.uint32))
    assert (I1 >= 0).all() and (I1 < 2560).mean() > 0.9  # the rows around the near centroids


def _tied_data(n, d, seed, metric):
    """30 % duplicated rows on small-integer coordinates: most queries are tied at rank k, inside the result and across the
    probed lists (inner product: integer scores collide all the time)"""
    rs = np.random.RandomState(seed)
    xb = rs.randint(-3, 4, size=(n, d)).astype(np.float32)
    dup = rs.rand(n) < 0.3
    xb[dup] = xb[rs.randint(0, n, size=int(dup.sum()))]
    xq = rs.randint(-3, 4, size=(257, d)).astype(np.float32)
    xq[:64] = xb[rs.randint(0, n, size=64)]  # queries ON rows: distance-0 ties among the copies
    return xb, xq


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("desc", ["IVF16,Flat", "IDMap,IVF16,Flat"])
@pytest.mark.parametrize("d", [32, 128, 200])
def test_ivf_exact_ties_follow_the_heap(mf, metric, desc, d):
    """duplicate-heavy IVF data, label-exact on EVERY query, every scan path (bf16 coarse filter, scanner
    kernel, f32 MFMA items, k beyond the coarse filter, the all-distances path, selectors, < 20 queries)"""
    n, nlist = 12000, 16
    xb, xq = _tied_data(n, d, 77 + d, metric)
    rs = np.random.RandomState(5)
    ids = (rs.permutation(4 * n)[:n] + 3).astype(np.int64)  # stored ids NOT in arrival order: the heap compares (value, id)
    o = orc.Index(d, desc, metric)
    g = mf.index_factory(d, desc, metric)
    o.train(xb)
    g.ivf_set_centroids(o.ivf_centroids())
    for a in (g, o):
        for i0 in range(0, n, 5000):
            a.add_with_ids(xb[i0 : i0 + 5000], ids[i0 : i0 + 5000])
    keep = ids[rs.rand(n) < 0.5]
    seen = set()
    for opts, k, nprobe, sel, nq in (
        ({}, 10, 4, None, 257),
        ({}, 10, 16, None, 257),
        ({}, 1, 3, None, 257),
        ({"ivf_collect": 0}, 10, 4, None, 257),
        ({"ivf_collect": 0, "ivf_mfma": 0}, 10, 5, None, 257),
        ({}, 16, 4, None, 100),
        ({}, 40, 8, None, 100),
        ({}, 10, 4, ("batch", keep), 257),
        ({"ivf_collect": 0}, 10, 4, ("batch", keep), 64),
        ({}, 5, 6, None, 7),
        ({"ivf_select": 1}, 10, 4, None, 64),
        ({}, 300, 16, None, 33),
    ):
        for key, v in opts.items():
            g.set_option(key, v)
        D, I = g.search(xq[:nq], k, nprobe=nprobe, sel=sel)
        seen.add(g.last_kernel_info()["name"].split(" ")[0])
        for key in opts:
            g.set_option(key, {"ivf_collect": -1, "ivf_mfma": -1, "ivf_select": 0}[key])
        Do, Io = o.search(xq[:nq], k, nprobe=nprobe, sel=sel)
        what = (opts, k, nprobe, sel and sel[0], nq)
        assert np.array_equal(D.view(np.uint32), Do.view(np.uint32)), what
        bad = np.flatnonzero((I != Io).any(axis=1))
        assert bad.size == 0, (what, bad[:5], I[bad[:1]], Io[bad[:1]], D[bad[:1]])
        if k == 10 and nprobe == 4 and sel is None and not opts:
            Dk1, _ = o.search(xq[:nq], k + 1, nprobe=nprobe)
            assert (Dk1[:, k - 1] == Dk1[:, k]).mean() > 0.2  # the boundary case is really exercised
    assert len(seen) >= (3 if d <= 128 else 2), seen  # (d > 128: no coarse filter, no f32 MFMA items)


def test_tie_emit_checks_its_preconditions(mf):
    """mvs_index_ivf_tie_emit_device reuses the coarse assignment of the search that has just run -- another batch, more
    flagged queries than that batch held or a query number outside it is refused; "ivf_ids_ascending" tells the cross-process merge
    whether (probe rank, id) still is FAISS's arrival order"""
    import torch

    d, n = 32, 20000
    xb = _clustered(n, d, 71)
    g = mf.index_factory(d, "IVF16,Flat", L2)
    g.train(xb)
    g.add(xb)
    assert g.get_stat("ivf_ids_ascending") == 1
    dev = torch.device("cuda", 0)
    xq = torch.from_numpy(_clustered(50, d, 72)).to(dev)
    other = xq.clone()
    g.set_option("ivf_exact_ties", 0)
    D, I = g.search_torch(xq, 6, nprobe=4)
    T = D[:3, 4].contiguous()
    v, ids, rk = g.ivf_tie_emit_torch(torch.tensor([0, 1, 2], device=dev), xq, T, 5)  # the batch that was searched: fine
    assert ids.shape == (3, 5) and int((ids >= 0).sum()) > 0
    with pytest.raises(Exception, match="is not the one of the search"):
        g.ivf_tie_emit_torch(torch.tensor([0, 1, 2], device=dev), other, T, 5)
    with pytest.raises(Exception, match="outside the last search"):
        g.ivf_tie_emit_torch(torch.tensor([0, 1, 50], device=dev), xq, T, 5)
    w = mf.index_factory(d, "IVF16,Flat", L2)  # (behind an IDMap the IVF index stores row numbers: ascending whatever the user's ids are)
    w.train(xb)
    w.add_with_ids(xb[:100], np.arange(100, dtype=np.int64) + 5)
    assert w.get_stat("ivf_ids_ascending") == 1
    w.add_with_ids(xb[100:200], np.arange(100, dtype=np.int64)[::-1].copy() + 1000)
    assert w.get_stat("ivf_ids_ascending") == 0


def test_ivf_exact_ties_option_off_keeps_the_pure_order(mf):
    """ivf_exact_ties = 0 (diagnostics): same values, the scan kernels' (value, position) order"""
    xb, xq = _tied_data(6000, 32, 3, L2)
    g = mf.index_factory(32, "IVF8,Flat", L2)
    g.train(xb)
    g.add(xb)
    D1, I1 = g.search(xq, 10, nprobe=3)
    g.set_option("ivf_exact_ties", 0)
    D0, I0 = g.search(xq, 10, nprobe=3)
    assert np.array_equal(D0, D1) and not np.array_equal(I0, I1)


def test_ivf_coarse_filter_switches_do_not_change_a_single_bit(mf):
    """residual-norm bound (cl_bound_mode), items of one list on one XCD (ivf_cl_xcd), the final-bound filter (ivf_cl_refilter): same
    labels and distances as the scanner kernel in every combination, first and second search"""
    d, nlist, n = 128, 64, 120_000
    xb = _clustered(n, d, 61)
    xq = _clustered(900, d, 62)
    ref = mf.index_factory(d, f"IVF{nlist},Flat", L2)
    ref.train(xb)
    ref.add(xb)
    ref.set_option("ivf_collect", 0)
    D0, I0 = ref.search(xq, 10, nprobe=8)
    cent = ref.ivf_centroids()
    for bound in (1, 0):
        for xcd in (1, 0, 2, 3):  # (2 / 3: the segments of an item as neighbouring workgroups -- measured slower, kept as options)
            for refilter in (1, 0):
                if xcd >= 2 and (bound == 0 or refilter == 0):
                    continue
                g = mf.index_factory(d, f"IVF{nlist},Flat", L2)
                g.ivf_set_centroids(cent)
                g.add(xb)
                g.set_option("cl_bound_mode", bound)
                g.set_option("ivf_cl_xcd", xcd)
                g.set_option("ivf_cl_refilter", refilter)
                for rep in range(2):
                    D, I = g.search(xq, 10, nprobe=8)
                    assert g.last_kernel_info()["name"].startswith("ivf_bf16_collect")
                    assert np.array_equal(I, I0) and np.array_equal(D.view(np.uint32), D0.view(np.uint32)), (bound, xcd, refilter, rep)


@pytest.mark.gpu
@pytest.mark.parametrize("metric", [L2, IP])
def test_ivf_stream_and_buckets_sized_too_small_are_grown_and_run_again(mf, metric):
    """The candidate count stays on the device; a stream or a per-query bucket that turns out too small (ivf_cl_stream_cap: 8 entries
    per query) is noticed at the search's one synchronisation, grown, and the pass repeated -- same bits as the scanner kernel."""
    d, nlist, n = 64, 32, 60_000
    xb = _clustered(n, d, 71)
    xq = _clustered(400, d, 72)
    ref = mf.index_factory(d, f"IVF{nlist},Flat", metric)
    ref.train(xb)
    ref.add(xb)
    ref.set_option("ivf_collect", 0)
    D0, I0 = ref.search(xq, 10, nprobe=6)
    g = mf.index_factory(d, f"IVF{nlist},Flat", metric)
    g.ivf_set_centroids(ref.ivf_centroids())
    g.add(xb)
    for cap in (0, 8, 64, 0):
        g.set_option("ivf_cl_stream_cap", cap)
        for rep in range(2):
            D, I = g.search(xq, 10, nprobe=6)
            assert g.last_kernel_info()["name"].startswith("ivf_bf16_collect")
            assert np.array_equal(I, I0) and np.array_equal(D.view(np.uint32), D0.view(np.uint32)), (cap, rep)


@pytest.mark.gpu
@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("k", [1, 16, 31, 32])
def test_ivf_k_on_the_coarse_filters_class_limits_with_exact_ties(mf, metric, k):
    """The exact-tie wrapper searches k + 1 entries; the coarse filter works with the user's k (a row tied with the k-th value passes
    any bound derived from k rows), so k = 32 stays on ivf_bf16_collect_kernel<32> and k = 16 on <16> (k = 32 used to fall
    to the scanner kernel).  Integer data with duplicated rows: ties at the k-th value in most queries; equal to the scanner kernel,
    which replays FAISS's heap too."""
    rs = np.random.RandomState(90 + k)
    d, nlist, n, nq = 32, 16, 40_000, 300
    xb = rs.randint(0, 3, size=(n, d)).astype(np.float32)
    m = n // 3
    xb[0 : 3 * m : 3] = xb[1 : 3 * m : 3]
    xq = rs.randint(0, 3, size=(nq, d)).astype(np.float32)
    ref = mf.index_factory(d, f"IVF{nlist},Flat", metric)
    ref.train(xb[:8000])
    ref.add(xb)
    ref.set_option("ivf_collect", 0)
    D0, I0 = ref.search(xq, k, nprobe=5)
    g = mf.index_factory(d, f"IVF{nlist},Flat", metric)
    g.ivf_set_centroids(ref.ivf_centroids())
    g.add(xb)
    D, I = g.search(xq, k, nprobe=5)
    assert g.last_kernel_info()["name"].startswith("ivf_bf16_collect"), g.last_kernel_info()["name"]
    assert np.array_equal(I, I0) and np.array_equal(D.view(np.uint32), D0.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("metric", [L2, IP])
def test_grouping_by_lds_histograms_at_an_odd_list_count(mf, metric):
    """>= 16 384 (query, list) pairs take the per-workgroup LDS histograms of csrc/ivf_scan.hip (ivf_group_*_lds_kernel); a list count
    that is no power of two, queries that crowd a few lists (clustered), chunks that end inside the last workgroup: same bits as the
    scanner kernel, and the same again when the batch is small enough for the plain atomics."""
    d, nlist, n = 48, 100, 50_000
    xb = _clustered(n, d, 81, ncent=20)
    xq = _clustered(2100, d, 82, ncent=20)
    ref = mf.index_factory(d, f"IVF{nlist},Flat", metric)
    ref.train(xb)
    ref.add(xb)
    ref.set_option("ivf_collect", 0)
    g = mf.index_factory(d, f"IVF{nlist},Flat", metric)
    g.ivf_set_centroids(ref.ivf_centroids())
    g.add(xb)
    for nq, nprobe in ((2100, 9), (1500, 9), (2100, 100)):
        D0, I0 = ref.search(xq[:nq], 10, nprobe=nprobe)
        D, I = g.search(xq[:nq], 10, nprobe=nprobe)
        assert g.last_kernel_info()["name"].startswith("ivf_bf16_collect")
        assert np.array_equal(I, I0) and np.array_equal(D.view(np.uint32), D0.view(np.uint32)), (nq, nprobe)


@pytest.mark.parametrize("where", ["first", "middle", "last"])
@pytest.mark.parametrize("bad", [np.nan, np.inf, -np.inf])
def test_training_rejects_non_finite_values_wherever_they_sit(mf, where, bad):
    """faiss/Clustering.cpp train_encoded checks EVERY training value ("input contains NaN's or Inf's"); the check runs on several host
    threads since round 6 -- a bad value in any thread's part, also the array's very last float, must be found."""
    d, n = 32, 300_000  # 9.6 M floats: more than one thread's share
    x = np.random.RandomState(1).rand(n, d).astype(np.float32)
    pos = {"first": (0, 0), "middle": (n // 2 + 17, 5), "last": (n - 1, d - 1)}[where]
    x[pos] = bad
    ix = mf.index_factory(d, "IVF64,Flat", L2)
    with pytest.raises(mf.FaissException, match="input contains NaN's or Inf's"):
        ix.train(x)
    x[pos] = 0.5
    ix.train(x)  # (and the clean array trains)
    assert ix.is_trained
