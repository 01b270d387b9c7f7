"""Self-consistency of the CPU oracle: the packed AVX2 'BLAS path' equals the naive k-ordered
fma triple loop bit for bit; FAISS dispatch thresholds; tie rules;
IVF restatement sanity.  These paths have no golden values in the reference (parity unpinned)."""
import numpy as np
import pytest

from oracle import oracle as orc


def _data(nb, nq, d, seed=0, dup=0):
    rng = np.random.default_rng(seed)
    xb = rng.random((nb, d), dtype=np.float32)
    xq = rng.random((nq, d), dtype=np.float32)
    if dup:
        # duplicated rows pin tie-breaking
        src = rng.integers(0, nb, dup)
        dst = rng.integers(0, nb, dup)
        xb[dst] = xb[src]
    return xb, xq


@pytest.mark.parametrize("metric", [orc.METRIC_L2, orc.METRIC_INNER_PRODUCT])
@pytest.mark.parametrize("d", [8, 33, 128])
def test_packed_blas_path_equals_naive_chain(metric, d):
    xb, xq = _data(3000, 37, d, seed=d, dup=200)
    D, I = orc.flat_search(metric, xb, xq, 10, force_path=orc.PATH_BLAS)
    Dn, In = orc.flat_search_naive(metric, xb, xq, 10, orc.PATH_BLAS)
    assert np.array_equal(I, In)
    assert np.array_equal(D.view(np.uint32), Dn.view(np.uint32))


@pytest.mark.parametrize("metric", [orc.METRIC_L2, orc.METRIC_INNER_PRODUCT])
def test_pair_path_equals_naive_chain(metric):
    xb, xq = _data(2000, 7, 24, seed=3, dup=100)
    D, I = orc.flat_search(metric, xb, xq, 5, force_path=orc.PATH_PAIR)
    Dn, In = orc.flat_search_naive(metric, xb, xq, 5, orc.PATH_PAIR)
    assert np.array_equal(I, In) and np.array_equal(D.view(np.uint32), Dn.view(np.uint32))


def test_dispatch_threshold_20():
    """nq < 20 -> per-pair sum((x-y)^2); nq >= 20 -> xn + yn - 2 ip (distance_compute_blas_threshold)"""
    xb, xq = _data(1500, 20, 16, seed=5)
    Dp, _ = orc.flat_search(orc.METRIC_L2, xb, xq[:19], 4)
    Dp_ref, _ = orc.flat_search_naive(orc.METRIC_L2, xb, xq[:19], 4, orc.PATH_PAIR)
    assert np.array_equal(Dp, Dp_ref)
    Db, _ = orc.flat_search(orc.METRIC_L2, xb, xq, 4)
    Db_ref, _ = orc.flat_search_naive(orc.METRIC_L2, xb, xq, 4, orc.PATH_BLAS)
    assert np.array_equal(Db, Db_ref)
    # float64 ground truth: both within 1e-4 relative (north_star tolerance)
    S = ((xq[:, None, :].astype(np.float64) - xb[None].astype(np.float64)) ** 2).sum(-1)
    ref = np.sort(S, axis=1)[:, :4]
    np.testing.assert_allclose(Db, ref, rtol=1e-4)


def test_l2_ties_are_lexicographic_and_ascending():
    """L2 (CMax heap): retained set = k smallest (dist, id); output ascending (dist, id)."""
    xb = np.zeros((12, 4), np.float32)
    xb[:, 0] = [5, 1, 1, 1, 3, 1, 0, 0, 9, 1, 2, 1]
    xq = np.zeros((1, 4), np.float32)
    D, I = orc.flat_search(orc.METRIC_L2, xb, xq, 5, force_path=orc.PATH_PAIR)
    assert I.tolist() == [[6, 7, 1, 2, 3]]
    assert D.tolist() == [[0, 0, 1, 1, 1]]
    Db, Ib = orc.flat_search(orc.METRIC_L2, xb, xq, 5, force_path=orc.PATH_BLAS)
    assert Ib.tolist() == I.tolist()


def test_ip_tie_rule_is_arrival_order_dependent():
    """IP (CMin heap): equal scores print in DESCENDING id order; a later strictly better element
    evicts the SMALLEST-id tie (root among equal values) -- SURVEY.md Appendix A.1."""
    xq = np.array([[1.0, 0.0]], np.float32)
    xb = np.array([[5, 0], [5, 0], [5, 0], [7, 0]], np.float32)
    D, I = orc.flat_search(orc.METRIC_INNER_PRODUCT, xb, xq, 2)
    assert D.tolist() == [[7, 5]] and I.tolist() == [[3, 1]]
    xb2 = np.array([[5, 0], [5, 0], [5, 0]], np.float32)
    D, I = orc.flat_search(orc.METRIC_INNER_PRODUCT, xb2, xq, 2)
    assert I.tolist() == [[1, 0]]


def test_k_larger_than_ntotal_and_empty_index():
    xb, xq = _data(3, 2, 4)
    D, I = orc.flat_search(orc.METRIC_L2, xb, xq, 5)
    assert (I[:, 3:] == -1).all() and (D[:, 3:] == np.finfo(np.float32).max).all()
    ix = orc.Index(4, "Flat", orc.METRIC_L2)
    D, I = ix.search(xq, 3)
    assert (I == -1).all()
    with pytest.raises(orc.OracleError, match="k > 0"):
        ix.search(xq, 0)


def test_factory_strings():
    assert orc.Index(8, "Flat").is_trained
    assert orc.Index(8, "IDMap,Flat").is_trained
    assert not orc.Index(8, "IVF16,Flat").is_trained
    assert not orc.Index(8, "IDMap,IVF16,Flat").is_trained
    with pytest.raises(orc.OracleError, match="could not parse index string"):
        orc.Index(8, "Bogus")
    with pytest.raises(orc.OracleError, match="add does not make sense"):
        orc.Index(8, "IDMap,Flat").add(np.zeros((1, 8), np.float32))


def test_ivf_full_probe_equals_pairwise_flat():
    """nprobe = nlist scans every vector with the per-pair arithmetic -> same set as Flat pair path"""
    rng = np.random.default_rng(11)
    centers = rng.normal(size=(8, 16)).astype(np.float32)
    xb = (centers[rng.integers(0, 8, 4000)] + 0.1 * rng.normal(size=(4000, 16))).astype(np.float32)
    xq = (centers[rng.integers(0, 8, 30)] + 0.1 * rng.normal(size=(30, 16))).astype(np.float32)
    ix = orc.Index(16, "IVF8,Flat", orc.METRIC_L2)
    ix.train(xb)
    assert ix.is_trained
    ix.add(xb)
    assert ix.ntotal == 4000
    assert sum(len(ix.ivf_list(l)[0]) for l in range(8)) == 4000
    D, I = ix.search(xq, 10, nprobe=8)
    Df, If = orc.flat_search(orc.METRIC_L2, xb, xq, 10, force_path=orc.PATH_PAIR)
    assert np.array_equal(np.sort(I, 1), np.sort(If, 1))
    assert np.array_equal(D, Df)
    # nprobe=1 recall on clustered data is high but not necessarily 1
    D1, I1 = ix.search(xq, 10, nprobe=1)
    rec = np.mean([len(set(a) & set(b)) / 10 for a, b in zip(I1, If)])
    assert rec > 0.8
    # within a list entries keep input order (Appendix A.6)
    ids0, _ = ix.ivf_list(0)
    assert np.all(np.diff(ids0) > 0)


def test_ivf_with_ids_and_selector():
    rng = np.random.default_rng(2)
    xb = rng.random((600, 8), dtype=np.float32)
    ids = np.arange(600, dtype=np.int64) * 3 + 7
    ix = orc.Index(8, "IDMap,IVF4,Flat", orc.METRIC_L2)
    ix.train(xb)
    ix.add_with_ids(xb, ids)
    D, I = ix.search(xb[:5], 1, nprobe=4)
    assert I.reshape(-1).tolist() == ids[:5].tolist()
    keep = ids[ids % 2 == 0]
    D, I = ix.search(xb[:25], 3, nprobe=4, sel=("batch", keep))
    assert np.isin(I[I >= 0], keep).all()


def test_merge_shards_equals_unsharded():
    xb, xq = _data(5000, 40, 32, seed=9, dup=300)
    for metric in (orc.METRIC_L2, orc.METRIC_INNER_PRODUCT):
        Dref, Iref = orc.flat_search(metric, xb, xq, 10)
        parts = np.array_split(np.arange(5000), 4)
        Ds, Is = [], []
        for p in parts:
            D, I = orc.flat_search(metric, xb[p], xq, 10, force_path=orc.PATH_BLAS)
            Ds.append(D)
            Is.append(np.where(I >= 0, I + p[0], -1))
        Dm, Im = orc.merge_shards(metric, np.stack(Ds), np.stack(Is))
        assert np.array_equal(Dm, Dref)
        if metric == orc.METRIC_L2:
            assert np.array_equal(Im, Iref)
        else:  # IP boundary ties are arrival-order dependent in FAISS; compare away from ties
            D11, _ = orc.flat_search(metric, xb, xq, 11)
            ok = np.array([len(np.unique(D11[q])) == 11 for q in range(40)])
            assert ok.sum() > 20
            assert np.array_equal(Im[ok], Iref[ok])


def test_synth_generators_are_deterministic_and_windowed():
    a = orc.synth_uniform(100, 16, 1234)
    b = orc.synth_uniform(40, 16, 1234, row0=60)
    assert np.array_equal(a[60:], b)
    assert 0 <= a.min() and a.max() < 1 and abs(a.mean() - 0.5) < 0.02
    c = orc.synth_clustered(200, 8, 7, n_centers=4, sigma=0.05)
    c2 = orc.synth_clustered(50, 8, 7, row0=150, n_centers=4, sigma=0.05)
    assert np.array_equal(c[150:], c2)


def _heap_stream(is_max, k, vals, ids):
    """FAISS's k-heap fed one (value, id) at a time with the strict insert rule, then heap_reorder (oracle/orc_core.c)"""
    import ctypes as C

    L = orc.lib()
    hv = np.empty(k, dtype=np.float32)
    hi = np.empty(k, dtype=np.int64)
    pf, pi = hv.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p)
    L.orc_heap_init.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
    L.orc_heap_replace_top.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int64]
    L.orc_heap_reorder.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
    L.orc_heap_init(k, pf, pi, is_max)
    for v, i in zip(vals, ids):
        if (hv[0] > v) if is_max else (hv[0] < v):
            L.orc_heap_replace_top(k, pf, pi, is_max, float(v), int(i))
    L.orc_heap_reorder(k, pf, pi, is_max)
    return hv, hi


def _closed_form(is_max, k, vals, ids):
    """what csrc/ivf_ties.hip computes: pure top-k by value, runs printed by id; at a boundary tie A_k = the first k arrivals
    not worse than T, result = {better than T} + {tied rows of A_k minus the G extreme ids}"""
    n = len(vals)
    sgn = 1.0 if is_max else -1.0
    key = sgn * vals  # smaller = better
    order = np.lexsort((np.arange(n), key))  # pure order: (value, arrival position)
    if n <= k:
        sel = list(order)
    else:
        T = key[order[k - 1]]
        if key[order[k]] != T:
            sel = list(order[:k])
        else:
            A = [i for i in range(n) if key[i] <= T][:k]
            better = [i for i in range(n) if key[i] < T]
            G = len(better) - sum(1 for i in A if key[i] < T)
            tied = sorted((i for i in A if key[i] == T), key=lambda i: ids[i])
            tied = tied[: len(tied) - G] if is_max else tied[G:]
            sel = better + tied
    # print order: L2 (value asc, id asc); inner product (value desc, id desc)
    sel.sort(key=lambda i: (key[i], ids[i] if is_max else -ids[i]))
    hv = np.full(k, np.float32(3.4028234663852886e38 if is_max else -3.4028234663852886e38), dtype=np.float32)
    hi = np.full(k, -1, dtype=np.int64)
    hv[: len(sel)] = vals[sel]
    hi[: len(sel)] = ids[sel]
    return hv, hi


@pytest.mark.parametrize("is_max", [1, 0])
def test_ivf_tie_closed_form_equals_the_heap_in_any_arrival_order(is_max):
    """IVF feeds the heap in probe order, not id order: the closed form the device implements (csrc/ivf_ties.hip) against
    the oracle's heap replay on streams with few distinct values and ids unrelated to arrival order"""
    rs = np.random.RandomState(11 + is_max)
    for trial in range(400):
        n = int(rs.choice([3, 8, 20, 60, 200]))
        k = int(rs.choice([1, 2, 5, 10, 17]))
        vals = rs.randint(0, int(rs.choice([2, 4, 12])), size=n).astype(np.float32)
        ids = rs.permutation(5 * n)[:n].astype(np.int64)
        hv, hi = _heap_stream(is_max, k, vals, ids)
        cv, ci = _closed_form(is_max, k, vals, ids)
        assert np.array_equal(hv, cv), (trial, n, k)
        assert np.array_equal(hi, ci), (trial, n, k, vals, ids, hi, ci)


def test_reservoir_from_k_100_equals_the_heap_for_l2_and_differs_for_ip():
    """FAISS's ReservoirTopN (k >= 100; oracle/orc_core.c reservoir_t) against the heap at the same k: for L2 (rows arrive in
    ascending id) both keep the k smallest (distance, id); for inner product the rows tied at the k-th score differ in a fair
    share of tie-heavy cases -- the values never do."""
    rs = np.random.RandomState(5)
    nd = {orc.METRIC_L2: 0, orc.METRIC_INNER_PRODUCT: 0}
    try:
        for trial in range(60):
            nb, k = int(rs.randint(300, 3000)), int(rs.choice([100, 120, 200]))
            vals = rs.randint(0, rs.randint(3, 40), size=nb).astype(np.float32)
            xb = np.zeros((nb, 8), np.float32)
            xb[:, 0] = vals
            for metric in nd:
                xq = np.zeros((1, 8), np.float32)
                xq[0, 0] = 1.0 if metric == orc.METRIC_INNER_PRODUCT else 0.0
                orc.set_reservoir(True)
                Dr, Ir = orc.flat_search(metric, xb, xq, k)
                orc.set_reservoir(False)
                Dh, Ih = orc.flat_search(metric, xb, xq, k)
                assert np.array_equal(Dr, Dh)
                nd[metric] += not np.array_equal(Ir, Ih)
                # either way: k distinct rows, each carrying its own value, in FAISS's print order
                assert len(set(Ir[0].tolist())) == k
                ref = vals[Ir[0]] if metric == orc.METRIC_INNER_PRODUCT else vals[Ir[0]] ** 2
                assert np.array_equal(ref, Dr[0])
    finally:
        orc.set_reservoir(True)
    assert nd[orc.METRIC_L2] == 0 and nd[orc.METRIC_INNER_PRODUCT] > 0, nd
    # the crafted stream of DESIGN.md 3.5: 150 tied rows, 50 better ones, 100 more tied; k = 100
    vals = np.array([5] * 150 + [9] * 50 + [5] * 100, np.float32)
    xb = np.zeros((len(vals), 8), np.float32)
    xb[:, 0] = vals
    xq = np.zeros((1, 8), np.float32)
    xq[0, 0] = 1.0
    _, Ir = orc.flat_search(orc.METRIC_INNER_PRODUCT, xb, xq, 100)
    assert sorted(Ir[0][50:].tolist()) == list(range(50))  # the shrink at the boundary kept the FIRST 50 tied rows ...
    orc.set_reservoir(False)
    try:
        _, Ih = orc.flat_search(orc.METRIC_INNER_PRODUCT, xb, xq, 100)
    finally:
        orc.set_reservoir(True)
    assert sorted(Ih[0][50:].tolist()) == list(range(50, 100))  # ... where the heap evicts them for the later, better rows

