"""CPU tests of the oracle's IndexHNSWFlat restatement (oracle/orc_hnsw.c).  The reference holds no golden values for
HNSW results (parity unpinned by the reference), so these pin the restatement against FAISS's documented
structure: level distribution, flat neighbour layout, link-count caps, and recall against exact Flat search."""
import numpy as np
import pytest

from oracle import oracle as orc

L2, IP = orc.METRIC_L2, orc.METRIC_INNER_PRODUCT


def _recall(I, I_ref):
    return np.mean([len(set(a) & set(b)) / len(b) for a, b in zip(I, I_ref)])


def _mt19937_level_draws(n, M):
    """HNSW::random_level with RandomGenerator(12345) -- numpy's MT19937 is std::mt19937 when seeded the legacy way"""
    rs = np.random.RandomState(12345)
    mult = 1.0 / np.log(M)
    probas = []
    level = 0
    while True:
        p = np.exp(-level / mult) * (1 - np.exp(-1 / mult))
        if p < 1e-9:
            break
        probas.append(p)
        level += 1
    # RandomGenerator::rand_float = mt() / float(mt.max())
    raw = rs.randint(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    out = []
    for r in raw:
        f = float(np.float32(r) / np.float32(4294967295.0))
        lv = len(probas) - 1
        for i, p in enumerate(probas):
            if f < p:
                lv = i
                break
            f -= p
        out.append(lv + 1)
    return np.array(out, dtype=np.int32)


def test_factory_strings_and_defaults():
    for desc in ("HNSW16", "HNSW32,Flat", "HNSW", "IDMap,HNSW32"):
        ix = orc.Index(8, desc, L2)
        assert ix.is_trained and ix.ntotal == 0
    with pytest.raises(orc.OracleError, match="could not parse"):
        orc.Index(8, "HNSW32,PQ8", L2)
    ix = orc.Index(8, "HNSW16", L2)
    with pytest.raises(orc.OracleError, match="add_with_ids not implemented"):
        ix.add_with_ids(np.zeros((1, 8), np.float32), [1])


def test_levels_follow_rng_12345():
    n, M = 5000, 16
    ix = orc.Index(4, f"HNSW{M}", L2)
    ix.add(orc.synth_uniform(n, 4, 1))
    g = ix.hnsw_graph()
    assert np.array_equal(g["levels"], _mt19937_level_draws(n, M))
    # offsets: 2M slots at level 0, M per level above
    want = np.concatenate([[0], np.cumsum(2 * M + (g["levels"].astype(np.int64) - 1) * M)])
    assert np.array_equal(g["offsets"], want)
    assert g["max_level"] == g["levels"].max() - 1
    assert g["levels"][g["entry_point"]] - 1 == g["max_level"]


def test_incremental_adds_continue_the_level_stream():
    n, M = 3000, 8
    x = orc.synth_uniform(n, 8, 2)
    a = orc.Index(8, f"HNSW{M}", L2)
    for i in range(0, n, 700):
        a.add(x[i : i + 700])
    assert np.array_equal(a.hnsw_graph()["levels"], _mt19937_level_draws(n, M))


@pytest.mark.parametrize("metric", [L2, IP])
def test_graph_invariants(metric):
    n, d, M = 3000, 16, 8
    x = orc.synth_uniform(n, d, 3)
    ix = orc.Index(d, f"HNSW{M}", metric)
    ix.add(x)
    g = ix.hnsw_graph()
    nb, off, lev = g["neighbors"], g["offsets"], g["levels"]
    for v in range(0, n, 7):
        for level in range(lev[v]):
            b = off[v] + (0 if level == 0 else (level + 1) * M)
            e = b + (2 * M if level == 0 else M)
            lst = nb[b:e]
            used = lst[lst >= 0]
            assert np.all(lst[: len(used)] >= 0) and np.all(lst[len(used) :] == -1)  # packed, -1 padded
            assert len(set(used.tolist())) == len(used) and v not in used  # no duplicates / self loops
            assert np.all(lev[used] > level)  # a link at `level` points to a vertex that exists there


@pytest.mark.parametrize("metric,min_recall", [(L2, 0.97), (IP, 0.93)])
def test_recall_against_flat(metric, min_recall):
    n, d = 8000, 32
    xb, xq = orc.synth_uniform(n, d, 5), orc.synth_uniform(200, d, 6)
    ix = orc.Index(d, "HNSW32", metric)
    ix.add(xb)
    fl = orc.Index(d, "Flat", metric)
    fl.add(xb)
    _, If = fl.search(xq, 10, force_path=orc.PATH_PAIR)
    r16 = _recall(ix.search(xq, 10, efSearch=16)[1], If)
    r128 = _recall(ix.search(xq, 10, efSearch=128)[1], If)
    assert r128 >= min_recall and r128 >= r16
    # distances returned are the true distances of the returned labels (canonical HNSW arithmetic, few ulp off the chain)
    D, I = ix.search(xq, 10, efSearch=128)
    if metric == L2:
        true = ((xq[:, None, :] - xb[I]) ** 2).sum(-1)
        assert np.all(np.diff(D, axis=1) >= 0)
    else:
        true = (xq[:, None, :] * xb[I]).sum(-1)
        assert np.all(np.diff(D, axis=1) <= 0)
    np.testing.assert_allclose(D, true, rtol=1e-4)


def test_k_larger_than_efsearch_and_ntotal():
    d = 8
    xb, xq = orc.synth_uniform(50, d, 7), orc.synth_uniform(5, d, 8)
    ix = orc.Index(d, "HNSW8", L2)
    ix.add(xb)
    D, I = ix.search(xq, 64, efSearch=4)  # ef = max(efSearch, k)
    assert np.all(I[:, 50:] == -1) and np.all(D[:, 50:] == np.finfo(np.float32).max)
    assert np.all(I[:, :4] >= 0)  # the walk stops once efSearch stored distances are below the popped one
    e = orc.Index(d, "HNSW8", IP)
    D, I = e.search(xq, 3)
    assert np.all(I == -1) and np.all(D == -np.finfo(np.float32).max)


def test_idmap_and_selector_filter_results_only():
    n, d = 4000, 16
    xb, xq = orc.synth_uniform(n, d, 9), orc.synth_uniform(50, d, 10)
    ids = (np.arange(n, dtype=np.int64) * 3 + 100)[::-1].copy()
    ix = orc.Index(d, "IDMap,HNSW16", L2)
    ix.hnsw_set_ef_construction(60)
    ix.add_with_ids(xb, ids)
    D, I = ix.search(xq, 10, efSearch=64)
    plain = orc.Index(d, "HNSW16", L2)
    plain.hnsw_set_ef_construction(60)
    plain.add(xb)
    D2, I2 = plain.search(xq, 10, efSearch=64)
    assert np.array_equal(I, ids[I2]) and np.array_equal(D, D2)
    keep = ids[(np.arange(n) % 3) == 0]
    Ds, Is = ix.search(xq, 10, efSearch=64, sel=("batch", keep))
    got = Is[Is >= 0]
    assert np.all(np.isin(got, keep)) and len(got) > 0
    bm = np.zeros(int(ids.max()) // 8 + 1, dtype=np.uint8)
    for i in keep:
        bm[i >> 3] |= 1 << (i & 7)
    Db, Ib = ix.search(xq, 10, efSearch=64, sel=("bitmap", bm))
    assert np.array_equal(Ib, Is) and np.array_equal(Db, Ds)


def test_ivf_with_hnsw_coarse_quantizer():
    """"IVF<n>_HNSW<m>,Flat" (reference Makefile:93): k-means on an IndexFlatL2 (quantizer_trains_alone = 2), so the
    centroids equal those of "IVF<n>,Flat" for L2; assign and coarse search walk the HNSW graph"""
    d, n = 16, 12000
    xb = orc.synth_clustered(n, d, 50, n_centers=32, sigma=0.2)
    xq = orc.synth_clustered(100, d, 51, n_centers=32, sigma=0.2)
    a = orc.Index(d, "IVF32,Flat", L2)
    b = orc.Index(d, "IVF32_HNSW8,Flat", L2)
    assert not b.is_trained
    a.train(xb)
    b.train(xb)
    assert np.array_equal(a.ivf_centroids(), b.ivf_centroids())
    g = b.quantizer_hnsw_graph()
    assert len(g["levels"]) == 32 and g["entry_point"] >= 0
    a.add(xb)
    b.add(xb)
    Da, Ia = a.search(xq, 10, nprobe=32)
    Db, Ib = b.search(xq, 10, nprobe=32, efSearch=64)  # every list probed: both exhaustive
    assert np.array_equal(Ia, Ib) and np.array_equal(Da, Db)
    with pytest.raises(orc.OracleError, match="could not parse"):
        orc.Index(d, "IVF32_HNSW8,PQ4", L2)


def test_pop_min_tie_rule_only_matters_under_exact_distance_ties():
    """oracle/orc_hnsw.c: the walk's MinimaxHeap::pop_min in FAISS's heap-array order (default) or by smallest id (the device walk's
    rule).  On float data without duplicate rows the two rules give the same results bit for bit; with every row stored twice
    (candidates at bit-equal distance everywhere) both still return k valid neighbours at the same distances."""
    d, n, nq, k = 16, 3000, 40, 5
    xb, xq = orc.synth_uniform(n, d, 31), orc.synth_uniform(nq, d, 32)
    ix = orc.Index(d, "HNSW16", orc.METRIC_L2)
    ix.add(xb)
    try:
        D0, I0 = ix.search(xq, k, efSearch=32)
        orc.hnsw_set_pop_min_rule(1)
        D1, I1 = ix.search(xq, k, efSearch=32)
        assert np.array_equal(I0, I1) and np.array_equal(D0.view(np.uint32), D1.view(np.uint32))
        orc.hnsw_set_pop_min_rule(0)
        dup = orc.Index(d, "HNSW16", orc.METRIC_L2)
        dup.add(np.concatenate([xb[:1500], xb[:1500]]))
        Da, Ia = dup.search(xq, k, efSearch=32)
        orc.hnsw_set_pop_min_rule(1)
        Db, Ib = dup.search(xq, k, efSearch=32)
        assert (Ia >= 0).all() and (Ib >= 0).all()
        assert np.array_equal(Da.view(np.uint32), Db.view(np.uint32))  # (which twin of a pair is returned may differ, its distance cannot)
    finally:
        orc.hnsw_set_pop_min_rule(0)
