"""IndexHNSWFlat on device vs the CPU oracle's restatement (oracle/orc_hnsw.c).  The reference has no golden values
for HNSW (parity unpinned by the reference); FAISS's own graph depends on OpenMP thread interleaving.  With one build
wave (option hnsw_build_waves = 1) the device inserts in FAISS's single-thread order and must reproduce the oracle's
graph, labels and distances BIT FOR BIT; with the default concurrent build only recall is comparable."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
L2, IP = orc.METRIC_L2, orc.METRIC_INNER_PRODUCT
FMAX = np.finfo(np.float32).max


@pytest.fixture(scope="module")
def mf():
    import mi355_faiss

    return mi355_faiss


def _recall(I, I_ref):
    return np.mean([len(set(a) & set(b)) / len(b) for a, b in zip(I, I_ref)])


def _pair(mf, d, desc, metric, xb, efc=None, chunk=None, waves=1):
    o = orc.Index(d, desc, metric)
    g = mf.index_factory(d, desc, metric)
    g.set_option("hnsw_build_waves", waves)
    if efc:
        o.hnsw_set_ef_construction(efc)
        g.set_ef_construction(efc)
    step = chunk or len(xb)
    for i in range(0, len(xb), step):
        o.add(xb[i : i + step])
        g.add(xb[i : i + step])
    return o, g


def _assert_same_graph(o, g):
    a, b = o.hnsw_graph(), g.hnsw_graph()
    assert a["max_level"] == b["max_level"] and a["entry_point"] == b["entry_point"]
    assert np.array_equal(a["levels"], b["levels"]) and np.array_equal(a["offsets"], b["offsets"])
    if not np.array_equal(a["neighbors"], b["neighbors"]):
        bad = np.flatnonzero(a["neighbors"] != b["neighbors"])
        v = int(np.searchsorted(a["offsets"], bad[0], side="right") - 1)
        raise AssertionError(f"{len(bad)} neighbour slots differ; first at vertex {v}: "
                             f"{a['neighbors'][a['offsets'][v]:a['offsets'][v+1]]} vs "
                             f"{b['neighbors'][b['offsets'][v]:b['offsets'][v+1]]}")


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("d,M,n", [(32, 16, 3000), (100, 8, 1500), (768, 32, 1200)])
def test_deterministic_build_reproduces_oracle_graph(mf, metric, d, M, n):
    xb = orc.synth_uniform(n, d, 21)
    o, g = _pair(mf, d, f"HNSW{M}", metric, xb)
    assert g.kind == mf.KIND_HNSW and g.ntotal == n and g.is_trained
    _assert_same_graph(o, g)


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("d,M,n", [(16, 4, 4000), (64, 8, 3000)])
def test_full_list_short_cut_is_the_full_shrink(mf, metric, d, M, n):
    """round 6: add_link's short cut for a list that is full and still the output of its last shrink (csrc/hnsw.hip) -- small M on
    uniform rows fills the lists early, so most back links take it; the graph is the oracle's (FAISS's pairwise pass) bit for bit,
    with the short cut and without, for fewer distance evaluations"""
    xb = orc.synth_uniform(n, d, 29)
    o, g = _pair(mf, d, f"HNSW{M}", metric, xb, chunk=1000)
    _assert_same_graph(o, g)
    assert g.get_stat("hnsw_build_shortcuts") > 100, g.get_stat("hnsw_build_shortcuts")
    g0 = mf.index_factory(d, f"HNSW{M}", metric)
    g0.set_option("hnsw_build_waves", 1)
    g0.set_option("hnsw_build_shortcut", 0)
    for i in range(0, n, 1000):
        g0.add(xb[i : i + 1000])
    _assert_same_graph(o, g0)
    assert g0.get_stat("hnsw_build_shortcuts") == 0
    assert g.get_stat("hnsw_build_distances") < g0.get_stat("hnsw_build_distances")
    # the option can be flipped between adds: the flags of lists that changed unflagged are dropped with it
    g1 = mf.index_factory(d, f"HNSW{M}", metric)
    g1.set_option("hnsw_build_waves", 1)
    for j, i in enumerate(range(0, n, 1000)):
        g1.set_option("hnsw_build_shortcut", j & 1)
        g1.add(xb[i : i + 1000])
    _assert_same_graph(o, g1)


def test_harness_shape_hnsw128_d1536(mf):
    """the Go harness index (go/benches_c.go:59): IDMap,HNSW128,Flat, d=1536, default metric inner product -- level-0
    lists of 256 slots (4 lane chunks), 6 float4 per lane per row"""
    d, n = 1536, 700
    xb = orc.synth_clustered(n, d, 40, n_centers=16, sigma=0.5)
    xq = orc.synth_clustered(20, d, 41, n_centers=16, sigma=0.5)
    ids = np.arange(n, dtype=np.int64) + 10**6
    o = orc.Index(d, "IDMap,HNSW128,Flat", IP)
    g = mf.index_factory(d, "IDMap,HNSW128,Flat", IP)
    g.set_option("hnsw_build_waves", 1)
    o.add_with_ids(xb, ids)
    g.add_with_ids(xb, ids)
    _assert_same_graph(o, g)
    for k, efs in ((11, 16), (200, 64)):  # the harness sweeps k far beyond efSearch
        Do, Io = o.search(xq, k, efSearch=efs)
        Dg, Ig = g.search(xq, k, efSearch=efs)
        assert np.array_equal(Ig, Io) and np.array_equal(Dg.view(np.uint32), Do.view(np.uint32))


def test_incremental_adds_like_duckdb_chunks(mf):
    """the glue adds <= 2048 rows per call (src/faiss_extension.cpp:510-512): levels and graph continue across calls"""
    xb = orc.synth_clustered(5000, 48, 22, n_centers=32, sigma=0.2)
    o, g = _pair(mf, 48, "HNSW16", L2, xb, chunk=2048)
    _assert_same_graph(o, g)


def test_ef_construction_is_honoured(mf):
    xb = orc.synth_uniform(2000, 24, 23)
    o, g = _pair(mf, 24, "HNSW8,Flat", L2, xb, efc=100)
    _assert_same_graph(o, g)
    o2 = orc.Index(24, "HNSW8", L2)
    o2.add(xb)
    assert not np.array_equal(o2.hnsw_graph()["neighbors"], o.hnsw_graph()["neighbors"])


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("efs,k", [(16, 10), (128, 10), (8, 40), (300, 100), (64, 1)])
def test_search_bit_exact_on_same_graph(mf, metric, efs, k):
    d, n = 64, 6000
    xb, xq = orc.synth_uniform(n, d, 24), orc.synth_uniform(333, d, 25)
    o, g = _pair(mf, d, "HNSW16", metric, xb)
    _assert_same_graph(o, g)
    Do, Io = o.search(xq, k, efSearch=efs)
    Dg, Ig = g.search(xq, k, efSearch=efs)
    assert np.array_equal(Ig, Io)
    assert np.array_equal(Dg.view(np.uint32), Do.view(np.uint32))


@pytest.mark.parametrize("opts", [{"hnsw_visited_lds": 0}, {"hnsw_visited_lds": 1024}, {"hnsw_search_g": 2},
                                  {"hnsw_search_g": 8, "hnsw_search_waves": 3}])
def test_search_variants_are_exact(mf, opts):
    """the visited set (LDS hash, its spill into the HBM byte table when it fills up, HBM table only) and the
    rows-in-flight / occupancy knobs must not change a single bit of the answer"""
    d, n = 48, 8000
    xb, xq = orc.synth_uniform(n, d, 34), orc.synth_uniform(200, d, 35)
    o, g = _pair(mf, d, "HNSW16", L2, xb)
    for key, v in opts.items():
        g.set_option(key, v)
    for efs, k in ((300, 10), (40, 10)):  # 300: > 1024 * 3/4 visited vertices -> the small hash spills
        Do, Io = o.search(xq, k, efSearch=efs)
        Dg, Ig = g.search(xq, k, efSearch=efs)
        assert np.array_equal(Ig, Io) and np.array_equal(Dg.view(np.uint32), Do.view(np.uint32))


def test_search_d768_bit_exact_and_recall(mf):
    """BASELINE config C5 shape at reduced N: IDMap,HNSW32 d=768, L2-normalised rows, efSearch=128"""
    d, n = 768, 4000
    xb = orc.synth_clustered(n, d, 26, n_centers=64, sigma=0.3)
    xb /= np.linalg.norm(xb, axis=1, keepdims=True)
    xq = orc.synth_clustered(100, d, 27, n_centers=64, sigma=0.3)
    xq /= np.linalg.norm(xq, axis=1, keepdims=True)
    ids = np.arange(n, dtype=np.int64) * 7 + 5
    o = orc.Index(d, "IDMap,HNSW32", L2)
    g = mf.index_factory(d, "IDMap,HNSW32", L2)
    g.set_option("hnsw_build_waves", 1)
    o.add_with_ids(xb, ids)
    g.add_with_ids(xb, ids)
    _assert_same_graph(o, g)
    Do, Io = o.search(xq, 10, efSearch=128)
    Dg, Ig = g.search(xq, 10, efSearch=128)
    assert np.array_equal(Ig, Io) and np.array_equal(Dg.view(np.uint32), Do.view(np.uint32))
    fl = orc.Index(d, "Flat", L2)
    fl.add(xb)
    _, If = fl.search(xq, 10, force_path=orc.PATH_PAIR)
    assert _recall(Ig, ids[If]) >= 0.95


def test_selectors_and_idmap(mf):
    d, n = 32, 5000
    xb, xq = orc.synth_uniform(n, d, 28), orc.synth_uniform(64, d, 29)
    ids = (np.arange(n, dtype=np.int64) * 3 + 100)[::-1].copy()
    o = orc.Index(d, "IDMap,HNSW16", IP)
    g = mf.index_factory(d, "IDMap,HNSW16", IP)
    g.set_option("hnsw_build_waves", 1)
    g.set_ef_construction(64)  # through the IDMap wrapper, like src/faiss_extension.cpp:127-139
    o.hnsw_set_ef_construction(64)
    o.add_with_ids(xb, ids)
    g.add_with_ids(xb, ids)
    keep = ids[(np.arange(n) % 4) == 0]
    bm = np.zeros(int(ids.max()) // 8 + 1, dtype=np.uint8)
    for i in keep:
        bm[i >> 3] |= 1 << (i & 7)
    for sel in (None, ("batch", keep), ("bitmap", bm)):
        Do, Io = o.search(xq, 10, efSearch=48, sel=sel)
        Dg, Ig = g.search(xq, 10, efSearch=48, sel=sel)
        assert np.array_equal(Ig, Io), sel and sel[0]
        assert np.array_equal(Dg.view(np.uint32), Do.view(np.uint32))
        if sel:
            assert np.all(np.isin(Ig[Ig >= 0], keep))


def test_edge_cases(mf):
    d = 8
    g = mf.index_factory(d, "HNSW8", IP)
    xq = orc.synth_uniform(5, d, 30)
    D, I = g.search(xq, 3)
    assert np.all(I == -1) and np.all(D == -FMAX)  # empty index
    with pytest.raises(mf.FaissException, match="add_with_ids not implemented"):
        g.add_with_ids(xq, np.arange(5))
    with pytest.raises(mf.FaissException, match="k > 0"):
        g.search(xq, 0)
    g.set_option("hnsw_build_waves", 1)
    xb = orc.synth_uniform(50, d, 31)
    g.add(xb[:1])  # a single vertex: entry point, no links
    D, I = g.search(xq, 3)
    assert np.all(I[:, 0] == 0) and np.all(I[:, 1:] == -1)
    g.add(xb[1:])
    o = orc.Index(d, "HNSW8", IP)
    o.add(xb[:1])
    o.add(xb[1:])
    _assert_same_graph(o, g)
    Do, Io = o.search(xq, 64, efSearch=4)  # k > ntotal, k > efSearch
    Dg, Ig = g.search(xq, 64, efSearch=4)
    assert np.array_equal(Ig, Io) and np.array_equal(Dg.view(np.uint32), Do.view(np.uint32))


@pytest.mark.parametrize("metric", [L2, IP])
def test_concurrent_build_recall(mf, metric):
    """default build = many waves under per-vertex locks (FAISS's OpenMP semantics): graph differs from the
    single-thread order, recall must not"""
    d, n = 64, 30000
    xb = orc.synth_clustered(n, d, 32, n_centers=256, sigma=0.25)
    xq = orc.synth_clustered(500, d, 33, n_centers=256, sigma=0.25)
    g = mf.index_factory(d, "HNSW32", metric)
    for i in range(0, n, 2048):
        g.add(xb[i : i + 2048])
    gr = g.hnsw_graph()
    nb, off, lev = gr["neighbors"], gr["offsets"], gr["levels"]
    for v in range(0, n, 97):  # structural invariants survive the concurrency
        lst = nb[off[v] : off[v] + 64]
        used = lst[lst >= 0]
        assert np.all(lst[: len(used)] >= 0) and np.all(lst[len(used) :] == -1)
        assert len(set(used.tolist())) == len(used) and v not in used
        assert np.all((used >= 0) & (used < n))
    fl = mf.index_factory(d, "Flat", metric)
    fl.add(xb)
    _, If = fl.search(xq, 10)
    r_conc = _recall(g.search(xq, 10, efSearch=128)[1], If)
    o = orc.Index(d, "HNSW32", metric)
    o.add(xb)
    r_orc = _recall(o.search(xq, 10, efSearch=128)[1], If)
    assert r_conc >= r_orc - 0.02, (r_conc, r_orc)
