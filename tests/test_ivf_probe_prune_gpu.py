"""Probe pruning of the IVF coarse-filter path (round 5, csrc/ivf_collect.hip ivf_probe_prune_kernel): a probed list is left out of
the scan when the triangle inequality on its coarse distance and radius PROVES that every one of its rows is farther away than k
rows of the query's nearest lists.  IndexIVF::search (faiss/IndexIVF.cpp, reached from src/faiss_extension.cpp:631) scans every
probed list, so the only acceptable outcome is: not one label, not one distance bit changes -- against the oracle's IVF, against
the same index with the option off, under exact ties, with k larger than the nearest list, on data where nothing can be pruned."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
L2, IP = orc.METRIC_L2, orc.METRIC_INNER_PRODUCT


@pytest.fixture(scope="module")
def mf():
    import mi355_faiss

    return mi355_faiss


def _pair(mf, d, desc, metric, xb, ids=None):
    g, o = mf.index_factory(d, desc, metric), orc.Index(d, desc, metric)
    o.train(xb)
    g.ivf_set_centroids(o.ivf_centroids())
    for a in (g, o):
        if ids is None:
            a.add(xb)
        else:
            a.add_with_ids(xb, ids)
    return g, o


def _same(a, b):
    return np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("idmap", [False, True])
@pytest.mark.parametrize("d,nlist,n,nq,k,nprobe", [(128, 256, 200000, 1000, 10, 32), (64, 64, 60000, 300, 5, 16), (96, 128, 80000, 257, 20, 24),
                                                   (128, 64, 50000, 200, 1, 8), (32, 256, 100000, 400, 31, 64)])
def test_separated_clusters_prune_most_probes_and_change_nothing(mf, d, nlist, n, nq, k, nprobe, idmap):
    """mixture centres far apart against the spread inside a cluster (what embeddings look like, and BASELINE C3's synthetic rows):
    most of the nprobe lists of a query belong to other clusters and are provably irrelevant"""
    rs = np.random.RandomState(5)
    cent = rs.randn(nlist // 4, d).astype(np.float32) * 1.0
    xb = (cent[rs.randint(0, len(cent), n)] + 0.1 * rs.randn(n, d)).astype(np.float32)
    xq = (cent[rs.randint(0, len(cent), nq)] + 0.1 * rs.randn(nq, d)).astype(np.float32)
    xb[n // 2 :: 9] = xb[: len(xb[n // 2 :: 9])]  # duplicates: exact ties near the top
    xq[: nq // 8] = xb[3 : 3 + nq // 8]
    ids = np.arange(n, dtype=np.int64) * 5 + 7 if idmap else None
    desc = ("IDMap," if idmap else "") + f"IVF{nlist},Flat"
    g, o = _pair(mf, d, desc, L2, xb, ids)
    r1 = g.search(xq, k, nprobe=nprobe)
    assert g.last_kernel_info()["name"].startswith("ivf_bf16_collect")
    st = g.ivf_probe_stats()
    assert st["pairs"] == nq * nprobe and 0 < st["scanned"] < st["pairs"] // 2, st
    g.set_option("ivf_probe_prune", 0)
    r0 = g.search(xq, k, nprobe=nprobe)
    st0 = g.ivf_probe_stats()
    assert st0["scanned"] == st0["pairs"] == nq * nprobe
    ro = o.search(xq, k, nprobe=nprobe)
    assert _same(r1, r0), "pruning changed a result"
    assert _same(r1, ro), "differs from the oracle's IVF"
    # an IDSelector may reject the witnesses: no pruning, same answers as the oracle
    g.set_option("ivf_probe_prune", 1)
    keep = (ids if idmap else np.arange(n, dtype=np.int64))[::3]
    r2 = g.search(xq, k, nprobe=nprobe, sel=("batch", keep))
    st2 = g.ivf_probe_stats()
    assert st2["scanned"] == st2["pairs"], st2
    assert _same(r2, o.search(xq, k, nprobe=nprobe, sel=("batch", keep)))


def test_k_larger_than_the_nearest_lists_needs_more_witness_lists(mf):
    """tiny lists: the k witnesses span several probes; lists before the witness rank are never pruned"""
    d, nlist, n, nq, k, nprobe = 32, 512, 6000, 300, 30, 48  # ~12 rows per list
    rs = np.random.RandomState(11)
    cent = rs.randn(64, d).astype(np.float32) * 2.0
    xb = (cent[rs.randint(0, 64, n)] + 0.2 * rs.randn(n, d)).astype(np.float32)
    xq = (cent[rs.randint(0, 64, nq)] + 0.2 * rs.randn(nq, d)).astype(np.float32)
    g, o = _pair(mf, d, f"IVF{nlist},Flat", L2, xb)
    g.set_option("ivf_collect", 1)
    r1 = g.search(xq, k, nprobe=nprobe)
    assert g.last_kernel_info()["name"].startswith("ivf_bf16_collect")
    st = g.ivf_probe_stats()
    assert st["scanned"] < st["pairs"], st
    assert _same(r1, o.search(xq, k, nprobe=nprobe))
    # k beyond everything the probed lists hold: fewer than k witnesses -> nothing pruned, -1 padded like FAISS
    g2, o2 = _pair(mf, d, "IVF512,Flat", L2, xb[:700])
    g2.set_option("ivf_collect", 1)
    r2 = g2.search(xq, 30, nprobe=4)
    st2 = g2.ivf_probe_stats()
    ro2 = o2.search(xq, 30, nprobe=4)
    assert (ro2[1] < 0).any()
    assert _same(r2, ro2), st2


def test_integer_grid_ties_across_lists_survive_pruning(mf):
    """rows on a small integer grid with 30 % copies: most queries are tied at rank k across lists -- the pruning inequality is strict,
    a list holding a tied row is never pruned, and the tie pass still replays FAISS's arrival order over the WHOLE probe list"""
    d, nlist, n, k, nprobe = 32, 64, 40000, 10, 16
    rs = np.random.RandomState(3)
    cent = rs.randint(-20, 21, size=(16, d)).astype(np.float32)
    xb = cent[rs.randint(0, 16, n)] + rs.randint(-2, 3, size=(n, d)).astype(np.float32)
    dup = rs.rand(n) < 0.3
    xb[dup] = xb[rs.randint(0, n, size=int(dup.sum()))]
    xq = cent[rs.randint(0, 16, 300)] + rs.randint(-2, 3, size=(300, d)).astype(np.float32)
    xq[:100] = xb[rs.randint(0, n, size=100)]
    g, o = _pair(mf, d, f"IVF{nlist},Flat", L2, xb)
    r1 = g.search(xq, k, nprobe=nprobe)
    st = g.ivf_probe_stats()
    assert st["scanned"] < st["pairs"], st
    g.set_option("ivf_probe_prune", 0)
    r0 = g.search(xq, k, nprobe=nprobe)
    assert _same(r1, r0) and _same(r1, o.search(xq, k, nprobe=nprobe))


@pytest.mark.parametrize("metric", [L2, IP])
def test_uniform_rows_and_inner_product_prune_nothing_harmful(mf, metric):
    """uniform rows: hardly a list can be excluded; inner product: never pruned (no triangle inequality) -- results as before"""
    d, nlist, n, nq, k, nprobe = 64, 64, 60000, 256, 10, 16
    xb = orc.synth_uniform(n, d, 21)
    xq = orc.synth_uniform(nq, d, 22)
    g, o = _pair(mf, d, f"IVF{nlist},Flat", metric, xb)
    r1 = g.search(xq, k, nprobe=nprobe)
    st = g.ivf_probe_stats()
    if metric == IP:
        assert st["scanned"] == st["pairs"]
    assert _same(r1, o.search(xq, k, nprobe=nprobe)), st


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("k", [1, 10, 31])
def test_final_bound_filter_drops_candidates_not_results(mf, metric, k):
    """option ivf_cl_refilter (csrc/ivf_collect.hip ivf_refilter_kernel): candidates admitted under an early, loose bound are dropped when
    they do not pass the bound the scan ENDED with -- fewer rows re-scored, the same answers, exact ties included"""
    d, nlist, n, nq, nprobe = 64, 64, 80000, 500, 12
    xb = orc.synth_clustered(n, d, 41, n_centers=nlist, sigma=0.2)
    xq = orc.synth_clustered(nq, d, 42, n_centers=nlist, sigma=0.2)
    xb[n // 2 :: 5] = xb[: len(xb[n // 2 :: 5])]  # copies: ties at and around rank k
    xq[:60] = xb[7:67]
    g, o = _pair(mf, d, f"IVF{nlist},Flat", metric, xb)
    c0 = g.collect_stats()
    r1 = g.search(xq, k, nprobe=nprobe)
    assert g.last_kernel_info()["name"].startswith("ivf_bf16_collect")
    c1, st = g.collect_stats(), g.ivf_probe_stats()
    rescored = c1["candidates"] - c0["candidates"]
    assert 0 < rescored <= st["admitted"], (rescored, st)
    g.set_option("ivf_cl_refilter", 0)
    r0 = g.search(xq, k, nprobe=nprobe)
    c2 = g.collect_stats()
    assert c2["candidates"] - c1["candidates"] == g.ivf_probe_stats()["admitted"]
    assert c2["candidates"] - c1["candidates"] >= rescored
    assert _same(r1, r0) and _same(r1, o.search(xq, k, nprobe=nprobe))
