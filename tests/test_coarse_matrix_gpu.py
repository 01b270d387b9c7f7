"""The IVF coarse quantiser's distance matrix (csrc/coarse_select.hip): three kernels, one set of bits.  IndexIVF::search asks its
quantizer -- an IndexFlat -- for the nprobe nearest centroids (faiss/IndexIVF.cpp search -> quantizer->search, reached from
src/faiss_extension.cpp:631); every coarse distance is IndexFlat's BLAS-branch value (xn + cn) - 2 <x, c> with a k-ordered f32 chain
for the inner product.  Option ivf_coarse_mfma = 0: vector-ALU chains; 1: v_mfma_f32_32x32x2_f32 with round 4's staging; 2 (default,
round 5): the same instruction with row-major LDS tiles, 128-byte-line loads issued a slab ahead.  All three must give the oracle's
lists -- shapes with ragged last tiles, slabs cut by d, d % 4 != 0 (falls back to kernel 1), both metrics, exact ties."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
L2, IP = orc.METRIC_L2, orc.METRIC_INNER_PRODUCT


@pytest.fixture(scope="module")
def mf():
    import mi355_faiss

    return mi355_faiss


def _same(a, b):
    return np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("d,nlist,n,nq,nprobe", [(128, 4096, 120000, 1000, 16), (128, 256, 30000, 129, 8), (36, 260, 20000, 300, 5),
                                                 (200, 1000, 40000, 257, 12), (64, 512, 30000, 128, 1), (16, 1024, 30000, 77, 32),
                                                 (130, 256, 20000, 200, 7), (96, 1500, 40000, 640, 40)])
def test_three_kernels_one_set_of_probe_lists(mf, metric, d, nlist, n, nq, nprobe):
    rs = np.random.RandomState(d * 7 + nlist)
    cent = rs.randn(nlist, d).astype(np.float32)
    xb = (cent[rs.randint(0, nlist, n)] + 0.3 * rs.randn(n, d)).astype(np.float32)
    xq = (cent[rs.randint(0, nlist, nq)] + 0.3 * rs.randn(nq, d)).astype(np.float32)
    xq[: nq // 4] = xb[: nq // 4]
    g, o = mf.index_factory(d, f"IVF{nlist},Flat", metric), orc.Index(d, f"IVF{nlist},Flat", metric)
    o.train(xb)
    g.ivf_set_centroids(o.ivf_centroids())
    g.add(xb)
    o.add(xb)
    k = 5
    ro = o.search(xq, k, nprobe=nprobe)
    # round 6: L2 quantisers of 16 < d <= 128, nlist % 16 == 0, nprobe <= 64 take csrc/coarse_bf16.hip (no matrix at all) -- same lists
    served = metric == L2 and 16 < d <= 128 and nlist % 16 == 0 and nprobe <= 64
    before = g.get_stat("coarse_bf16_queries")
    assert _same(g.search(xq, k, nprobe=nprobe), ro), ("bf16", d, nlist)
    assert g.get_stat("coarse_bf16_queries") - before == (nq if served else 0)
    g.set_option("ivf_coarse_bf16", 0)
    for mode in (2, 1, 0):
        g.set_option("ivf_coarse_mfma", mode)
        assert _same(g.search(xq, k, nprobe=nprobe), ro), (mode, d, nlist)


def test_integer_centroids_with_tied_coarse_distances(mf):
    """centroids on a small integer grid, queries ON grid points: many coarse distances are exactly equal, the (dis, id) order at the
    nprobe boundary decides which lists are scanned"""
    d, nlist, n, nq, nprobe = 32, 512, 40000, 300, 6
    rs = np.random.RandomState(9)
    cent = rs.randint(-2, 3, size=(nlist, d)).astype(np.float32)
    cent[1::2] = cent[0::2]  # pairs of IDENTICAL centroids: every coarse distance comes twice
    xb = (cent[rs.randint(0, nlist, n)] + rs.randint(-1, 2, size=(n, d))).astype(np.float32)
    xq = cent[rs.randint(0, nlist, nq)].copy()
    g, o = mf.index_factory(d, f"IVF{nlist},Flat", L2), orc.Index(d, f"IVF{nlist},Flat", L2)
    o.ivf_set_centroids(cent)
    g.ivf_set_centroids(o.ivf_centroids())
    g.add(xb)
    o.add(xb)
    ro = o.search(xq, 10, nprobe=nprobe)
    before = g.get_stat("coarse_bf16_queries")  # (add() assigns its rows through the same kernels, np = 1)
    assert _same(g.search(xq, 10, nprobe=nprobe), ro), "bf16"
    assert g.get_stat("coarse_bf16_queries") - before == nq
    g.set_option("ivf_coarse_bf16", 0)
    for mode in (2, 1, 0):
        g.set_option("ivf_coarse_mfma", mode)
        assert _same(g.search(xq, 10, nprobe=nprobe), ro), mode


@pytest.mark.parametrize("case", ["overflow", "nonfinite", "np64", "nearly_all"])
def test_bf16_coarse_quantiser_edge_cases(mf, case):
    """csrc/coarse_bf16.hip: a candidate list that overflows (every centroid the same point), a query whose bound is not finite, the
    largest served nprobe, nprobe close to nlist -- the exhaustive fallback inside the exact kernel must give the matrix path's lists."""
    d, nlist, n, nq, nprobe, k = 64, 1024, 40000, 200, 16, 4
    rs = np.random.RandomState(5)
    cent = rs.randn(nlist, d).astype(np.float32)
    if case == "overflow":
        cent[:] = cent[0]  # 1024 identical centroids: all tied, all candidates (> 512 per query)
        cent[::7] += 1e-3
    if case == "np64":
        nprobe = 64
    if case == "nearly_all":
        nlist, nprobe = 256, 60
        cent = cent[:nlist]
    xb = (cent[rs.randint(0, nlist, n)] + 0.3 * rs.randn(n, d)).astype(np.float32)
    xq = (cent[rs.randint(0, nlist, nq)] + 0.3 * rs.randn(nq, d)).astype(np.float32)
    if case == "nonfinite":
        xq[3] *= 1e19  # ||x||^2 overflows: no finite bound, FAISS's own distances are inf / nan for this query
    g, o = mf.index_factory(d, f"IVF{nlist},Flat", L2), orc.Index(d, f"IVF{nlist},Flat", L2)
    o.ivf_set_centroids(cent)
    g.ivf_set_centroids(o.ivf_centroids())
    g.add(xb)
    o.add(xb)
    before, ex0 = g.get_stat("coarse_bf16_queries"), g.get_stat("coarse_bf16_exhaustive")
    got = g.search(xq, k, nprobe=nprobe)
    assert g.get_stat("coarse_bf16_queries") - before == nq
    if case in ("overflow", "nonfinite"):
        assert g.get_stat("coarse_bf16_exhaustive") - ex0 >= 1
    g.set_option("ivf_coarse_bf16", 0)
    ref = g.search(xq, k, nprobe=nprobe)
    assert _same(got, ref), case
    if case != "nonfinite":
        assert _same(got, o.search(xq, k, nprobe=nprobe)), case
