"""Flat L2 on rows that CLUSTER (round 5): the index keeps an IVF index of its rows as a shadow (csrc/index.hip
FlatIndex::shadow_search) -- per-list centring in the coarse filter, the Flat arithmetic in the re-scoring, and a proof per query
that the unprobed lists cannot hold a better row.  The answers must be the Flat index's, bit for bit: labels AND distances equal to
the exact f32 kernel and to the oracle (the reference reaches this through Index::search, src/faiss_extension.cpp:631)."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
L2 = orc.METRIC_L2


@pytest.fixture(scope="module")
def mf():
    import mi355_faiss

    return mi355_faiss


def _same(a, b, what):
    assert np.array_equal(a[1], b[1]), what + ": labels"
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), what + ": distances"


def test_clustered_rows_take_the_shadow_and_stay_bit_exact(mf):
    import torch

    d, n, nq, k = 128, 1_048_576, 2048, 10
    dev = torch.device("cuda", 0)
    ix = mf.index_factory(d, "Flat", L2)
    xb = mf.synth_clustered_torch(n, d, 1234, row0=0, n_centers=1024, sigma=0.1, device=dev)
    ix.add_torch(xb)
    xq = mf.synth_clustered_torch(nq, d, 4321, row0=0, n_centers=1024, sigma=0.1, device=dev)
    D0, I0 = ix.search_torch(xq, k)  # the global-centring coarse filter: ~ a cluster's worth of candidates per query -> the shadow is wanted
    torch.cuda.synchronize()
    assert ix.last_kernel_info()["name"].startswith("flat_bf16_collect")
    assert ix.collect_stats()["candidates"] > 600 * nq
    D1, I1 = ix.search_torch(xq, k)  # builds the shadow, answers through it
    torch.cuda.synchronize()
    assert "flat shadow" in ix.last_kernel_info()["name"], ix.last_kernel_info()
    D2, I2 = ix.search_torch(xq, k)
    torch.cuda.synchronize()
    assert "flat shadow" in ix.last_kernel_info()["name"]
    assert torch.equal(I1, I0) and torch.equal(D1.view(torch.int32), D0.view(torch.int32))
    assert torch.equal(I2, I0) and torch.equal(D2.view(torch.int32), D0.view(torch.int32))
    # ... and the oracle on a query sample (BLAS branch: nq >= 20)
    xb_h = orc.synth_clustered(n, d, 1234, n_centers=1024, sigma=0.1)
    xq_h = xq[:256].cpu().numpy()
    _same((D2[:256].cpu().numpy(), I2[:256].cpu().numpy()), orc.flat_search(L2, xb_h, xq_h, k, force_path=orc.PATH_BLAS), "shadow vs oracle")
    # the exact f32 kernel
    ix.set_option("prefilter", 0)
    De, Ie = ix.search_torch(xq[:512].contiguous(), k)
    torch.cuda.synchronize()
    assert ix.last_kernel_info()["name"].startswith("flat_mfma")
    assert torch.equal(Ie, I2[:512]) and torch.equal(De.view(torch.int32), D2[:512].view(torch.int32))


@pytest.mark.parametrize("kind", ["uniform", "dups"])
def test_rows_that_do_not_cluster_give_the_shadow_up_and_stay_exact(mf, kind):
    """Forced on (option flat_shadow = 1) over uniform rows no unprobed list can be excluded: every query is unproven, the batch takes
    the normal path, the shadow is dropped for good.  Clustered rows with many exact duplicates (ties at the k-th distance inside and
    across lists) stay on the shadow and must order ties by row number as the Flat heap does."""
    d, n, nq, k = 64, 300_000, 600, 10
    if kind == "uniform":
        xb = orc.synth_uniform(n, d, 7)
        xq = orc.synth_uniform(nq, d, 8)
    else:
        xb = orc.synth_clustered(n, d, 7, n_centers=256, sigma=0.05)
        xb = np.round(xb * 8) / 8  # a coarse grid: many rows coincide exactly
        xq = xb[np.random.RandomState(3).randint(0, n, nq)].copy()
    ix = mf.index_factory(d, "Flat", L2)
    ix.add(xb)
    ix.set_option("flat_shadow", 1)
    ref = orc.flat_search(L2, xb, xq, k, force_path=orc.PATH_BLAS)
    for rep in range(3):
        got = ix.search(xq, k)
        _same(got, ref, f"{kind} rep {rep}")
    name = ix.last_kernel_info()["name"]
    if kind == "uniform":
        assert "flat shadow" not in name, name
    else:
        assert "flat shadow" in name, name


def test_shadow_is_extended_by_add_instead_of_rebuilt(mf):
    """round 6: rows added after the shadow was built are APPENDED to it at the next large search (one
    k-means per index, not per add/search cycle); the answers stay the Flat index's bit for bit; the stats say what happened."""
    import torch

    d, n0, step, nq, k = 128, 1_048_576, 65_536, 1024, 10
    dev = torch.device("cuda", 0)
    ix = mf.index_factory(d, "Flat", L2)
    ix.set_option("flat_shadow", 1)  # from the first large search on
    xb = mf.synth_clustered_torch(n0 + 3 * step, d, 1234, row0=0, n_centers=1024, sigma=0.1, device=dev)
    xq = mf.synth_clustered_torch(nq, d, 4321, row0=0, n_centers=1024, sigma=0.1, device=dev)
    ix.add_torch(xb[:n0])
    ix.search_torch(xq, k)
    torch.cuda.synchronize()
    st = ix.shadow_stats()
    assert st["state"] >= 0 and st["rows"] == n0 and st["builds"] == 1 and st["extends"] == 0 and st["device_bytes"] > n0 * d * 2, st
    assert st["build_seconds"] > 0
    for i in range(3):  # DuckDB: insert, then query
        ix.add_torch(xb[n0 + i * step : n0 + (i + 1) * step])
        D, I = ix.search_torch(xq, k)
        torch.cuda.synchronize()
        assert "flat shadow" in ix.last_kernel_info()["name"], ix.last_kernel_info()
        st = ix.shadow_stats()
        assert st["builds"] == 1 and st["extends"] == i + 1 and st["rows"] == n0 + (i + 1) * step, st
    assert int(I.max()) >= n0, "rows added after the build must be found"
    ix.set_option("prefilter", 0)
    ix.set_option("flat_shadow", 0)
    De, Ie = ix.search_torch(xq, k)
    torch.cuda.synchronize()
    assert ix.last_kernel_info()["name"].startswith("flat_mfma")
    assert torch.equal(Ie, I) and torch.equal(De.view(torch.int32), D.view(torch.int32))


def test_a_batch_beyond_the_coarse_matrix_is_served_in_pieces_and_keeps_the_shadow(mf):
    """one batch the coarse quantiser's distance matrix does not cover used to drop a just-built shadow for good."""
    import torch

    d, n, k = 64, 262_144, 10
    dev = torch.device("cuda", 0)
    ix = mf.index_factory(d, "Flat", L2)
    ix.set_option("flat_shadow", 1)
    ix.set_option("flat_shadow_nprobe", 64)
    xb = mf.synth_clustered_torch(n, d, 77, row0=0, n_centers=512, sigma=0.1, device=dev)
    ix.add_torch(xb)
    nq = 262_144 + 5_000  # nlist 512: the 512 MB distance matrix covers 262 144 queries -> two pieces
    xq = mf.synth_clustered_torch(nq, d, 78, row0=0, n_centers=512, sigma=0.1, device=dev)
    D, I = ix.search_torch(xq, k)
    torch.cuda.synchronize()
    assert "flat shadow" in ix.last_kernel_info()["name"], ix.last_kernel_info()
    st = ix.shadow_stats()
    assert st["state"] >= 0 and st["rows"] == n and st["queries"] == nq, st
    ix.set_option("prefilter", 0)
    ix.set_option("flat_shadow", 0)
    for q0 in (0, 262_144 - 512, nq - 1024):
        De, Ie = ix.search_torch(xq[q0 : q0 + 1024].contiguous(), k)
        torch.cuda.synchronize()
        assert torch.equal(Ie, I[q0 : q0 + 1024]) and torch.equal(De.view(torch.int32), D[q0 : q0 + 1024].view(torch.int32))
