"""Every BASELINE.json configuration at FULL size through the C ABI, checked against the oracle.

  H   FlatL2 d=128 N=10M nq=10k k=10                      (headline; src/faiss_extension.cpp:631 -> IndexFlat::search)
  C2  FlatL2 d=128 N=1M  nq=10k k=10
  C3  IVF4096,Flat d=128 N=10M nprobe=32 nq=10k k=10      (train :583, add :609, search :631 with SearchParametersIVF)
  C4  FlatIP d=768 N=100M row-sharded over 8 GPUs         -> the shard ONE GPU holds (N=12.5M, global labels), and the
                                                             same rows as 8 virtual shards merged by mvs_merge_shards
  C5  IDMap,HNSW32 d=768 N=1M efSearch=128 nq=10k k=10

The oracle runs on a query sample where a full batch would take minutes on the host (sample sizes in the tests); on
ALL queries the size-independent properties are checked (FAISS output order, labels in range and distinct, self
queries, shard merge == unsharded).  Rows come from the counter-based generators, identical on host and device.
"""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
L2, IP = orc.METRIC_L2, orc.METRIC_INNER_PRODUCT
DB_SEED, Q_SEED = 1234, 4321
SLAB = 1 << 20


@pytest.fixture(scope="module")
def mf():
    import mi355_faiss

    return mi355_faiss


@pytest.fixture(scope="module")
def torch():
    import torch as t

    return t


def _check_order_and_range(D, I, lo, hi, is_l2):
    """FAISS output contract on every query: heap_reorder order, labels inside the shard's id range, no repeats."""
    assert I.min() >= lo and I.max() < hi
    d = np.diff(D, axis=1)
    assert (d >= 0).all() if is_l2 else (d <= 0).all()
    s = np.sort(I, axis=1)
    assert (np.diff(s, axis=1) != 0).all(), "a label repeats inside one result row"
    assert np.isfinite(D).all()


def _check_vs_openblas(metric, xb_h, xq_h, k, D, I, ns):
    """The INDEPENDENT label reference: FAISS's BLAS branch summed by the real OpenBLAS sgemm (0.3.29, the
    reference's vcpkg pin; numpy bundles it) instead of the oracle's k-ordered chain.  sgemm sums in its own order, so labels may
    differ at rounding-level near-ties: every differing (query, rank) slot must sit inside the rounding band, and values agree to
    1e-5 relative (north_star asks 1e-4)."""
    if orc.openblas_path() is None:
        pytest.skip("no OpenBLAS with the scipy 64-bit prefix on this host")
    orc.openblas_load()
    orc.openblas_set_num_threads(16)  # (FAISS's 1024-row sgemm blocks: 64 threads are 3x slower than 16 on the box's host)
    cen = orc.openblas_census(metric, xb_h, xq_h[:ns], k, D[:ns], I[:ns])
    print("openblas census:", cen)
    assert cen["differing_slots_inside_band"] == cen["slots_label_differs"], cen
    assert cen["slots_label_differs"] <= cen["fragile_adjacent_pairs"], cen  # only fragile slots can flip
    assert cen["max_value_rel_diff"] is not None and cen["max_value_rel_diff"] <= 1e-5, cen
    return cen


def _build_flat_uniform(mf, torch, n, d, metric, desc="Flat"):
    ix = mf.index_factory(d, desc, metric)
    for s0 in range(0, n, SLAB):
        m = min(SLAB, n - s0)
        ix.add_torch(mf.synth_uniform_torch(m, d, DB_SEED, row0=s0))
        torch.cuda.synchronize()
    return ix


@pytest.mark.parametrize("metric", [L2, IP])
def test_headline_store_with_lists_of_1000(mf, torch, metric):
    """round 6 (the bench line's H_k1000): the headline store asked for 1000 rows per query -- the list length of the reference's
    post-filter use (README.md:222-271, go/main_test.go:26-32) -- stays on the bf16 filter (range bounds, frozen scan, segmented sort);
    the exact kernels give the same rows and bits on a sample, and so does the oracle (FAISS's reservoir from k = 100 on)"""
    n, d, nq, k = 10_000_000, 128, 2048, 1000
    ix = _build_flat_uniform(mf, torch, n, d, metric)
    xq = mf.synth_uniform_torch(nq, d, Q_SEED)
    D, I = ix.search_torch(xq, k)
    torch.cuda.synchronize()
    assert ix.last_kernel_info()["name"] == "flat_bf16_collect_kernel"
    cs = ix.collect_stats()
    assert cs["queries"] == nq and k * nq <= cs["candidates"] <= 40 * k * nq, cs
    assert ix.prefilter_stats()["fallback_queries"] == 0
    D, I = D.cpu().numpy(), I.cpu().numpy()
    ns = 48
    ix.set_option("cl_bigk", 0)
    Dx, Ix = ix.search_torch(xq[:ns].contiguous(), k)
    torch.cuda.synchronize()
    assert ix.last_kernel_info()["name"] != "flat_bf16_collect_kernel"
    assert np.array_equal(Ix.cpu().numpy(), I[:ns]) and np.array_equal(Dx.cpu().numpy().view(np.uint32), D[:ns].view(np.uint32))
    xb_h = orc.synth_uniform(n, d, DB_SEED)
    Do, Io = orc.flat_search(metric, xb_h, xq[:ns].cpu().numpy(), k, force_path=orc.PATH_BLAS)
    assert np.array_equal(I[:ns], Io), "labels differ from the oracle"
    assert np.array_equal(D[:ns].view(np.uint32), Do.view(np.uint32)), "distances differ from the oracle"


def test_headline_flat_l2_10m(mf, torch):
    n, d, nq, k = 10_000_000, 128, 10_000, 10
    ix = _build_flat_uniform(mf, torch, n, d, L2)
    xq = mf.synth_uniform_torch(nq, d, Q_SEED)
    D, I = ix.search_torch(xq, k)
    torch.cuda.synchronize()
    # default path at this size: bf16 coarse filter with a proven bound + exact f32 re-scoring (csrc/flat_collect.hip)
    assert ix.last_kernel_info()["name"] == "flat_bf16_collect_kernel"
    cs = ix.collect_stats()
    assert cs["queries"] == nq and cs["overflows"] == 0 and k * nq <= cs["candidates"] <= 1000 * nq, cs
    assert ix.prefilter_stats()["fallback_queries"] == 0
    D, I = D.cpu().numpy(), I.cpu().numpy()
    # ... and the exact f32 MFMA kernel on the same index gives the same answers, bit for bit, on all 10k queries
    ix.set_option("prefilter", 0)
    Dx, Ix = ix.search_torch(xq, k)
    torch.cuda.synchronize()
    assert ix.last_kernel_info()["name"].startswith("flat_mfma")
    assert np.array_equal(Ix.cpu().numpy(), I) and np.array_equal(Dx.cpu().numpy().view(np.uint32), D.view(np.uint32))
    # ... and so does the bf16x3 prefilter (three bf16 products per pair, top-k' lists + proof: csrc/flat_bf16.hip)
    ix.set_option("prefilter", 1)
    D3, I3 = ix.search_torch(xq, k)
    torch.cuda.synchronize()
    assert ix.last_kernel_info()["name"] == "flat_bf16x3_kernel"
    st = ix.prefilter_stats()
    assert st["fallback_queries"] <= 10 and st["max_rel_err"] < st["err_bound"] / 5, st
    assert np.array_equal(I3.cpu().numpy(), I) and np.array_equal(D3.cpu().numpy().view(np.uint32), D.view(np.uint32))
    ix.set_option("prefilter", -1)
    _check_order_and_range(D, I, 0, n, True)
    ns = 256  # 0.65 TFLOP on the host
    xb_h = orc.synth_uniform(n, d, DB_SEED)
    Do, Io = orc.flat_search(L2, xb_h, xq[:ns].cpu().numpy(), k, force_path=orc.PATH_BLAS)
    assert np.array_equal(I[:ns], Io), "labels differ from the oracle"
    assert np.array_equal(D[:ns].view(np.uint32), Do.view(np.uint32)), "distances differ from the oracle"
    _check_vs_openblas(L2, xb_h, xq.cpu().numpy(), k, D, I, 1024)
    # the DuckDB granularity (<= 2048 queries per call, :903-925) returns the same rows
    D2, I2 = ix.search_torch(xq[2048:4096].contiguous(), k)
    torch.cuda.synchronize()
    assert np.array_equal(I2.cpu().numpy(), I[2048:4096]) and np.array_equal(D2.cpu().numpy(), D[2048:4096])
    # distances within 1e-4 relative of float64 truth (north_star's tolerance) on the sampled queries
    xq64 = xq[:32].cpu().numpy().astype(np.float64)
    for q in range(32):
        rows = xb_h[I[q]].astype(np.float64)
        truth = ((rows - xq64[q]) ** 2).sum(axis=1)
        np.testing.assert_allclose(D[q], truth, rtol=1e-4)


def test_c2_flat_l2_1m_full_batch(mf, torch):
    n, d, nq, k = 1_000_000, 128, 10_000, 10
    ix = _build_flat_uniform(mf, torch, n, d, L2)
    xq = mf.synth_uniform_torch(nq, d, Q_SEED)
    D, I = ix.search_torch(xq, k)
    torch.cuda.synchronize()
    D, I = D.cpu().numpy(), I.cpu().numpy()
    _check_order_and_range(D, I, 0, n, True)
    xb_h = orc.synth_uniform(n, d, DB_SEED)
    ns = 2048
    Do, Io = orc.flat_search(L2, xb_h, xq[:ns].cpu().numpy(), k, force_path=orc.PATH_BLAS)
    assert np.array_equal(I[:ns], Io) and np.array_equal(D[:ns].view(np.uint32), Do.view(np.uint32))
    _check_vs_openblas(L2, xb_h, xq.cpu().numpy(), k, D, I, 4096)


def test_c3_ivf4096_10m_nprobe32(mf, torch):
    n, d, nq, k, nprobe = 10_000_000, 128, 10_000, 10, 32
    gen = lambda m, seed, row0=0: mf.synth_clustered_torch(m, d, seed, row0=row0, n_centers=1024, sigma=0.1)
    xb = gen(n, DB_SEED)
    torch.cuda.synchronize()
    xb_h = xb.cpu().numpy()
    ix = mf.index_factory(d, "IVF4096,Flat", L2)
    ix.train(xb_h)  # the reference trains on ALL rows it was given (:583)
    for s0 in range(0, n, SLAB):
        ix.add_torch(xb[s0 : s0 + SLAB])
    torch.cuda.synchronize()
    assert ix.is_trained and ix.ntotal == n
    xq = gen(nq, Q_SEED)
    D, I = ix.search_torch(xq, k, nprobe=nprobe)
    torch.cuda.synchronize()
    # default path at this size: bf16 coarse filter on residual rows + exact scanner-arithmetic re-scoring (csrc/ivf_collect.hip)
    assert ix.last_kernel_info()["name"].startswith("ivf_bf16_collect")
    D, I = D.cpu().numpy(), I.cpu().numpy()
    _check_order_and_range(D, I, 0, n, True)
    # ... and the scanner kernel on the same index gives the same distances and labels on all 10k queries
    ix.set_option("ivf_collect", 0)
    Ds, Is = ix.search_torch(xq, k, nprobe=nprobe)
    torch.cuda.synchronize()
    assert ix.last_kernel_info()["name"].startswith("ivf_scan_kernel")
    Ds, Is = Ds.cpu().numpy(), Is.cpu().numpy()
    assert np.array_equal(Ds.view(np.uint32), D.view(np.uint32))
    assert np.array_equal(Is, I)  # labels on all 10k queries, exact ties included (csrc/ivf_ties.hip)
    ix.set_option("ivf_collect", -1)
    # oracle IVF sharing the trained centroids, 512 queries, bit-exact
    o = orc.Index(d, "IVF4096,Flat", L2)
    o.ivf_set_centroids(ix.ivf_centroids())
    o.add(xb_h)
    ns = 512
    Do, Io = o.search(xq[:ns].cpu().numpy(), k, nprobe=nprobe)
    assert np.array_equal(I[:ns], Io), "IVF labels differ from the oracle"
    assert np.array_equal(D[:ns].view(np.uint32), Do.view(np.uint32)), "IVF distances differ from the oracle"
    # round 6 (the bench line's C3_k100): lists beyond the scan's class slots stay on the bf16 filter (collect_search_big) -- the scanner
    # kernel gives the same rows and bits, and so does the oracle
    xq2 = xq[:2048].contiguous()
    D100, I100 = ix.search_torch(xq2, 100, nprobe=nprobe)
    torch.cuda.synchronize()
    assert ix.last_kernel_info()["name"].startswith("ivf_bf16_collect")
    D100, I100 = D100.cpu().numpy(), I100.cpu().numpy()
    ix.set_option("ivf_cl_big", 0)
    Dsc, Isc = ix.search_torch(xq2[:256].contiguous(), 100, nprobe=nprobe)
    torch.cuda.synchronize()
    assert ix.last_kernel_info()["name"].startswith("ivf_scan_kernel")
    ix.set_option("ivf_cl_big", 1)
    assert np.array_equal(Dsc.cpu().numpy().view(np.uint32), D100[:256].view(np.uint32)) and np.array_equal(Isc.cpu().numpy(), I100[:256])
    Do100, Io100 = o.search(xq[:32].cpu().numpy(), 100, nprobe=nprobe)
    assert np.array_equal(I100[:32], Io100) and np.array_equal(D100[:32].view(np.uint32), Do100.view(np.uint32))
    # recall@10 against exact search on the same rows
    flat = mf.index_factory(d, "Flat", L2)
    for s0 in range(0, n, SLAB):
        flat.add_torch(xb[s0 : s0 + SLAB])
    nr = 1000
    _, Igt = flat.search_torch(xq[:nr].contiguous(), k)
    torch.cuda.synchronize()
    Igt = Igt.cpu().numpy()
    recall = np.mean([len(set(a.tolist()) & set(b.tolist())) / k for a, b in zip(I[:nr], Igt)])
    assert recall >= 0.99, recall


def _normalised(mf, torch, m, d, seed, row0):
    x = mf.synth_clustered_torch(m, d, seed, row0=row0, n_centers=1024, sigma=1.0)
    x /= x.norm(dim=1, keepdim=True)
    return x


def test_c5_idmap_hnsw32_768_1m(mf, torch):
    n, d, nq, k, ef = 1_000_000, 768, 10_000, 10, 128
    ix = mf.index_factory(d, "IDMap,HNSW32", L2)
    xb_h = np.empty((n, d), dtype=np.float32)
    slab = 1 << 16
    for s0 in range(0, n, slab):
        m = min(slab, n - s0)
        xb = _normalised(mf, torch, m, d, DB_SEED, s0)
        ids = torch.arange(s0, s0 + m, dtype=torch.int64, device=xb.device) * 2 + 1  # external ids != row numbers
        ix.add_torch(xb, ids=ids)
        torch.cuda.synchronize()
        xb_h[s0 : s0 + m] = xb.cpu().numpy()
    assert ix.ntotal == n
    xq = _normalised(mf, torch, nq, d, Q_SEED, 0)
    D, I = ix.search_torch(xq, k, efSearch=ef)
    torch.cuda.synchronize()
    D, I = D.cpu().numpy(), I.cpu().numpy()
    assert (I % 2 == 1).all() and I.min() >= 1 and I.max() < 2 * n
    assert (np.diff(D, axis=1) >= 0).all()
    # the oracle walks the graph the device built (FAISS's own multi-thread build is not reproducible either):
    # labels AND distances bit-exact on the whole batch
    o = orc.Index(d, "HNSW32", L2)
    o.hnsw_set_graph(xb_h, ix.hnsw_graph())
    Do, Io = o.search(xq.cpu().numpy(), k, efSearch=ef)
    assert np.array_equal((I - 1) // 2, Io), "HNSW labels differ from the oracle walking the same graph"
    assert np.array_equal(D.view(np.uint32), Do.view(np.uint32))
    # recall@10 vs exact search (FAISS's default efConstruction = 40 graph; DESIGN.md 5 explains the 0.8)
    flat = mf.index_factory(d, "Flat", L2)
    for s0 in range(0, n, slab * 4):
        flat.add(xb_h[s0 : s0 + slab * 4])
    nr = 1000
    _, Igt = flat.search(xq[:nr].cpu().numpy(), k)
    recall = np.mean([len(set(a.tolist()) & set(b.tolist())) / k for a, b in zip((I[:nr] - 1) // 2, Igt)])
    assert recall >= 0.75, recall


def test_c4_flat_ip_768_one_gpu_shard_of_100m(mf, torch):
    """What ONE of the 8 GPUs holds at C4: rows [87.5M, 100M) of the 100M x 768 database (global labels through
    label_offset), nq=10k.  Oracle: block-wise over the same rows (host memory stays at one block) + the host merge."""
    N, G, d, nq, k = 100_000_000, 8, 768, 10_000, 10
    r0, r1 = N * (G - 1) // G, N
    n = r1 - r0
    blk = n // 8
    assert blk * 8 == n
    ix = mf.index_factory(d, "Flat", IP)
    ix.set_label_offset(r0)
    shards = []
    xq = _normalised(mf, torch, nq, d, Q_SEED, 0)
    xq_h = xq.cpu().numpy()
    ns = 64
    Dblk, Iblk = [], []
    for b in range(8):
        s0 = r0 + b * blk
        sh = mf.index_factory(d, "Flat", IP)
        sh.set_label_offset(s0)
        for t0 in range(s0, s0 + blk, 1 << 19):
            m = min(1 << 19, s0 + blk - t0)
            xb = _normalised(mf, torch, m, d, DB_SEED, t0)
            ix.add_torch(xb)
            sh.add_torch(xb)
            torch.cuda.synchronize()
        shards.append(sh)
        # oracle on this block, 64 queries (the block is regenerated on the device and copied out: normalisation is
        # a device op, so the host sees exactly the rows the index holds)
        xb_h = np.empty((blk, d), dtype=np.float32)
        for t0 in range(0, blk, 1 << 19):
            m = min(1 << 19, blk - t0)
            xb_h[t0 : t0 + m] = _normalised(mf, torch, m, d, DB_SEED, s0 + t0).cpu().numpy()
        Do, Io = orc.flat_search(IP, xb_h, xq_h[:ns], k, force_path=orc.PATH_BLAS)
        Dblk.append(Do)
        Iblk.append(Io + s0)
        del xb_h
    assert ix.ntotal == n
    D, I = ix.search_torch(xq, k)
    torch.cuda.synchronize()
    # the bf16 coarse filter with the k dimension split over wave pairs (csrc/flat_collect_wide.hip) serves this shape ...
    assert ix.last_kernel_info()["name"] == "flat_bf16_big_kernel", ix.last_kernel_info()
    D, I = D.cpu().numpy(), I.cpu().numpy()
    _check_order_and_range(D, I, r0, r1, False)
    # ... with the answers of the exact f32 kernel on all 10k queries, bit for bit
    ix.set_option("prefilter", 0)
    D0, I0 = ix.search_torch(xq, k)
    torch.cuda.synchronize()
    assert ix.last_kernel_info()["name"] == "flat_mfma_kernel"
    assert np.array_equal(I0.cpu().numpy(), I) and np.array_equal(D0.cpu().numpy().view(np.uint32), D.view(np.uint32))
    ix.set_option("prefilter", -1)
    Dor, Ior = orc.merge_shards(IP, np.stack(Dblk), np.stack(Iblk))
    assert np.array_equal(I[:ns], Ior), "labels differ from the oracle"
    assert np.array_equal(D[:ns].view(np.uint32), Dor.view(np.uint32))
    # 8 virtual shards of N/8 rows, merged with the host k-way merge == the unsharded search, on all 10k queries
    Ds, Is = [], []
    for sh in shards:
        Dj, Ij = sh.search_torch(xq, k)
        torch.cuda.synchronize()
        Ds.append(Dj.cpu().numpy())
        Is.append(Ij.cpu().numpy())
    Dm, Im = mf.merge_shards(IP, np.stack(Ds), np.stack(Is))
    assert np.array_equal(Im, I) and np.array_equal(Dm.view(np.uint32), D.view(np.uint32))
