"""Seeded differential fuzz: random shapes / parameters / selectors through the C ABI vs the CPU oracle.  Complements
the hand-picked cases of the other GPU tests with odd dimensions, tiny and ragged N, k > N, every selector kind and
IDMap wrapping, for all three index families.  Labels and distances must be bit-identical (IP rows whose k-th and
(k+1)-th scores tie are skipped: FAISS's own IP tie membership is arrival-order dependent)."""
import numpy as np
import pytest

from oracle import oracle as orc

import os

pytestmark = pytest.mark.gpu
# MVS_FUZZ_SCALE=4 runs four times as many seeds (one-off soak runs)
_SCALE = int(os.environ.get("MVS_FUZZ_SCALE", "1"))
L2, IP = orc.METRIC_L2, orc.METRIC_INNER_PRODUCT


@pytest.fixture(scope="module")
def mf():
    import mi355_faiss

    return mi355_faiss


def _selector(rs, ids):
    kind = rs.randint(3)
    if kind == 0 or len(ids) == 0:
        return None
    keep = ids[rs.rand(len(ids)) < rs.choice([0.05, 0.5, 0.95])]
    if kind == 1:
        return ("batch", keep.astype(np.int64))
    bm = np.zeros(int(ids.max()) // 8 + 1, dtype=np.uint8)
    for i in keep:
        bm[i >> 3] |= 1 << (i & 7)
    return ("bitmap", bm)


@pytest.mark.parametrize("seed", range(24 * _SCALE))
def test_flat_fuzz(mf, seed):
    rs = np.random.RandomState(1000 + seed)
    d = int(rs.choice([1, 3, 8, 17, 32, 64, 100, 128, 130, 200, 300]))
    n = int(rs.choice([1, 7, 63, 64, 65, 500, 2049, 5000]))
    nq = int(rs.choice([1, 2, 5, 19, 20, 21, 33, 70]))
    k = int(rs.choice([1, 2, 10, 16, 17, 40, 64]))
    metric = [L2, IP][rs.randint(2)]
    idmap = bool(rs.randint(2))
    xb = (rs.rand(n, d).astype(np.float32) - 0.5) * rs.choice([1.0, 30.0])
    xq = (rs.rand(nq, d).astype(np.float32) - 0.5) * 2
    if n > 10 and rs.randint(2):
        xb[n // 2] = xb[0]  # a duplicate row: exact distance ties
    ids = (rs.permutation(4 * n)[:n] + 3).astype(np.int64) if idmap else np.arange(n, dtype=np.int64)
    desc = "IDMap,Flat" if idmap else "Flat"
    g, o = mf.index_factory(d, desc, metric), orc.Index(d, desc, metric)
    for a in (g, o):
        a.add_with_ids(xb, ids) if idmap else a.add(xb)
    sel = _selector(rs, ids)
    D, I = g.search(xq, k, sel=sel)
    Do, Io = o.search(xq, k, sel=sel)
    what = f"flat seed={seed} d={d} n={n} nq={nq} k={k} metric={metric} idmap={idmap} sel={sel and sel[0]}"
    assert np.array_equal(I, Io), what
    assert np.array_equal(D.view(np.uint32), Do.view(np.uint32)), what


@pytest.mark.parametrize("seed", range(10 * _SCALE))
def test_ivf_fuzz(mf, seed):
    rs = np.random.RandomState(2000 + seed)
    d = int(rs.choice([4, 8, 16, 20, 33, 64, 96, 128, 200]))
    nlist = int(rs.choice([1, 2, 8, 32]))
    n = int(rs.choice([nlist * 40, 3000, 9000]))
    nq = int(rs.choice([1, 7, 25, 150]))
    k = int(rs.choice([1, 5, 10, 33]))
    nprobe = int(rs.choice([1, 2, nlist, nlist + 3]))
    metric = [L2, IP][rs.randint(2)]
    idmap = bool(rs.randint(2))
    xb = orc.synth_clustered(n, d, 300 + seed, n_centers=max(4, nlist), sigma=0.2)
    xq = orc.synth_clustered(nq, d, 400 + seed, n_centers=max(4, nlist), sigma=0.2)
    ids = (rs.permutation(3 * n)[:n] + 11).astype(np.int64)
    desc = ("IDMap," if idmap else "") + f"IVF{nlist},Flat"
    g, o = mf.index_factory(d, desc, metric), orc.Index(d, desc, metric)
    o.train(xb)
    g.ivf_set_centroids(o.ivf_centroids())
    for a in (g, o):
        a.add_with_ids(xb, ids)  # IndexIVF implements add_with_ids itself
    sel = _selector(rs, ids)
    D, I = g.search(xq, k, nprobe=nprobe, sel=sel)
    Do, Io = o.search(xq, k, nprobe=nprobe, sel=sel)
    what = f"ivf seed={seed} d={d} nlist={nlist} n={n} nq={nq} k={k} nprobe={nprobe} metric={metric} sel={sel and sel[0]}"
    assert np.array_equal(I, Io), what  # every query: exact ties follow FAISS's heap (csrc/ivf_ties.hip)
    assert np.array_equal(D.view(np.uint32), Do.view(np.uint32)), what


@pytest.mark.parametrize("seed", range(10 * _SCALE))
def test_hnsw_fuzz(mf, seed):
    rs = np.random.RandomState(3000 + seed)
    d = int(rs.choice([2, 5, 16, 31, 64, 100, 260, 520]))
    M = int(rs.choice([2, 4, 8, 16, 32, 48]))
    n = int(rs.choice([1, 2, 40, 700, 2500]))
    nq = int(rs.choice([1, 9, 80]))
    k = int(rs.choice([1, 10, 37]))
    efc = int(rs.choice([8, 40, 90]))
    efs = int(rs.choice([1, 16, 50, 200]))
    metric = [L2, IP][rs.randint(2)]
    idmap = bool(rs.randint(2))
    xb = (rs.rand(n, d).astype(np.float32) - 0.3)
    xq = (rs.rand(nq, d).astype(np.float32) - 0.3)
    ids = (rs.permutation(5 * n)[:n] + 1).astype(np.int64)
    desc = ("IDMap," if idmap else "") + f"HNSW{M}"
    g, o = mf.index_factory(d, desc, metric), orc.Index(d, desc, metric)
    g.set_option("hnsw_build_waves", 1)
    g.set_ef_construction(efc)
    o.hnsw_set_ef_construction(efc)
    chunk = int(rs.choice([n, 333, 2048]))
    for i0 in range(0, n, max(1, chunk)):
        for a in (g, o):
            a.add_with_ids(xb[i0 : i0 + chunk], ids[i0 : i0 + chunk]) if idmap else a.add(xb[i0 : i0 + chunk])
    what = f"hnsw seed={seed} d={d} M={M} n={n} nq={nq} k={k} efC={efc} efS={efs} metric={metric} idmap={idmap}"
    ga, gb = o.hnsw_graph(), g.hnsw_graph()
    assert np.array_equal(ga["neighbors"], gb["neighbors"]) and ga["entry_point"] == gb["entry_point"], what
    sel = _selector(rs, ids if idmap else np.arange(n, dtype=np.int64))
    D, I = g.search(xq, k, efSearch=efs, sel=sel)
    Do, Io = o.search(xq, k, efSearch=efs, sel=sel)
    assert np.array_equal(I, Io), what
    assert np.array_equal(D.view(np.uint32), Do.view(np.uint32)), what
