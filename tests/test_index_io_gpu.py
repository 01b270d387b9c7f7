"""faiss::write_index / read_index (src/faiss_extension.cpp:199,234) on the MI355X path: round trips through FAISS's
on-disk layout (restated in csrc/index_io.hip; no FAISS-written file exists here to pin byte compatibility) and the
layout's fixed points (fourcc, header fields) checked directly in the bytes."""
import os
import struct

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
L2, IP = orc.METRIC_L2, orc.METRIC_INNER_PRODUCT


@pytest.fixture(scope="module")
def mf():
    import mi355_faiss

    return mi355_faiss


def _roundtrip(mf, ix, tmp_path, name):
    p = str(tmp_path / name)
    mf.write_index(ix, p)
    return mf.read_index(p), p


def test_flat_roundtrip_and_bytes(mf, tmp_path):
    xb, xq = orc.synth_uniform(1000, 24, 1), orc.synth_uniform(40, 24, 2)
    ix = mf.index_factory(24, "Flat", L2)
    ix.add(xb)
    ld, p = _roundtrip(mf, ix, tmp_path, "flat.index")
    raw = open(p, "rb").read()
    assert raw[:4] == b"IxF2"
    d, ntotal, dum1, dum2, trained, metric = struct.unpack_from("<iqqqBi", raw, 4)
    assert (d, ntotal, dum1, dum2, trained, metric) == (24, 1000, 1 << 20, 1 << 20, 1, 1)
    (nfloat,) = struct.unpack_from("<Q", raw, 4 + 33)
    assert nfloat == 1000 * 24 and len(raw) == 4 + 33 + 8 + nfloat * 4
    assert np.array_equal(np.frombuffer(raw, dtype=np.float32, offset=4 + 33 + 8).reshape(1000, 24), xb)
    assert ld.ntotal == 1000 and ld.d == 24 and ld.metric_type == L2 and ld.kind == mf.KIND_FLAT
    D0, I0 = ix.search(xq, 10)
    D1, I1 = ld.search(xq, 10)
    assert np.array_equal(I0, I1) and np.array_equal(D0, D1)
    ip = mf.index_factory(24, "Flat", IP)
    _, p2 = _roundtrip(mf, ip, tmp_path, "flat_ip.index")
    assert open(p2, "rb").read()[:4] == b"IxFI"


def test_idmap_roundtrip(mf, tmp_path):
    xb, xq = orc.synth_uniform(500, 8, 3), orc.synth_uniform(10, 8, 4)
    ids = np.arange(500, dtype=np.int64) * 11 + 7
    ix = mf.index_factory(8, "IDMap,Flat", IP)
    ix.add_with_ids(xb, ids)
    ld, p = _roundtrip(mf, ix, tmp_path, "idmap.index")
    raw = open(p, "rb").read()
    assert raw[:4] == b"IxMp" and raw[37:41] == b"IxFI"
    assert np.array_equal(np.frombuffer(raw[-500 * 8 :], dtype=np.int64), ids)
    assert ld.kind == mf.KIND_IDMAP and ld.ntotal == 500
    D0, I0 = ix.search(xq, 5)
    D1, I1 = ld.search(xq, 5)
    assert np.array_equal(I0, I1) and np.array_equal(D0, D1)
    ld.add_with_ids(xb[:3] * 2, np.array([9001, 9002, 9003]))  # a loaded index keeps working
    assert ld.ntotal == 503


@pytest.mark.parametrize("desc", ["IVF16,Flat", "IDMap,IVF16,Flat"])
def test_ivf_roundtrip(mf, tmp_path, desc):
    xb = orc.synth_clustered(6000, 32, 5, n_centers=32, sigma=0.2)
    xq = orc.synth_clustered(100, 32, 6, n_centers=32, sigma=0.2)
    ix = mf.index_factory(32, desc, L2)
    ix.train(xb)
    if desc.startswith("IDMap"):
        ix.add_with_ids(xb, np.arange(6000, dtype=np.int64) + 50000)
    else:
        ix.add(xb)
    ld, p = _roundtrip(mf, ix, tmp_path, "ivf.index")
    raw = open(p, "rb").read()
    assert b"IwFl" in raw[:48] and b"ilar" in raw and b"full" in raw
    assert ld.is_trained and ld.ntotal == 6000
    for nprobe in (1, 8):
        D0, I0 = ix.search(xq, 10, nprobe=nprobe)
        D1, I1 = ld.search(xq, 10, nprobe=nprobe)
        assert np.array_equal(I0, I1) and np.array_equal(D0, D1)


def test_ivf_sparse_lists(mf, tmp_path):
    """fewer than nlist/2 non-empty lists -> the 'sprs' size encoding"""
    xb = orc.synth_clustered(600, 16, 7, n_centers=4, sigma=0.05)
    ix = mf.index_factory(16, "IVF16,Flat", L2)
    ix.train(xb)
    ix.add(xb[:3])
    ld, p = _roundtrip(mf, ix, tmp_path, "ivf_sparse.index")
    assert b"sprs" in open(p, "rb").read()
    D0, I0 = ix.search(xb[:5], 3, nprobe=16)
    D1, I1 = ld.search(xb[:5], 3, nprobe=16)
    assert np.array_equal(I0, I1) and np.array_equal(D0, D1)


@pytest.mark.parametrize("desc,metric", [("HNSW16", L2), ("IDMap,HNSW8,Flat", IP)])
def test_hnsw_roundtrip(mf, tmp_path, desc, metric):
    xb, xq = orc.synth_uniform(3000, 20, 8), orc.synth_uniform(64, 20, 9)
    ix = mf.index_factory(20, desc, metric)
    ix.set_ef_construction(50)
    if desc.startswith("IDMap"):
        ix.add_with_ids(xb, np.arange(3000, dtype=np.int64) * 2)
    else:
        ix.add(xb)
    ld, p = _roundtrip(mf, ix, tmp_path, "hnsw.index")
    raw = open(p, "rb").read()
    assert b"IHNf" in raw[:48]
    g0, g1 = ix.hnsw_graph(), ld.hnsw_graph()
    for key in ("levels", "offsets", "neighbors"):
        assert np.array_equal(g0[key], g1[key])
    assert g0["entry_point"] == g1["entry_point"] and g0["max_level"] == g1["max_level"]
    D0, I0 = ix.search(xq, 10, efSearch=64)
    D1, I1 = ld.search(xq, 10, efSearch=64)
    assert np.array_equal(I0, I1) and np.array_equal(D0, D1)
    # the oracle reads the same structure: same graph -> same answers
    o = orc.Index(20, desc.replace("IDMap,", "").replace(",Flat", ""), metric)
    o.hnsw_set_graph(xb, g1)
    Do, Io = o.search(xq, 10, efSearch=64)
    lab = I1 // 2 if desc.startswith("IDMap") else I1
    assert np.array_equal(lab, Io) and np.array_equal(D1.view(np.uint32), Do.view(np.uint32))


def test_errors(mf, tmp_path):
    with pytest.raises(mf.FaissException, match="could not open"):
        mf.read_index(str(tmp_path / "missing.index"))
    p = tmp_path / "junk.index"
    p.write_bytes(b"ABCD" + b"\\0" * 64)
    with pytest.raises(mf.FaissException, match="not recognized"):
        mf.read_index(str(p))
    p.write_bytes(b"IxF2" + b"\\0" * 10)
    with pytest.raises(mf.FaissException, match="read error"):
        mf.read_index(str(p))


def test_crafted_hnsw_image_is_rejected(mf, tmp_path):
    """A neighbour id / entry point outside [0, ntotal) would make the walk kernels read out of bounds: read_index must
    refuse the file (FAISS-style read error) instead of building a device graph from it."""
    xb = orc.synth_uniform(400, 12, 21)
    ix = mf.index_factory(12, "HNSW8", L2)
    ix.add(xb)
    p = str(tmp_path / "h.index")
    mf.write_index(ix, p)
    raw = bytearray(open(p, "rb").read())
    g = ix.hnsw_graph()
    nb = g["neighbors"].astype(np.int32)
    pos = raw.find(nb[:64].tobytes())  # the neighbour table inside the file
    assert pos > 0
    first = int(np.flatnonzero(nb >= 0)[0])
    bad = bytearray(raw)
    bad[pos + 4 * first : pos + 4 * first + 4] = struct.pack("<i", 400)  # one link past the last vertex
    q = tmp_path / "bad_link.index"
    q.write_bytes(bytes(bad))
    with pytest.raises(mf.FaissException, match="neighbour slot"):
        mf.read_index(str(q))
    # entry point: stored right after the neighbour table (HNSW::entry_point, then max_level, ...)
    ep_pos = pos + 4 * len(nb)
    assert struct.unpack_from("<i", raw, ep_pos)[0] == g["entry_point"]
    bad = bytearray(raw)
    bad[ep_pos : ep_pos + 4] = struct.pack("<i", 100000)
    q = tmp_path / "bad_entry.index"
    q.write_bytes(bytes(bad))
    with pytest.raises(mf.FaissException, match="entry point"):
        mf.read_index(str(q))


# ---- two implementations, both directions: tests/faiss_format.py is a pure-Python restatement of the layout
# that shares no code with csrc/index_io.hip; images it writes from ORACLE state must load and search like the oracle, files
# the device library writes must parse into the state that was added.  (Not a check against a FAISS-written file: none exists.)
import faiss_format as ff


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("idmap", [False, True])
def test_python_written_flat_image_loads_and_searches_like_the_oracle(mf, tmp_path, metric, idmap):
    d, n = 20, 3000
    xb, xq = orc.synth_uniform(n, d, 5), orc.synth_uniform(50, d, 6)
    ids = (np.random.RandomState(1).permutation(5 * n)[:n] + 9).astype(np.int64)
    img = {"kind": "flat", "metric": metric, "x": xb}
    if idmap:
        img = {"kind": "idmap", "metric": metric, "ids": ids, "sub": img}
    p = tmp_path / "py_flat.index"
    p.write_bytes(ff.dumps(img))
    g = mf.read_index(str(p))
    assert g.ntotal == n and g.d == d and g.metric_type == metric and g.kind == (mf.KIND_IDMAP if idmap else mf.KIND_FLAT)
    D, I = g.search(xq, 10)
    Do, Io = orc.flat_search(metric, xb, xq, 10, id_map=ids if idmap else None)
    assert np.array_equal(I, Io) and np.array_equal(_bits(D), _bits(Do))
    # ... and what the device writes back is, byte for byte, the image the Python writer made
    p2 = str(tmp_path / "dev_flat.index")
    mf.write_index(g, p2)
    assert open(p2, "rb").read() == p.read_bytes()


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("sparse", [False, True])
def test_ivf_images_cross_both_ways(mf, tmp_path, metric, sparse):
    d, nlist, n = 16, 32 if sparse else 8, 4000
    xb = orc.synth_clustered(n, d, 21, n_centers=4 if sparse else 16, sigma=0.05 if sparse else 0.2)
    xq = orc.synth_clustered(60, d, 22, n_centers=4 if sparse else 16, sigma=0.05 if sparse else 0.2)
    ids = (np.random.RandomState(2).permutation(3 * n)[:n] + 1).astype(np.int64)
    o = orc.Index(d, f"IVF{nlist},Flat", metric)
    if sparse:  # k-means on wide data (it leaves no centroid without points), rows from four tight clusters only: most lists stay empty
        o.train(orc.synth_clustered(n, d, 23, n_centers=64, sigma=0.3))
    else:
        o.train(xb)
    o.add_with_ids(xb, ids)
    lists = [o.ivf_list(l) for l in range(nlist)]
    if sparse:
        assert sum(1 for i, _ in lists if len(i)) <= nlist // 2  # the "sprs" size table
    img = {"kind": "ivfflat", "d": d, "metric": metric, "is_trained": True, "nprobe": 1, "lists": lists,
           "quantizer": {"kind": "flat", "metric": metric if metric == IP else L2, "x": o.ivf_centroids()}}
    p = tmp_path / "py_ivf.index"
    p.write_bytes(ff.dumps(img))
    g = mf.read_index(str(p))  # Python image -> device
    assert g.ntotal == n and g.is_trained and g.nlist == nlist
    for nprobe in (1, 5):
        D, I = g.search(xq, 10, nprobe=nprobe)
        Do, Io = o.search(xq, 10, nprobe=nprobe)
        assert np.array_equal(I, Io) and np.array_equal(_bits(D), _bits(Do)), nprobe
    g2 = mf.index_factory(d, f"IVF{nlist},Flat", metric)  # device-built -> file -> Python parser
    g2.ivf_set_centroids(o.ivf_centroids())
    g2.add_with_ids(xb, ids)
    p2 = str(tmp_path / "dev_ivf.index")
    mf.write_index(g2, p2)
    back = ff.loads(open(p2, "rb").read())
    assert back["kind"] == "ivfflat" and back["nlist"] == nlist and back["ntotal"] == n and back["metric"] == metric
    assert back["list_type"] == ("sprs" if sparse else "full")
    assert np.array_equal(_bits(back["quantizer"]["x"]), _bits(o.ivf_centroids()))
    for (i0, c0), (i1, c1) in zip(lists, back["lists"]):
        assert np.array_equal(i0, i1) and np.array_equal(_bits(c0), _bits(c1))


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("idmap", [False, True])
def test_hnsw_images_cross_both_ways(mf, tmp_path, metric, idmap):
    d, M, n = 24, 8, 1500
    xb, xq = orc.synth_uniform(n, d, 31) - 0.3, orc.synth_uniform(40, d, 32) - 0.3
    ids = (np.random.RandomState(3).permutation(4 * n)[:n] + 5).astype(np.int64)
    o = orc.Index(d, ("IDMap," if idmap else "") + f"HNSW{M}", metric)
    o.add_with_ids(xb, ids) if idmap else o.add(xb)
    go = o.hnsw_graph()
    probas, cum = ff.hnsw_level_tables(M)
    graph = dict(go, assign_probas=probas, cum_nneighbor_per_level=cum, efConstruction=40, efSearch=16)
    img = {"kind": "hnswflat", "metric": metric, "graph": graph, "storage": {"kind": "flat", "metric": metric, "x": xb}}
    if idmap:
        img = {"kind": "idmap", "metric": metric, "ids": ids, "sub": dict(img, d=d)}
    p = tmp_path / "py_hnsw.index"
    p.write_bytes(ff.dumps(img))
    g = mf.read_index(str(p))  # the oracle's graph through the Python writer into the device walk
    assert g.ntotal == n
    gg = g.hnsw_graph()
    assert np.array_equal(gg["neighbors"], go["neighbors"]) and gg["entry_point"] == go["entry_point"] and gg["max_level"] == go["max_level"]
    for efs in (16, 64):
        D, I = g.search(xq, 10, efSearch=efs)
        Do, Io = o.search(xq, 10, efSearch=efs)
        assert np.array_equal(I, Io) and np.array_equal(_bits(D), _bits(Do)), efs
    g2 = mf.index_factory(d, ("IDMap," if idmap else "") + f"HNSW{M}", metric)  # device-built (single-wave order) -> Python parser
    g2.set_option("hnsw_build_waves", 1)
    g2.add_with_ids(xb, ids) if idmap else g2.add(xb)
    p2 = str(tmp_path / "dev_hnsw.index")
    mf.write_index(g2, p2)
    back = ff.loads(open(p2, "rb").read())
    h = back["sub"] if idmap else back
    if idmap:
        assert back["kind"] == "idmap" and np.array_equal(back["ids"], ids)
    assert h["kind"] == "hnswflat" and h["metric"] == metric and h["ntotal"] == n
    bg = h["graph"]
    assert np.array_equal(bg["neighbors"], go["neighbors"]) and np.array_equal(bg["levels"], go["levels"])
    assert np.array_equal(bg["offsets"].astype(np.int64), go["offsets"]) and bg["entry_point"] == go["entry_point"]
    assert np.allclose(bg["assign_probas"], probas, rtol=1e-12) and np.array_equal(bg["cum_nneighbor_per_level"], cum)
    assert (bg["efConstruction"], bg["upper_beam"]) == (40, 1) and np.array_equal(_bits(h["storage"]["x"]), _bits(xb))
