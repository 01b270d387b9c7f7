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
