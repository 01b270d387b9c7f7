"""tests/faiss_format.py (the independent Python implementation of FAISS's on-disk layout) against bytes assembled by hand
from the published layout (faiss/impl/index_write.cpp [UPSTREAM]) and against itself.  CPU only."""
import struct

import numpy as np
import pytest

import faiss_format as ff
from oracle import oracle as orc


def test_flat_image_equals_hand_assembled_bytes():
    x = np.arange(6, dtype=np.float32).reshape(2, 3)
    raw = ff.dumps({"kind": "flat", "metric": ff.METRIC_L2, "x": x})
    want = b"IxF2" + struct.pack("<i", 3) + struct.pack("<q", 2) + struct.pack("<qq", 1 << 20, 1 << 20) + b"\x01" + struct.pack("<i", 1)
    want += struct.pack("<Q", 6) + x.tobytes()
    assert raw == want and len(raw) == 4 + 33 + 8 + 24
    assert ff.dumps({"kind": "flat", "metric": ff.METRIC_INNER_PRODUCT, "x": x})[:4] == b"IxFI"
    lp = ff.dumps({"kind": "flat", "metric": 4, "metric_arg": 3.0, "x": x})  # METRIC_Lp carries metric_arg
    assert lp[:4] == b"IxFl" and lp[4 + 33 : 4 + 37] == struct.pack("<f", 3.0)


def test_idmap_image_bytes():
    x = np.ones((3, 2), dtype=np.float32)
    raw = ff.dumps({"kind": "idmap", "metric": ff.METRIC_INNER_PRODUCT, "ids": [7, 8, 9], "sub": {"kind": "flat", "metric": ff.METRIC_INNER_PRODUCT, "x": x}})
    assert raw[:4] == b"IxMp" and raw[37:41] == b"IxFI"
    assert raw[-32:] == struct.pack("<Qqqq", 3, 7, 8, 9)
    back = ff.loads(raw)
    assert back["kind"] == "idmap" and back["ids"].tolist() == [7, 8, 9] and np.array_equal(back["sub"]["x"], x)


@pytest.mark.parametrize("sparse", [False, True])
def test_ivf_image_roundtrip_full_and_sparse_list_tables(sparse):
    d, nlist = 4, 8
    rs = np.random.RandomState(1)
    lists = []
    for l in range(nlist):
        n = 0 if (sparse and l not in (2, 5)) else int(rs.randint(1, 5))
        lists.append((rs.randint(0, 1000, size=n).astype(np.int64), rs.rand(n, d).astype(np.float32)))
    ix = {"kind": "ivfflat", "d": d, "metric": ff.METRIC_L2, "is_trained": True, "nprobe": 3, "lists": lists,
          "quantizer": {"kind": "flat", "metric": ff.METRIC_L2, "x": rs.rand(nlist, d).astype(np.float32)}}
    raw = ff.dumps(ix)
    assert raw[:4] == b"IwFl" and (b"sprs" if sparse else b"full") in raw and b"ilar" in raw
    back = ff.loads(raw)
    assert back["list_type"] == ("sprs" if sparse else "full") and back["nlist"] == nlist and back["nprobe"] == 3
    assert back["ntotal"] == sum(len(i) for i, _ in lists)
    for (i0, c0), (i1, c1) in zip(lists, back["lists"]):
        assert np.array_equal(i0, i1) and np.array_equal(c0, c1)
    with pytest.raises(ValueError):
        ff.loads(raw + b"\0")
    with pytest.raises(Exception):
        ff.loads(raw[:-3])


def test_hnsw_image_from_oracle_state_roundtrip():
    d, M, n = 8, 4, 200
    xb = orc.synth_uniform(n, d, 3)
    o = orc.Index(d, f"HNSW{M}", orc.METRIC_L2)
    o.add(xb)
    g = o.hnsw_graph()
    probas, cum = ff.hnsw_level_tables(M)
    assert cum[:3].tolist() == [0, 2 * M, 3 * M] and abs(probas.sum() - 1.0) < 1e-6
    graph = dict(g, assign_probas=probas, cum_nneighbor_per_level=cum, efConstruction=40, efSearch=16)
    raw = ff.dumps({"kind": "hnswflat", "metric": ff.METRIC_L2, "graph": graph, "storage": {"kind": "flat", "metric": ff.METRIC_L2, "x": xb}})
    assert raw[:4] == b"IHNf" and b"IxF2" in raw
    back = ff.loads(raw)
    assert back["graph"]["upper_beam"] == 1 and back["graph"]["entry_point"] == g["entry_point"]
    assert np.array_equal(back["graph"]["neighbors"], g["neighbors"]) and np.array_equal(back["graph"]["levels"], g["levels"])
    assert np.array_equal(back["graph"]["offsets"].astype(np.int64), g["offsets"]) and np.array_equal(back["storage"]["x"], xb)
