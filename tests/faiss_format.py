"""An INDEPENDENT second implementation of FAISS's on-disk index layout (faiss/impl/index_write.cpp, index_read.cpp,
impl/io_macros.h [UPSTREAM]), in pure Python `struct` -- test infrastructure only.

csrc/index_io.hip is the product's reader/writer (faiss::read_index / write_index behind
/root/reference/src/faiss_extension.cpp:199,234; the Go harness loads `indices/IDMap,HNSW128,Flat.index` through it,
go/benches_c.go:59).  Until round 3 it was only ever compared with itself.  This module shares no code with it: images
written here from ORACLE state are loaded by the device library, files written by the device library are parsed here, and
both directions are compared field by field and through searches (tests/test_index_io_gpu.py, tests/test_faiss_format_cpu.py).
Still NOT verified against a file written by FAISS itself -- none exists in the reference or in this image.

Layout (little-endian):
  fourcc            4 ASCII bytes
  header            int32 d | int64 ntotal | int64 dummy = 1 << 20 (twice) | uint8 is_trained | int32 metric_type
                    [| float32 metric_arg   only when metric_type > 1]
  vector<T>         uint64 n | n x T
  IxF2 / IxFI / IxFl   header | uint64 n_floats | n_floats x float32         (IndexFlatL2 / IndexFlatIP / IndexFlat)
  IxMp / IxM2          header | <sub-index> | vector<int64> id_map          (IndexIDMap / IndexIDMap2)
  IwFl                 ivf header | inverted lists                           (IndexIVFFlat)
     ivf header        header | uint64 nlist | uint64 nprobe | <quantizer index> | direct map
     direct map        uint8 type (0 = NoMap) | vector<int64> array [| hashtable pairs when type == 2]
     inverted lists    "ilar" | uint64 nlist | uint64 code_size | "full" vector<uint64> sizes[nlist]
                                                              | "sprs" vector<uint64> (list_no, size) pairs
                       then per NON-EMPTY list: size x code_size bytes of codes, size x int64 ids
  IHNf                 header | HNSW | <storage index>                       (IndexHNSWFlat)
     HNSW              vector<float64> assign_probas | vector<int32> cum_nneighbor_per_level | vector<int32> levels |
                       vector<uint64> offsets | vector<int32> neighbors | int32 entry_point | int32 max_level |
                       int32 efConstruction | int32 efSearch | int32 upper_beam (= 1)
"""
import io
import struct

import numpy as np

METRIC_INNER_PRODUCT, METRIC_L2 = 0, 1
DUMMY = 1 << 20


# ---------------------------------------------------------------------------------------------------------------- writer
class _W:
    def __init__(self):
        self.b = io.BytesIO()

    def cc(self, s):
        assert len(s) == 4
        self.b.write(s.encode("ascii"))

    def pack(self, fmt, *v):
        self.b.write(struct.pack("<" + fmt, *v))

    def vec(self, arr, dtype):
        a = np.ascontiguousarray(arr, dtype=dtype)
        self.pack("Q", a.size)
        self.b.write(a.tobytes())


def _header(w, d, ntotal, is_trained, metric, metric_arg=0.0):
    w.pack("iqqqBi", d, ntotal, DUMMY, DUMMY, 1 if is_trained else 0, metric)
    if metric > 1:
        w.pack("f", metric_arg)


def _write(w, ix):
    k = ix["kind"]
    if k == "flat":
        x = np.ascontiguousarray(ix["x"], dtype=np.float32)
        n, d = x.shape
        w.cc({METRIC_INNER_PRODUCT: "IxFI", METRIC_L2: "IxF2"}.get(ix["metric"], "IxFl"))
        _header(w, d, n, True, ix["metric"], ix.get("metric_arg", 0.0))
        w.pack("Q", n * d)
        w.b.write(x.tobytes())
    elif k == "idmap":
        sub = ix["sub"]
        w.cc("IxM2" if ix.get("idmap2") else "IxMp")
        _header(w, sub["d"] if "d" in sub else sub["x"].shape[1], len(ix["ids"]), ix.get("is_trained", True), ix["metric"])
        _write(w, sub)
        w.vec(ix["ids"], np.int64)
    elif k == "ivfflat":
        d, lists = ix["d"], ix["lists"]  # lists: [(ids int64[n], codes float32[n, d])]
        ntotal = sum(len(i) for i, _ in lists)
        w.cc("IwFl")
        _header(w, d, ntotal, ix["is_trained"], ix["metric"])
        w.pack("QQ", len(lists), ix.get("nprobe", 1))
        _write(w, ix["quantizer"])
        w.pack("B", 0)  # DirectMap::NoMap
        w.vec([], np.int64)
        w.cc("ilar")
        w.pack("QQ", len(lists), d * 4)
        sizes = [len(i) for i, _ in lists]
        non0 = sum(1 for s in sizes if s)
        if non0 > len(lists) // 2:
            w.cc("full")
            w.vec(sizes, np.uint64)
        else:
            w.cc("sprs")
            w.vec([v for l, s in enumerate(sizes) if s for v in (l, s)], np.uint64)
        for ids, codes in lists:
            if len(ids):
                w.b.write(np.ascontiguousarray(codes, dtype=np.float32).tobytes())
                w.b.write(np.ascontiguousarray(ids, dtype=np.int64).tobytes())
    elif k == "hnswflat":
        g, st = ix["graph"], ix["storage"]
        w.cc("IHNf")
        _header(w, st["x"].shape[1], st["x"].shape[0], True, ix["metric"])
        w.vec(g["assign_probas"], np.float64)
        w.vec(g["cum_nneighbor_per_level"], np.int32)
        w.vec(g["levels"], np.int32)
        w.vec(g["offsets"], np.uint64)
        w.vec(g["neighbors"], np.int32)
        w.pack("iiiii", g["entry_point"], g["max_level"], g["efConstruction"], g["efSearch"], 1)
        _write(w, st)
    else:
        raise ValueError(k)


def dumps(ix):
    """index description (nested dicts, see loads) -> bytes of a .index file"""
    w = _W()
    _write(w, ix)
    return w.b.getvalue()


# ---------------------------------------------------------------------------------------------------------------- parser
class _R:
    def __init__(self, raw):
        self.raw, self.o = raw, 0

    def cc(self):
        s = self.raw[self.o : self.o + 4].decode("ascii")
        self.o += 4
        return s

    def unpack(self, fmt):
        v = struct.unpack_from("<" + fmt, self.raw, self.o)
        self.o += struct.calcsize("<" + fmt)
        return v

    def arr(self, n, dtype):
        a = np.frombuffer(self.raw, dtype=dtype, count=n, offset=self.o).copy()
        self.o += a.nbytes
        return a

    def vec(self, dtype):
        (n,) = self.unpack("Q")
        return self.arr(n, dtype)


def _read_header(r):
    d, ntotal, d1, d2, trained, metric = r.unpack("iqqqBi")
    if d1 != DUMMY or d2 != DUMMY:
        raise ValueError("header dummies %d %d" % (d1, d2))
    h = {"d": d, "ntotal": ntotal, "is_trained": bool(trained), "metric": metric}
    if metric > 1:
        (h["metric_arg"],) = r.unpack("f")
    return h


def _read(r):
    cc = r.cc()
    if cc in ("IxF2", "IxFI", "IxFl"):
        h = _read_header(r)
        (nf,) = r.unpack("Q")
        if nf != h["ntotal"] * h["d"]:
            raise ValueError("flat payload %d != %d x %d" % (nf, h["ntotal"], h["d"]))
        h.update(kind="flat", fourcc=cc, x=r.arr(nf, np.float32).reshape(h["ntotal"], h["d"]))
        return h
    if cc in ("IxMp", "IxM2"):
        h = _read_header(r)
        h.update(kind="idmap", fourcc=cc, idmap2=cc == "IxM2", sub=_read(r), ids=r.vec(np.int64))
        return h
    if cc == "IwFl":
        h = _read_header(r)
        nlist, nprobe = r.unpack("QQ")
        h.update(kind="ivfflat", fourcc=cc, nlist=nlist, nprobe=nprobe, quantizer=_read(r))
        (dm_type,) = r.unpack("B")
        dm = r.vec(np.int64)
        if dm_type != 0 or dm.size:
            raise ValueError("direct map type %d with %d entries" % (dm_type, dm.size))
        il = r.cc()
        if il != "ilar":
            raise ValueError("inverted lists fourcc " + il)
        nl2, code_size = r.unpack("QQ")
        if nl2 != nlist or code_size != 4 * h["d"]:
            raise ValueError("ilar nlist %d code_size %d" % (nl2, code_size))
        lt = r.cc()
        sizes = np.zeros(nlist, dtype=np.int64)
        if lt == "full":
            v = r.vec(np.uint64)
            if v.size != nlist:
                raise ValueError("full sizes")
            sizes[:] = v
        elif lt == "sprs":
            v = r.vec(np.uint64).reshape(-1, 2)
            sizes[v[:, 0].astype(np.int64)] = v[:, 1]
        else:
            raise ValueError("list type " + lt)
        h["list_type"] = lt
        lists = []
        for s in sizes:
            codes = r.arr(int(s) * h["d"], np.float32).reshape(int(s), h["d"])
            ids = r.arr(int(s), np.int64)
            lists.append((ids, codes))
        h["lists"] = lists
        return h
    if cc == "IHNf":
        h = _read_header(r)
        g = {
            "assign_probas": r.vec(np.float64),
            "cum_nneighbor_per_level": r.vec(np.int32),
            "levels": r.vec(np.int32),
            "offsets": r.vec(np.uint64),
            "neighbors": r.vec(np.int32),
        }
        g["entry_point"], g["max_level"], g["efConstruction"], g["efSearch"], g["upper_beam"] = r.unpack("iiiii")
        h.update(kind="hnswflat", fourcc=cc, graph=g, storage=_read(r))
        return h
    raise ValueError("fourcc " + repr(cc))


def loads(raw):
    """bytes of a .index file -> nested dicts: kind in {flat, idmap, ivfflat, hnswflat} + the fields of the layout above;
    raises ValueError on anything that does not follow it, including trailing bytes"""
    r = _R(raw)
    ix = _read(r)
    if r.o != len(raw):
        raise ValueError("%d trailing bytes" % (len(raw) - r.o))
    return ix


# ------------------------------------------------------------------------------------------- HNSW parameters FAISS derives
def hnsw_level_tables(M, n_levels_hint=None):
    """HNSW::set_default_probas(M, 1 / ln M): assign_probas[l] = exp(-l / mult) (1 - exp(-1 / mult)) while >= 1e-9;
    cum_nneighbor_per_level = [0, 2M, 3M, 4M, ...] (2M links at level 0, M above)"""
    mult = 1.0 / np.log(M)
    probas, cum, nn = [], [0], 0
    level = 0
    while True:
        p = np.exp(-level / mult) * (1 - np.exp(-1 / mult))
        if p < 1e-9:
            break
        probas.append(p)
        nn += 2 * M if level == 0 else M
        cum.append(nn)
        level += 1
    return np.array(probas, dtype=np.float64), np.array(cum, dtype=np.int32)
