"""ctypes wrapper around oracle/liborc.so -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The class mirrors the faiss.Index surface the reference reaches from
src/faiss_extension.cpp (:154 index_factory, :396/:583 train, :510/:607 add_with_ids,
:512/:609 add, :631 search) so parity tests read like the reference's own tests.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("ORC_LIB_PATH") or os.path.join(_HERE, "liborc.so")  # (override: the ASan/UBSan build of `make sanitize`)

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1
SEL_NONE, SEL_BITMAP, SEL_BATCH = 0, 1, 2
PATH_AUTO, PATH_PAIR, PATH_BLAS, PATH_OPENBLAS = 0, 1, 2, 3


class OracleError(RuntimeError):
    """Carries FAISS's exception text (the reference greps substrings of it)."""


class _Params(C.Structure):
    _fields_ = [
        ("nprobe", C.c_int64),
        ("efSearch", C.c_int64),
        ("sel_kind", C.c_int),
        ("sel_data", C.c_void_p),
        ("sel_n", C.c_int64),
        ("force_path", C.c_int),
    ]


def hnsw_set_pop_min_rule(rule):
    """tie rule of the search walk's MinimaxHeap::pop_min: 0 = FAISS's heap-array order (default), 1 = smallest id among equal
    minima (what the device walk does) -- oracle/orc_hnsw.c"""
    lib().orc_hnsw_set_pop_min_rule(int(rule))


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("orc_core.c", "orc_hnsw.c", "orc.h", "orc_internal.h", "Makefile")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liborc.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        p, i64, f32p, i64p = C.c_void_p, C.c_int64, C.POINTER(C.c_float), C.POINTER(C.c_int64)
        L.orc_last_error.restype = C.c_char_p
        L.orc_index_factory.restype = p
        L.orc_index_factory.argtypes = [C.c_int, C.c_char_p, C.c_int]
        L.orc_index_free.argtypes = [p]
        L.orc_d.argtypes = [p]
        L.orc_ntotal.argtypes = [p]
        L.orc_ntotal.restype = i64
        L.orc_is_trained.argtypes = [p]
        L.orc_metric.argtypes = [p]
        L.orc_train.argtypes = [p, i64, p]
        L.orc_add.argtypes = [p, i64, p]
        L.orc_add_with_ids.argtypes = [p, i64, p, p]
        L.orc_search.argtypes = [p, i64, p, i64, p, p, C.POINTER(_Params)]
        L.orc_ivf_nlist.argtypes = [p]
        L.orc_ivf_nlist.restype = i64
        L.orc_ivf_get_centroids.argtypes = [p, p]
        L.orc_ivf_set_centroids.argtypes = [p, p]
        L.orc_ivf_list_size.argtypes = [p, i64]
        L.orc_ivf_list_size.restype = i64
        L.orc_ivf_get_list.argtypes = [p, i64, p, p]
        L.orc_hnsw_set_ef_construction_ix.argtypes = [p, C.c_int]
        L.orc_hnsw_graph_size.argtypes = [p, C.POINTER(C.c_int), C.POINTER(C.c_int32)]
        L.orc_hnsw_graph_size.restype = i64
        L.orc_hnsw_get_graph.argtypes = [p, p, p, p]
        L.orc_hnsw_set_graph.argtypes = [p, i64, p, p, p, p, C.c_int32, C.c_int]
        L.orc_norms.argtypes = [p, i64, C.c_int, p]
        L.orc_flat_search.argtypes = [C.c_int, C.c_int, i64, p, i64, p, i64, p, p, C.POINTER(_Params), p]
        L.orc_flat_search_naive.argtypes = [C.c_int, C.c_int, i64, p, i64, p, i64, p, p, C.c_int]
        L.orc_merge_shards.argtypes = [C.c_int, i64, i64, C.c_int, p, p, p, p]
        L.orc_synth_uniform.argtypes = [p, i64, C.c_int, C.c_uint64, i64]
        L.orc_synth_clustered.argtypes = [p, i64, C.c_int, C.c_uint64, i64, C.c_int, C.c_float]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _check(rc):
    if rc:
        raise OracleError(lib().orc_last_error().decode())


def _mk_params(nprobe=0, efSearch=0, sel=None, force_path=PATH_AUTO):
    """sel: None | ("bitmap", uint8 array) | ("batch", int64 array)"""
    p = _Params()
    p.nprobe, p.efSearch, p.force_path = nprobe, efSearch, force_path
    keep = None
    if sel is not None:
        kind, data = sel
        if kind == "bitmap":
            keep = np.ascontiguousarray(data, dtype=np.uint8)
            p.sel_kind, p.sel_n = SEL_BITMAP, keep.size
        elif kind == "batch":
            keep = _i64(data)
            p.sel_kind, p.sel_n = SEL_BATCH, keep.size
        else:
            raise ValueError(kind)
        p.sel_data = keep.ctypes.data
    return p, keep


class Index:
    """CPU oracle index; same method names as faiss.Index."""

    def __init__(self, d, description, metric=METRIC_INNER_PRODUCT):
        # default metric INNER_PRODUCT: src/faiss_extension.cpp:105
        self._h = lib().orc_index_factory(int(d), description.encode(), int(metric))
        if not self._h:
            raise OracleError(lib().orc_last_error().decode())
        self.d = int(d)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_index_free(self._h)
            self._h = None

    @property
    def ntotal(self):
        return lib().orc_ntotal(self._h)

    @property
    def is_trained(self):
        return bool(lib().orc_is_trained(self._h))

    @property
    def metric_type(self):
        return lib().orc_metric(self._h)

    def train(self, x):
        x = _f32(x).reshape(-1, self.d)
        _check(lib().orc_train(self._h, x.shape[0], _ptr(x)))

    def add(self, x):
        x = _f32(x).reshape(-1, self.d)
        _check(lib().orc_add(self._h, x.shape[0], _ptr(x)))

    def add_with_ids(self, x, ids):
        x = _f32(x).reshape(-1, self.d)
        ids = _i64(ids)
        assert ids.size == x.shape[0]
        _check(lib().orc_add_with_ids(self._h, x.shape[0], _ptr(x), _ptr(ids)))

    def search(self, x, k, nprobe=0, efSearch=0, sel=None, force_path=PATH_AUTO):
        x = _f32(x).reshape(-1, self.d)
        nq = x.shape[0]
        D = np.empty((nq, max(k, 0)), dtype=np.float32)
        I = np.empty((nq, max(k, 0)), dtype=np.int64)
        p, keep = _mk_params(nprobe, efSearch, sel, force_path)
        _check(lib().orc_search(self._h, nq, _ptr(x), k, _ptr(D), _ptr(I), C.byref(p)))
        del keep
        return D, I

    # IVF introspection (parity tests share centroids with the device index)
    @property
    def nlist(self):
        return lib().orc_ivf_nlist(self._h)

    def ivf_centroids(self):
        out = np.empty((self.nlist, self.d), dtype=np.float32)
        _check(lib().orc_ivf_get_centroids(self._h, _ptr(out)))
        return out

    def ivf_set_centroids(self, c):
        c = _f32(c).reshape(self.nlist, self.d)
        _check(lib().orc_ivf_set_centroids(self._h, _ptr(c)))

    def ivf_list(self, list_no):
        n = lib().orc_ivf_list_size(self._h, list_no)
        ids = np.empty(n, dtype=np.int64)
        codes = np.empty((n, self.d), dtype=np.float32)
        _check(lib().orc_ivf_get_list(self._h, list_no, _ptr(ids), _ptr(codes)))
        return ids, codes


    # HNSW (src/faiss_extension.cpp:133-139 sets hnsw.efConstruction through the IDMap wrapper)
    def hnsw_set_ef_construction(self, v):
        _check(lib().orc_hnsw_set_ef_construction_ix(self._h, int(v)))

    def quantizer_hnsw_graph(self):
        """graph of the HNSW coarse quantizer of an "IVF<n>_HNSW<m>,Flat" index"""
        return self.hnsw_graph(n=self.nlist)

    def hnsw_graph(self, n=None):
        """-> dict(levels[n], offsets[n+1], neighbors[...], max_level, entry_point)"""
        ml, ep = C.c_int(0), C.c_int32(0)
        nb = lib().orc_hnsw_graph_size(self._h, C.byref(ml), C.byref(ep))
        if nb < 0:
            raise OracleError("not an HNSW index")
        n = self.ntotal if n is None else n
        levels = np.empty(n, dtype=np.int32)
        offsets = np.empty(n + 1, dtype=np.int64)
        neighbors = np.empty(nb, dtype=np.int32)
        _check(lib().orc_hnsw_get_graph(self._h, _ptr(levels), _ptr(offsets), _ptr(neighbors)))
        return dict(levels=levels, offsets=offsets, neighbors=neighbors, max_level=ml.value, entry_point=ep.value)


    def hnsw_set_graph(self, x, graph):
        """adopt rows + a graph dict as returned by hnsw_graph() (e.g. the device index's); search-only afterwards"""
        x = _f32(x).reshape(-1, self.d)
        lv = np.ascontiguousarray(graph["levels"], dtype=np.int32)
        of = _i64(graph["offsets"])
        nb = np.ascontiguousarray(graph["neighbors"], dtype=np.int32)
        assert lv.size == x.shape[0] and of.size == x.shape[0] + 1 and nb.size == of[-1]
        _check(lib().orc_hnsw_set_graph(self._h, x.shape[0], _ptr(x), _ptr(lv), _ptr(of), _ptr(nb),
                                        int(graph["entry_point"]), int(graph["max_level"])))


def norms(x):
    x = _f32(x)
    out = np.empty(x.shape[0], dtype=np.float32)
    lib().orc_norms(_ptr(x), x.shape[0], x.shape[1], _ptr(out))
    return out


def flat_search(metric, xb, xq, k, force_path=PATH_AUTO, sel=None, id_map=None):
    xb, xq = _f32(xb), _f32(xq)
    d = xb.shape[1]
    nq = xq.shape[0]
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    p, keep = _mk_params(sel=sel, force_path=force_path)
    idm = _i64(id_map) if id_map is not None else None
    _check(
        lib().orc_flat_search(
            metric, d, xb.shape[0], _ptr(xb), nq, _ptr(xq), k, _ptr(D), _ptr(I), C.byref(p), _ptr(idm) if idm is not None else None
        )
    )
    del keep
    return D, I


def flat_search_naive(metric, xb, xq, k, path):
    xb, xq = _f32(xb), _f32(xq)
    nq = xq.shape[0]
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    _check(lib().orc_flat_search_naive(metric, xb.shape[1], xb.shape[0], _ptr(xb), nq, _ptr(xq), k, _ptr(D), _ptr(I), path))
    return D, I


def merge_shards(metric, D, I):
    """D, I: [nshard, nq, k] with global labels -> merged [nq, k]"""
    D, I = _f32(D), _i64(I)
    ns, nq, k = D.shape
    Do = np.empty((nq, k), dtype=np.float32)
    Io = np.empty((nq, k), dtype=np.int64)
    lib().orc_merge_shards(metric, nq, k, ns, _ptr(D), _ptr(I), _ptr(Do), _ptr(Io))
    return Do, Io


def synth_uniform(n, d, seed, row0=0):
    out = np.empty((n, d), dtype=np.float32)
    lib().orc_synth_uniform(_ptr(out), n, d, seed, row0)
    return out


def synth_clustered(n, d, seed, row0=0, n_centers=1024, sigma=0.1):
    out = np.empty((n, d), dtype=np.float32)
    lib().orc_synth_clustered(_ptr(out), n, d, seed, row0, n_centers, sigma)
    return out


def num_threads():
    return lib().orc_num_threads()


def set_metric_arg(v):
    """Index::metric_arg (Lp exponent) for the metrics beyond L2 / inner product"""
    lib().orc_set_metric_arg(C.c_float(v))


def set_num_threads(n):
    lib().orc_set_num_threads(n)


def set_reservoir(on):
    """test hook: False = heaps at every k (FAISS: ReservoirTopN from k = 100 on)"""
    lib().orc_set_reservoir(1 if on else 0)


# ---- FAISS's BLAS branch on the real OpenBLAS (PATH_OPENBLAS): the independent reference for label stability ----------
def openblas_path():
    """numpy's bundled OpenBLAS with the 64-bit-integer scipy prefix (0.3.29 = /root/reference/vcpkg_ports/openblas/vcpkg.json:3)"""
    import glob

    cands = sorted(glob.glob(os.path.join(os.path.dirname(np.__file__), "..", "numpy.libs", "libscipy_openblas64_*.so")))
    return os.path.abspath(cands[0]) if cands else None


def openblas_load(path=None):
    """-> OpenBLAS's config string (version, core, threading); raises OracleError when no such library exists"""
    path = path or openblas_path()
    if path is None:
        raise OracleError("no libscipy_openblas64_*.so under numpy.libs")
    L = lib()
    L.orc_openblas_load.argtypes = [C.c_char_p]
    L.orc_openblas_config.restype = C.c_char_p
    _check(L.orc_openblas_load(path.encode()))
    return L.orc_openblas_config().decode()


def openblas_set_num_threads(n):
    lib().orc_openblas_set_num_threads(int(n))


def openblas_census(metric, xb, xq, k, D_dev, I_dev, ref=None):
    """Compare a device result (labels I_dev, values D_dev; [nq, k]) with FAISS's BLAS branch summed by the REAL OpenBLAS sgemm
    (search at k + 1, so that the gap behind the last slot is known) and classify every (query, rank) slot.

    Rounding band: an f32 inner product of d terms, summed in ANY order, deviates from the real value by at most
    gamma_d |x|.|y| <= d u ||x|| ||y|| (u = 2^-24); with the norms (same bound each) and the two roundings of (xn + yn) - 2 ip a
    computed L2 distance is within E = (d + 2) u (||x|| + ||y||)^2 of the real one, an inner product within d u ||x|| ||y||.
    Two correct implementations can therefore rank two rows differently only if their REAL values are within 2 E of each other.
    -> dict(slots, slots_label_differs, queries_label_differs, differing_slots_inside_band (must equal slots_label_differs),
            fragile_adjacent_pairs (OpenBLAS neighbours in rank closer than 2 E: the slots that CAN flip), max_value_rel_diff)"""
    xb, xq = _f32(xb), _f32(xq)
    nq, d = xq.shape
    if ref is None:  # (bench.py hands over the k + 1 run it timed)
        openblas_load()
        ref = flat_search(metric, xb, xq, k + 1, force_path=PATH_OPENBLAS)
    Dob, Iob = ref
    I_dev = np.asarray(I_dev)[:nq]
    D_dev = np.asarray(D_dev)[:nq]
    u = 2.0**-24
    xnorm = np.sqrt((xq.astype(np.float64) ** 2).sum(1))
    ymax = float(np.sqrt(float(norms(xb).max())))  # (f32 chain norms: the band is not sensitive to their last bits)
    if metric == METRIC_L2:
        E = (d + 2) * u * (xnorm + ymax) ** 2
    else:
        E = d * u * xnorm * ymax
    diff = I_dev != Iob[:, :k]
    qs, rs = np.nonzero(diff)
    inside = 0
    for q, r in zip(qs, rs):
        a, b = int(I_dev[q, r]), int(Iob[q, r])
        if a < 0 or b < 0:
            continue
        x64 = xq[q].astype(np.float64)
        if metric == METRIC_L2:
            ta = float(((x64 - xb[a].astype(np.float64)) ** 2).sum())
            tb = float(((x64 - xb[b].astype(np.float64)) ** 2).sum())
        else:
            ta = float((x64 * xb[a].astype(np.float64)).sum())
            tb = float((x64 * xb[b].astype(np.float64)).sum())
        inside += abs(ta - tb) <= 2.0 * E[q]
    gaps = np.abs(Dob[:, 1:].astype(np.float64) - Dob[:, :-1].astype(np.float64))  # [nq, k]: slot r vs r + 1 (the last: k vs k + 1)
    valid = Iob[:, 1:] >= 0
    fragile = int(((gaps <= 2.0 * E[:, None]) & valid).sum())
    denom = np.maximum(np.abs(Dob[:, :k]).astype(np.float64), 1e-30)
    rel = np.abs(D_dev.astype(np.float64) - Dob[:, :k].astype(np.float64)) / denom
    return {
        "reference": "FAISS BLAS branch (4096 x 1024 blocks, norms formula, CMax/CMin heaps) on OpenBLAS sgemm",
        "queries": int(nq),
        "slots": int(nq * k),
        "slots_label_differs": int(diff.sum()),
        "queries_label_differs": int(diff.any(1).sum()),
        "differing_slots_inside_band": int(inside),
        "fragile_adjacent_pairs": fragile,
        "band": "real values within 2E; E = (d+2) u (|x|+|y|max)^2 (L2) | d u |x| |y|max (IP), u = 2^-24",
        "max_value_rel_diff": float(rel[~diff].max()) if (~diff).any() else None,
    }
