"""ctypes host binding of libmi355faiss.so (include/mi355_faiss.h).

Mirrors the faiss.Index surface the reference DuckDB extension reaches from
/root/reference/src/faiss_extension.cpp (:154 index_factory, :396/:583 train, :510/:607
add_with_ids, :512/:609 add, :631 search, gpu.cpp:48 index_cpu_to_gpu) with the same names,
argument meaning and error text, so the parity tests read like the reference's tests.

There is NO CPU fallback: if the HIP library is missing this module raises at import, and
every call fails loudly when no gfx950 device is usable.
"""
import ctypes as C
import os
import sys

import numpy as np

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1
KIND_FLAT, KIND_IDMAP, KIND_IVFFLAT, KIND_HNSW = 1, 2, 3, 4
SEL_NONE, SEL_BITMAP, SEL_BATCH = 0, 1, 2

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("MVS_LIB_PATH") or os.path.join(_PKG, "libmi355faiss.so")  # (override: A/B of two builds)


class FaissException(RuntimeError):
    """faiss::FaissException -- .msg carries the text the reference greps (src/faiss_extension.cpp:400,523,592)"""

    @property
    def msg(self):
        return self.args[0]


class SearchParams(C.Structure):
    _fields_ = [
        ("nprobe", C.c_int64),
        ("efSearch", C.c_int64),
        ("sel_kind", C.c_int32),
        ("reserved", C.c_int32),
        ("sel_data", C.c_void_p),
        ("sel_n", C.c_int64),
    ]


class KernelInfo(C.Structure):
    _fields_ = [
        ("name", C.c_char * 64),
        ("flops", C.c_double),
        ("bytes", C.c_double),
        ("last_ms", C.c_double),
        ("grid", C.c_int32),
        ("block", C.c_int32),
        ("lds_bytes", C.c_int32),
        ("nsplit", C.c_int32),
    ]


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950); the MI355X vector-search path has no CPU fallback"
    )


def _preload_hip_runtime():
    """libmi355faiss.so is linked without a NEEDED entry for libamdhip64 (csrc/Makefile): the host picks the
    HIP runtime.  PyTorch wheels bundle their own libamdhip64 + libhsa-runtime64, and two HSA runtimes cannot
    coexist in one process, so when torch is installed its copy is the one to share."""
    import importlib.util

    cands = []
    spec = importlib.util.find_spec("torch")
    if spec and spec.submodule_search_locations:
        cands.append(os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so"))
    cands += ["/opt/rocm/lib/libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so"]
    errs = []
    for c in cands:
        if os.path.isabs(c) and not os.path.exists(c):
            continue
        try:
            return C.CDLL(c, mode=C.RTLD_GLOBAL)
        except OSError as e:  # pragma: no cover
            errs.append(f"{c}: {e}")
    raise ImportError("no HIP runtime (libamdhip64) could be loaded: " + "; ".join(errs))


_HIP = _preload_hip_runtime()
_L = C.CDLL(LIB_PATH)
_p, _i64 = C.c_void_p, C.c_int64
_L.mvs_last_error.restype = C.c_char_p
_L.mvs_version.restype = C.c_char_p
_L.mvs_index_factory.argtypes = [C.POINTER(_p), C.c_int, C.c_char_p, C.c_int]
_L.mvs_index_free.argtypes = [_p]
_L.mvs_index_d.argtypes = [_p]
_L.mvs_index_ntotal.argtypes = [_p]
_L.mvs_index_ntotal.restype = _i64
_L.mvs_index_is_trained.argtypes = [_p]
_L.mvs_index_metric_type.argtypes = [_p]
_L.mvs_index_kind.argtypes = [_p]
_L.mvs_index_device.argtypes = [_p]
_L.mvs_index_idmap_sub.argtypes = [_p]
_L.mvs_index_idmap_sub.restype = _p
_L.mvs_index_ivf_quantizer.argtypes = [_p]
_L.mvs_index_ivf_quantizer.restype = _p
_L.mvs_index_hnsw_set_ef_construction.argtypes = [_p, C.c_int]
_L.mvs_index_hnsw_get_ef_construction.argtypes = [_p]
_L.mvs_index_hnsw_graph_info.argtypes = [_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
_L.mvs_index_hnsw_graph_info.restype = _i64
_L.mvs_index_hnsw_walk_stats.argtypes = [_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
_L.mvs_index_hnsw_get_graph.argtypes = [_p, _p, _p, _p]
_L.mvs_index_ivf_nlist.argtypes = [_p]
_L.mvs_index_ivf_nlist.restype = _i64
_L.mvs_index_ivf_get_centroids.argtypes = [_p, _p]
_L.mvs_index_ivf_set_centroids.argtypes = [_p, _p]
_L.mvs_index_train.argtypes = [_p, _i64, _p]
_L.mvs_index_add.argtypes = [_p, _i64, _p]
_L.mvs_index_add_with_ids.argtypes = [_p, _i64, _p, _p]
_L.mvs_index_search.argtypes = [_p, _i64, _p, _i64, _p, _p, C.POINTER(SearchParams)]
_L.mvs_index_to_gpu.argtypes = [_p, C.c_int]
_L.mvs_index_clone_to_gpu.argtypes = [C.POINTER(_p), _p, C.c_int]
_L.mvs_index_prefilter_stats.argtypes = [_p, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(C.c_float), C.POINTER(C.c_float)]
_L.mvs_index_collect_stats.argtypes = [_p, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]
_L.mvs_index_ivf_probe_stats.argtypes = [_p, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]
_L.mvs_index_shadow_stats.argtypes = [_p, C.POINTER(_i64), C.POINTER(C.c_double)]
_L.mvs_index_get_stat.argtypes = [_p, C.c_char_p, C.POINTER(_i64)]
_L.mvs_trace_push.argtypes = [C.c_char_p]
_L.mvs_trace_pop.argtypes = []
_L.mvs_index_shard_to_gpus.argtypes = [_p, C.POINTER(C.c_int), C.c_int]
_L.mvs_index_shard_info.argtypes = [_p, C.POINTER(C.c_int), C.c_int, C.POINTER(_i64), C.POINTER(_i64)]
_L.mvs_write_index.argtypes = [_p, C.c_char_p]
_L.mvs_read_index.argtypes = [C.POINTER(_p), C.c_char_p]
_L.mvs_index_add_device.argtypes = [_p, _i64, _p, _p, _p]
_L.mvs_index_search_device.argtypes = [_p, _i64, _p, _i64, _p, _p, C.POINTER(SearchParams), _p]
_L.mvs_index_set_label_offset.argtypes = [_p, _i64]
_L.mvs_merge_shards.argtypes = [C.c_int, _i64, _i64, C.c_int, _p, _p, _p, _p]
_L.mvs_merge_shards_raw.argtypes = [C.c_int, _i64, _i64, C.c_int, _p, _p, _p, _p]
_L.mvs_merge_records_device.argtypes = [C.c_int, _i64, C.c_int, C.c_int, C.c_int, _p, C.c_int, _p, _p, _p]
_L.mvs_finish_ip_ties.argtypes = [_i64, _i64, _i64, _p, _p, _i64, _p, _p, _p, _p]
_L.mvs_index_tie_candidates_device.argtypes = [_p, _i64, _p, _p, _i64, _p, C.POINTER(SearchParams), _p]
_L.mvs_index_ivf_tie_emit_device.argtypes = [_p, _i64, _p, _p, _p, _i64, _p, _p, _p, C.POINTER(SearchParams), _p]
_L.mvs_synth_uniform_device.argtypes = [_p, _i64, C.c_int, C.c_uint64, _i64, _p]
_L.mvs_debug_mfma_bf16_16x16x32.argtypes = [_p, _p, _p, _p, _i64]
_L.mvs_synth_clustered_device.argtypes = [_p, _i64, C.c_int, C.c_uint64, _i64, C.c_int, C.c_float, _p]
_L.mvs_index_last_kernel_info.argtypes = [_p, C.POINTER(KernelInfo)]
_L.mvs_index_set_kernel_timing.argtypes = [_p, C.c_int]
_L.mvs_index_kernel_time_stats.argtypes = [_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]
_L.mvs_index_set_option.argtypes = [_p, C.c_char_p, _i64]

# every symbol include/mi355_faiss.h declares (tests check the library exports all of them)
DECLARED_SYMBOLS = [
    "mvs_last_error", "mvs_index_factory", "mvs_index_free", "mvs_index_d", "mvs_index_ntotal",
    "mvs_index_is_trained", "mvs_index_metric_type", "mvs_index_kind", "mvs_index_idmap_sub",
    "mvs_index_ivf_quantizer", "mvs_index_ivf_nlist", "mvs_index_ivf_get_centroids", "mvs_index_ivf_set_centroids",
    "mvs_index_hnsw_set_ef_construction", "mvs_index_hnsw_get_ef_construction", "mvs_index_hnsw_graph_info", "mvs_index_hnsw_walk_stats", "mvs_index_hnsw_get_graph",
    "mvs_index_train", "mvs_index_add",
    "mvs_index_add_with_ids", "mvs_index_search", "mvs_index_to_gpu", "mvs_index_device", "mvs_index_clone_to_gpu",
    "mvs_index_prefilter_stats", "mvs_index_collect_stats", "mvs_index_ivf_probe_stats", "mvs_index_shadow_stats", "mvs_index_get_stat", "mvs_trace_push", "mvs_trace_pop", "mvs_index_shard_to_gpus", "mvs_index_shard_info", "mvs_write_index",
    "mvs_read_index", "mvs_index_add_device", "mvs_index_search_device", "mvs_index_set_label_offset",
    "mvs_merge_shards", "mvs_merge_shards_raw", "mvs_merge_records_device", "mvs_finish_ip_ties", "mvs_index_tie_candidates_device", "mvs_index_ivf_tie_emit_device", "mvs_synth_uniform_device", "mvs_synth_clustered_device", "mvs_debug_mfma_bf16_16x16x32", "mvs_index_last_kernel_info",
    "mvs_index_set_kernel_timing", "mvs_index_kernel_time_stats", "mvs_index_set_option", "mvs_device_count",
    "mvs_version",
]  # fmt: skip


def lib():
    return _L


class trace_range:
    """with trace_range("exchange"): ... -- a roctx range (include/mi355_faiss.h mvs_trace_push / mvs_trace_pop)"""

    def __init__(self, name):
        self.name = name.encode()

    def __enter__(self):
        _L.mvs_trace_push(self.name)
        return self

    def __exit__(self, *exc):
        _L.mvs_trace_pop()
        return False


def _check(rc):
    if rc:
        raise FaissException(_L.mvs_last_error().decode())


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64a(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def make_params(nprobe=0, efSearch=0, sel=None):
    """sel: None | ("bitmap", uint8 array) | ("batch", int64 array) -- IDSelectorBitmap / IDSelectorBatch"""
    p = SearchParams()
    p.nprobe, p.efSearch = nprobe, efSearch
    keep = None
    if sel is not None:
        kind, data = sel
        if kind == "bitmap":
            keep = np.ascontiguousarray(data, dtype=np.uint8)
            p.sel_kind = SEL_BITMAP
        elif kind == "batch":
            keep = _i64a(data)
            p.sel_kind = SEL_BATCH
        else:
            raise ValueError(kind)
        p.sel_n = keep.size
        p.sel_data = keep.ctypes.data
    return p, keep


def device_count():
    return _L.mvs_device_count()


class Index:
    """Device-native index; same method names as faiss.Index."""

    def __init__(self, handle, owned=True, parent=None):
        self._h = handle
        self._owned = owned
        self._parent = parent  # keeps the owning index alive for borrowed handles

    def __del__(self, _free=_L.mvs_index_free, _finalizing=sys.is_finalizing):
        # (both bound at definition: module globals are None during interpreter shutdown.  Indexes still alive when the interpreter
        # finalizes are NOT freed: the HIP runtime / RCCL may already be gone -- freeing a sharded index then aborted the process with
        # "double free or corruption" after the test summary; the OS reclaims the device memory with the process)
        if getattr(self, "_h", None) and getattr(self, "_owned", False) and not _finalizing():
            _free(self._h)
        self._h = None

    d = property(lambda s: _L.mvs_index_d(s._h))
    ntotal = property(lambda s: _L.mvs_index_ntotal(s._h))
    is_trained = property(lambda s: bool(_L.mvs_index_is_trained(s._h)))
    metric_type = property(lambda s: _L.mvs_index_metric_type(s._h))
    kind = property(lambda s: _L.mvs_index_kind(s._h))
    device = property(lambda s: _L.mvs_index_device(s._h))

    @property
    def index(self):
        """IndexIDMap::index (src/faiss_extension.cpp:129)"""
        h = _L.mvs_index_idmap_sub(self._h)
        return Index(h, owned=False, parent=self) if h else None

    @property
    def quantizer(self):
        """IndexIVF::quantizer (src/faiss_extension.cpp:680)"""
        h = _L.mvs_index_ivf_quantizer(self._h)
        return Index(h, owned=False, parent=self) if h else None

    @property
    def nlist(self):
        return _L.mvs_index_ivf_nlist(self._h)

    def ivf_centroids(self):
        out = np.empty((self.nlist, self.d), dtype=np.float32)
        _check(_L.mvs_index_ivf_get_centroids(self._h, _ptr(out)))
        return out

    def ivf_set_centroids(self, c):
        c = _f32(c).reshape(self.nlist, self.d)
        _check(_L.mvs_index_ivf_set_centroids(self._h, _ptr(c)))

    def set_ef_construction(self, v):
        _check(_L.mvs_index_hnsw_set_ef_construction(self._h, int(v)))

    def hnsw_walk_stats(self):
        """counters of the last search run with kernel timing on: distance evaluations, f32 rows fetched, bf16 rows looked at"""
        ev, f32, bf = C.c_double(), C.c_double(), C.c_double()
        _check(_L.mvs_index_hnsw_walk_stats(self._h, C.byref(ev), C.byref(f32), C.byref(bf)))
        return {"evaluations": ev.value, "f32_rows": f32.value, "bf16_rows": bf.value}

    def hnsw_graph(self):
        """-> dict(levels[n], offsets[n+1], neighbors[...], max_level, entry_point) (FAISS's HNSW arrays)"""
        ml, ep = C.c_int(0), C.c_int(0)
        nb = _L.mvs_index_hnsw_graph_info(self._h, C.byref(ml), C.byref(ep))
        if nb < 0:
            raise FaissException("not an HNSW index")
        n = self.ntotal
        levels = np.empty(n, dtype=np.int32)
        offsets = np.empty(n + 1, dtype=np.int64)
        neighbors = np.empty(max(nb, 1), dtype=np.int32)
        _check(_L.mvs_index_hnsw_get_graph(self._h, _ptr(levels), _ptr(offsets), _ptr(neighbors)))
        return dict(levels=levels, offsets=offsets, neighbors=neighbors[:nb], max_level=ml.value, entry_point=ep.value)

    def train(self, x):
        x = _f32(x).reshape(-1, self.d)
        _check(_L.mvs_index_train(self._h, x.shape[0], _ptr(x)))

    def add(self, x):
        x = _f32(x).reshape(-1, self.d)
        _check(_L.mvs_index_add(self._h, x.shape[0], _ptr(x)))

    def add_with_ids(self, x, ids):
        x = _f32(x).reshape(-1, self.d)
        ids = _i64a(ids)
        assert ids.size == x.shape[0]
        _check(_L.mvs_index_add_with_ids(self._h, x.shape[0], _ptr(x), _ptr(ids)))

    def search(self, x, k, nprobe=0, efSearch=0, sel=None):
        x = _f32(x).reshape(-1, self.d)
        nq = x.shape[0]
        D = np.empty((nq, max(k, 0)), dtype=np.float32)
        I = np.empty((nq, max(k, 0)), dtype=np.int64)
        p, keep = make_params(nprobe, efSearch, sel)
        _check(_L.mvs_index_search(self._h, nq, _ptr(x), k, _ptr(D), _ptr(I), C.byref(p)))
        del keep
        return D, I

    def to_gpu(self, device):
        """faiss_to_gpu(name, device): src/gpu/gpu.cpp:48 (in place)"""
        _check(_L.mvs_index_to_gpu(self._h, int(device)))

    def clone_to_gpu(self, device):
        """faiss.index_cpu_to_gpu(res, device, index): returns a new index on `device`"""
        h = _p()
        _check(_L.mvs_index_clone_to_gpu(C.byref(h), self._h, int(device)))
        return Index(h)

    def prefilter_stats(self):
        q, f, e, b = _i64(0), _i64(0), C.c_float(0), C.c_float(0)
        _check(_L.mvs_index_prefilter_stats(self._h, C.byref(q), C.byref(f), C.byref(e), C.byref(b)))
        return {"queries": q.value, "fallback_queries": f.value, "max_rel_err": e.value, "err_bound": b.value}

    def ivf_probe_stats(self):
        """(query, list) pairs of the last IVF coarse-filter search and how many of them were scanned (probe pruning)."""
        a, b, c, e = _i64(0), _i64(0), _i64(0), _i64(0)
        _check(_L.mvs_index_ivf_probe_stats(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(e)))
        return {"pairs": a.value, "scanned": b.value, "forced_drains": c.value, "admitted": e.value}

    def get_stat(self, name):
        """a named diagnostic counter -- include/mi355_faiss.h mvs_index_get_stat"""
        v = _i64(0)
        _check(_L.mvs_index_get_stat(self._h, name.encode(), C.byref(v)))
        return v.value

    def shadow_stats(self):
        """Flat L2: the shadow clustering's state, size and cost -- include/mi355_faiss.h mvs_index_shadow_stats"""
        st, sec = (_i64 * 8)(), C.c_double(0)
        _check(_L.mvs_index_shadow_stats(self._h, st, C.byref(sec)))
        keys = ("state", "rows", "queries", "unproven", "builds", "extends", "device_bytes", "nlist")
        out = {k: int(v) for k, v in zip(keys, st)}
        out["build_seconds"] = sec.value
        return out

    def collect_stats(self):
        q, c, o = _i64(0), _i64(0), _i64(0)
        _check(_L.mvs_index_collect_stats(self._h, C.byref(q), C.byref(c), C.byref(o)))
        return {"queries": q.value, "candidates": c.value, "overflows": o.value}

    def shard_to_gpus(self, devices):
        """spread this index over `devices` in place (row shards; HNSW: replicas) -- include/mi355_faiss.h"""
        arr = (C.c_int * len(devices))(*[int(v) for v in devices])
        _check(_L.mvs_index_shard_to_gpus(self._h, arr, len(devices)))

    def shard_info(self):
        """-> None (not sharded) | dict(devices, rows_per_shard, last_tie_queries)"""
        devs, rows, ties = (C.c_int * 64)(), (_i64 * 64)(), _i64(0)
        n = _L.mvs_index_shard_info(self._h, devs, 64, rows, C.byref(ties))
        if n <= 0:
            return None
        return {"devices": list(devs[:n]), "rows_per_shard": list(rows[:n]), "last_tie_queries": ties.value}

    # ---- device-resident variants (torch tensors on the index's device) ----
    def add_torch(self, x, ids=None, stream=None):
        import torch

        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        if stream is None:
            stream = torch.cuda.current_stream(x.device).cuda_stream
        _check(
            _L.mvs_index_add_device(
                self._h, x.shape[0], x.data_ptr(), ids.data_ptr() if ids is not None else None, stream
            )
        )

    def search_torch(self, x, k, D=None, I=None, nprobe=0, efSearch=0, sel=None, stream=None):
        import torch

        assert x.is_cuda and x.is_contiguous()
        nq = x.shape[0]
        if D is None:
            D = torch.empty((nq, k), dtype=torch.float32, device=x.device)
        if I is None:
            I = torch.empty((nq, k), dtype=torch.int64, device=x.device)
        p, keep = make_params(nprobe, efSearch, sel)
        if stream is None:
            stream = torch.cuda.current_stream(x.device).cuda_stream
        _check(_L.mvs_index_search_device(self._h, nq, x.data_ptr(), k, D.data_ptr(), I.data_ptr(), C.byref(p), stream))
        del keep
        return D, I

    def ivf_tie_emit_torch(self, flagged, xq, T, k, sel=None, stream=None):
        """IVF row shard: for the flagged queries (int64 tensor of query numbers of the batch `xq` that has JUST been searched on this
        index) this shard's first k rows not worse than T in arrival order: (value f32, stored id i64, probe rank i32), each [nf, k],
        -1 padded -- include/mi355_faiss.h mvs_index_ivf_tie_emit_device"""
        import torch

        nf = int(flagged.shape[0])
        dev = xq.device
        v = torch.zeros((nf, k), dtype=torch.float32, device=dev)
        ids = torch.full((nf, k), -1, dtype=torch.int64, device=dev)
        rk = torch.full((nf, k), -1, dtype=torch.int32, device=dev)
        if nf == 0:
            return v, ids, rk
        flag = torch.empty(nf + 1, dtype=torch.int32, device=dev)
        flag[0] = nf
        flag[1:] = flagged.to(torch.int32)
        p, keep = make_params(0, 0, sel)
        if stream is None:
            stream = torch.cuda.current_stream(dev).cuda_stream
        _check(_L.mvs_index_ivf_tie_emit_device(self._h, nf, C.c_void_p(flag.data_ptr()), C.c_void_p(xq.data_ptr()), C.c_void_p(T.data_ptr()),
                                                int(k), C.c_void_p(v.data_ptr()), C.c_void_p(ids.data_ptr()), C.c_void_p(rk.data_ptr()),
                                                C.byref(p), C.c_void_p(stream)))
        del keep
        return v, ids, rk

    def tie_candidates_torch(self, xf, T, k, sel=None, stream=None):
        """per flagged query the k smallest GLOBAL rows with score >= T (ascending, -1 padded) -- include/mi355_faiss.h"""
        import torch

        nf = xf.shape[0]
        out = torch.empty((nf, k), dtype=torch.int64, device=xf.device)
        p, keep = make_params(0, 0, sel)
        if stream is None:
            stream = torch.cuda.current_stream(xf.device).cuda_stream
        _check(_L.mvs_index_tie_candidates_device(self._h, nf, xf.data_ptr(), T.data_ptr(), k, out.data_ptr(), C.byref(p), stream))
        del keep
        return out

    def set_label_offset(self, off):
        _check(_L.mvs_index_set_label_offset(self._h, int(off)))

    def set_option(self, key, value):
        _check(_L.mvs_index_set_option(self._h, key.encode(), int(value)))

    def set_kernel_timing(self, on):
        _check(_L.mvs_index_set_kernel_timing(self._h, 1 if on else 0))

    def kernel_time_stats(self):
        n, t = C.c_int(0), C.c_double(0)
        _check(_L.mvs_index_kernel_time_stats(self._h, C.byref(n), C.byref(t)))
        return n.value, t.value

    def last_kernel_info(self):
        ki = KernelInfo()
        _check(_L.mvs_index_last_kernel_info(self._h, C.byref(ki)))
        return {
            "name": ki.name.decode(), "flops": ki.flops, "bytes": ki.bytes, "last_ms": ki.last_ms,
            "grid": ki.grid, "block": ki.block, "lds_bytes": ki.lds_bytes, "nsplit": ki.nsplit,
        }  # fmt: skip


def index_factory(d, description, metric=METRIC_INNER_PRODUCT):
    """faiss.index_factory; default metric INNER_PRODUCT as the extension (src/faiss_extension.cpp:105)"""
    h = _p()
    _check(_L.mvs_index_factory(C.byref(h), int(d), description.encode(), int(metric)))
    return Index(h)


def write_index(index, filename):
    _check(_L.mvs_write_index(index._h, filename.encode()))


def read_index(filename):
    h = _p()
    _check(_L.mvs_read_index(C.byref(h), filename.encode()))
    return Index(h)


def merge_shards(metric, D, I):
    """Host k-way merge after the all-gather: D, I [nshard, nq, k] (global labels) -> [nq, k]"""
    D, I = _f32(D), _i64a(I)
    ns, nq, k = D.shape
    Do = np.empty((nq, k), dtype=np.float32)
    Io = np.empty((nq, k), dtype=np.int64)
    _check(_L.mvs_merge_shards(metric, nq, k, ns, _ptr(D), _ptr(I), _ptr(Do), _ptr(Io)))
    return Do, Io


def merge_records_torch(metric, rec, kout, raw=False):
    """device merge of gathered records: rec [nshard, nq, kk, 2] int64 (torch, on the GPU) -> (D [nq, kout] f32, I [nq, kout]
    i64) torch tensors on the same device; raw: pure order instead of FAISS's print order"""
    import torch

    ns, nq, kk, _ = rec.shape
    D = torch.empty((nq, kout), dtype=torch.float32, device=rec.device)
    I = torch.empty((nq, kout), dtype=torch.int64, device=rec.device)
    st = torch.cuda.current_stream(rec.device).cuda_stream
    _check(_L.mvs_merge_records_device(int(metric), nq, kk, int(kout), ns, C.c_void_p(rec.data_ptr()), 1 if raw else 0,
                                       C.c_void_p(D.data_ptr()), C.c_void_p(I.data_ptr()), C.c_void_p(st)))
    return D, I


def merge_shards_raw(metric, D, I):
    """[nshard, nq, kk] blocks -> merged top-kk per query in the PURE order (score desc / dist asc, id asc)"""
    D, I = _f32(D), _i64a(I)
    ns, nq, kk = D.shape
    Do = np.empty((nq, kk), dtype=np.float32)
    Io = np.empty((nq, kk), dtype=np.int64)
    _check(_L.mvs_merge_shards_raw(metric, nq, kk, ns, _ptr(D), _ptr(I), _ptr(Do), _ptr(Io)))
    return Do, Io


def finish_ip_ties(k, rawD, rawI, flagged, first_rows):
    """FAISS print order of the first k of every raw list + the CMin-heap outcome for the flagged queries"""
    rawD, rawI = _f32(rawD), _i64a(rawI)
    nq, kk = rawD.shape
    flagged, first_rows = _i64a(flagged), _i64a(first_rows)
    Do = np.empty((nq, k), dtype=np.float32)
    Io = np.empty((nq, k), dtype=np.int64)
    _check(_L.mvs_finish_ip_ties(nq, k, kk, _ptr(rawD), _ptr(rawI), flagged.size, _ptr(flagged), _ptr(first_rows), _ptr(Do), _ptr(Io)))
    return Do, Io


def synth_uniform_torch(n, d, seed, row0=0, device="cuda:0", out=None):
    import torch

    if out is None:
        out = torch.empty((n, d), dtype=torch.float32, device=device)
    st = torch.cuda.current_stream(out.device).cuda_stream
    _check(_L.mvs_synth_uniform_device(out.data_ptr(), n, d, seed, row0, st))
    return out


def mfma_bf16_16x16x32(A_bits, Bt_bits, C):
    """D = A B + C by ONE v_mfma_f32_16x16x32_bf16 per tile (diagnostics): A_bits [n][16][32], Bt_bits [n][16][32] uint16 bf16
    patterns (Bt = the columns of B as rows), C [n][16][16] float32"""
    import numpy as np

    A_bits = np.ascontiguousarray(A_bits, dtype=np.uint16)
    Bt_bits = np.ascontiguousarray(Bt_bits, dtype=np.uint16)
    Cc = np.ascontiguousarray(C, dtype=np.float32)
    n = A_bits.shape[0]
    assert A_bits.shape == (n, 16, 32) and Bt_bits.shape == (n, 16, 32) and Cc.shape == (n, 16, 16)
    D = np.empty((n, 16, 16), dtype=np.float32)
    _check(_L.mvs_debug_mfma_bf16_16x16x32(A_bits.ctypes.data, Bt_bits.ctypes.data, Cc.ctypes.data, D.ctypes.data, n))
    return D


def synth_clustered_torch(n, d, seed, row0=0, n_centers=1024, sigma=0.1, device="cuda:0", out=None):
    import torch

    if out is None:
        out = torch.empty((n, d), dtype=torch.float32, device=device)
    st = torch.cuda.current_stream(out.device).cuda_stream
    _check(_L.mvs_synth_clustered_device(out.data_ptr(), n, d, seed, row0, n_centers, sigma, st))
    return out
