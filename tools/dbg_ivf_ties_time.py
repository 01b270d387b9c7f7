"""debug: per-search wall time of tests/test_ivf_gpu.py::test_ivf_exact_ties_follow_the_heap's loop (which step is slow?)"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "duckdb-faiss-ext_amd", "pyhost")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mi355_faiss as mf
from oracle import oracle as orc
import test_ivf_gpu as T
L2 = orc.METRIC_L2
d, n, nlist = 32, 12000, 16
xb, xq = T._tied_data(n, d, 77 + d, L2)
rs = np.random.RandomState(5)
ids = (rs.permutation(4 * n)[:n] + 3).astype(np.int64)
t0 = time.perf_counter(); o = orc.Index(d, "IVF16,Flat", L2); o.train(xb); print("oracle train", time.perf_counter() - t0)
g = mf.index_factory(d, "IVF16,Flat", L2)
g.ivf_set_centroids(o.ivf_centroids())
for a in (g, o):
    t0 = time.perf_counter()
    for i0 in range(0, n, 5000):
        a.add_with_ids(xb[i0 : i0 + 5000], ids[i0 : i0 + 5000])
    print("add", type(a).__name__, time.perf_counter() - t0)
for opts, k, nprobe, nq in (({}, 10, 4, 257), ({}, 1, 3, 257), ({"ivf_collect": 0}, 10, 4, 257), ({}, 16, 4, 100), ({}, 40, 8, 100), ({}, 5, 6, 7), ({"ivf_select": 1}, 10, 4, 64), ({}, 300, 16, 33)):
    for key, v in opts.items():
        g.set_option(key, v)
    t0 = time.perf_counter(); D, I = g.search(xq[:nq], k, nprobe=nprobe); t1 = time.perf_counter()
    for key in opts:
        g.set_option(key, {"ivf_collect": -1, "ivf_mfma": -1, "ivf_select": 0}[key])
    Do, Io = o.search(xq[:nq], k, nprobe=nprobe); t2 = time.perf_counter()
    print(opts, k, nprobe, nq, "device %.3f s oracle %.3f s" % (t1 - t0, t2 - t1), g.last_kernel_info()["name"], flush=True)
