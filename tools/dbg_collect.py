import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "duckdb-faiss-ext_amd", "pyhost"))
import numpy as np
import mi355_faiss as mf
rs = np.random.RandomState(5)
for metric in (0, 1):
    for n, nq, shift in ((120_000, 400, 0.0), (120_000, 400, 0.5), (1_000_000, 1000, 0.0)):
        xb = rs.rand(n, 128).astype(np.float32) - shift
        xq = rs.rand(nq, 128).astype(np.float32) - shift
        ix = mf.index_factory(128, "Flat", metric)
        ix.set_option("prefilter", 2)
        ix.add(xb)
        D, I = ix.search(xq, 10)
        print("metric", metric, "n", n, "nq", nq, "shift", shift, ix.last_kernel_info()["name"], ix.collect_stats(), flush=True)
