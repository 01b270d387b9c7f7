import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "duckdb-faiss-ext_amd", "pyhost"))
import numpy as np, torch
import mi355_faiss as mf
d = 768
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ix = mf.index_factory(d, "IDMap,HNSW32", mf.METRIC_L2)
x = np.random.RandomState(1).rand(300_000, d).astype(np.float32)
ids = np.arange(len(x), dtype=np.int64)
t_last = time.perf_counter(); n_last = 0
for i0 in range(0, len(x), chunk):
    ix.add_with_ids(x[i0:i0 + chunk], ids[i0:i0 + chunk])
    if (i0 // chunk) % max(1, (20000 // chunk)) == 0 and i0 > 0:
        torch.cuda.synchronize()
        t = time.perf_counter()
        print("ntotal %7d: %.1f ms per %d-row add (%.0f rows/s)" % (i0, (t - t_last) / ((i0 - n_last) / chunk) * 1e3, chunk, (i0 - n_last) / (t - t_last)), flush=True)
        t_last, n_last = t, i0
