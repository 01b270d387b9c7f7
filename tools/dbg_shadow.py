import os, sys, faulthandler
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "duckdb-faiss-ext_amd", "pyhost"))
import numpy as np
import mi355_faiss as mf
from oracle import oracle as orc
kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
d, n, nq, k = 64, 300_000, 600, 10
if kind == "uniform":
    xb = orc.synth_uniform(n, d, 7); xq = orc.synth_uniform(nq, d, 8)
else:
    xb = orc.synth_clustered(n, d, 7, n_centers=256, sigma=0.05); xb = np.round(xb * 8) / 8
    xq = xb[np.random.RandomState(3).randint(0, n, nq)].copy()
ix = mf.index_factory(d, "Flat", mf.METRIC_L2)
ix.add(xb)
ix.set_option("flat_shadow", 1)
print("searching", flush=True)
for rep in range(3):
    D, I = ix.search(xq, k)
    print(rep, ix.last_kernel_info()["name"], flush=True)
ref = orc.flat_search(mf.METRIC_L2, xb, xq, k, force_path=orc.PATH_BLAS)
print("labels", np.array_equal(I, ref[1]), "dist", np.array_equal(D.view(np.uint32), ref[0].view(np.uint32)))
