import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "duckdb-faiss-ext_amd", "pyhost"))
import numpy as np, mi355_faiss as mf
from oracle import oracle as orc
rs = np.random.RandomState(0)
def cmp(name, g, o, xq, k, **kw):
    D, I = g.search(xq, k, **kw); Do, Io = o.search(xq, k, **kw)
    print(name, "labels", np.array_equal(I, Io), "dist", np.array_equal(D.view(np.uint32), Do.view(np.uint32)), flush=True)
for d in (1030, 2500, 4096):
    xb = rs.rand(1500, d).astype(np.float32) - 0.5; xq = rs.rand(40, d).astype(np.float32) - 0.5
    for metric in (1, 0):
        g = mf.index_factory(d, "Flat", metric); o = orc.Index(d, "Flat", metric); g.add(xb); o.add(xb)
        cmp("flat d=%d m=%d nq=40" % (d, metric), g, o, xq, 7)
        cmp("flat d=%d m=%d nq=3" % (d, metric), g, o, xq[:3], 7)
    g = mf.index_factory(d, "HNSW8", 1); g.set_option("hnsw_build_waves", 1); o = orc.Index(d, "HNSW8", 1); g.add(xb[:400]); o.add(xb[:400])
    cmp("hnsw d=%d" % d, g, o, xq, 5, efSearch=32)
    g = mf.index_factory(d, "IVF4,Flat", 1); o = orc.Index(d, "IVF4,Flat", 1); o.train(xb); g.ivf_set_centroids(o.ivf_centroids()); g.add(xb); o.add(xb)
    cmp("ivf d=%d" % d, g, o, xq, 5, nprobe=2)
try:
    mf.index_factory(5000, "HNSW8", 1)
except mf.FaissException as e:
    print("hnsw d=5000:", str(e)[:80])
