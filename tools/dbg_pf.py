import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/duckdb-faiss-ext_amd/pyhost")
import numpy as np
import mi355_faiss as mf
from oracle import oracle as orc
rs = np.random.RandomState(128 + 200000)
d, nb, nq, k = 128, 200000, 700, 10
xb = rs.rand(nb, d).astype(np.float32); xq = rs.rand(nq, d).astype(np.float32)
pf, ex = mf.index_factory(d, "Flat", 1), mf.index_factory(d, "Flat", 1)
pf.set_option("prefilter", 1); ex.set_option("prefilter", 0)
pf.add(xb); ex.add(xb)
D1, I1 = pf.search(xq, k); D0, I0 = ex.search(xq, k)
Do, Io = orc.flat_search(1, xb, xq, k, force_path=orc.PATH_BLAS)
print("labels pf==ex", np.array_equal(I1, I0), "ex==orc", np.array_equal(I0, Io), np.array_equal(D0, Do))
bad = np.argwhere(D1.view(np.uint32) != D0.view(np.uint32))
print("mismatch slots", len(bad), "of", D1.size)
for q, j in bad[:10]:
    r = I1[q, j]
    chain = np.float32(0)
    ip = np.float32(0)
    for t in range(d):
        ip = np.float32(np.float64(xq[q, t]) * np.float64(xb[r, t]) + np.float64(ip))  # fma in f64 then round: exact fma emulation
    print(q, j, r, D1[q, j], D0[q, j], Do[q, j], "ulps", int(D1[q, j].view(np.uint32)) - int(D0[q, j].view(np.uint32)), "row bit4", (r >> 4) & 1, "q mod 64", q % 64)
print("stats", pf.prefilter_stats())
