#!/bin/bash
python3 - <<'PY'
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "duckdb-faiss-ext_amd/pyhost")
import mi355_faiss as mf
from oracle import oracle as orc
d, nb = 32, 30_000
rs = np.random.RandomState(11)
xb = rs.rand(nb, d).astype(np.float32)
xb[rs.randint(0, nb, 2000)] = xb[rs.randint(0, nb, 2000)]
xq = rs.rand(43, d).astype(np.float32)
for metric in (orc.METRIC_L2, orc.METRIC_INNER_PRODUCT):
    one, o = mf.index_factory(d, "Flat", metric), orc.Index(d, "Flat", metric)
    one.add(xb); o.add(xb)
    for k in (100, 500, 2048):
        D, I = one.search(xq, k); Do, Io = o.search(xq, k)
        print("metric", metric, "k", k, "kernel", one.last_kernel_info()["name"], "labels equal", np.array_equal(I, Io), "dist equal", np.array_equal(D.view(np.uint32), Do.view(np.uint32)))
        bad = np.argwhere(I != Io)
        for q, j in bad[:6]:
            a, b = I[q, j], Io[q, j]
            print("  q", q, "slot", j, "dev", a, D[q, j], "orc", b, Do[q, j], "rows equal", np.array_equal(xb[a], xb[b]),
                  "dev label in orc row:", a in Io[q], "orc label in dev row:", b in I[q])
PY
