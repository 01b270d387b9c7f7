"""debug: inner product, 40 distinct vectors x 2500 copies (tests/test_collect_gpu.py::test_duplicates_and_ties_need_no_fall_back[IP])"""
import numpy as np

import mi355_faiss as mf
from oracle import oracle as orc

IP = orc.METRIC_INNER_PRODUCT
rs = np.random.RandomState(5)
base = rs.rand(40, 128).astype(np.float32)
xb = base[rs.randint(0, 40, 100_000)]
xq = rs.rand(300, 128).astype(np.float32)
Do, Io = orc.flat_search(IP, xb, xq[:32], 10, force_path=orc.PATH_BLAS)
for name, opts in (("coarse", {"prefilter": 2}), ("coarse,no-bucket", {"prefilter": 2, "cl_fbucket": 0}), ("coarse,ties-rescan", {"prefilter": 2, "tie_from_candidates": 0}),
                   ("exact", {"prefilter": 0})):
    ix = mf.index_factory(128, "Flat", IP)
    for kk, v in opts.items():
        ix.set_option(kk, v)
    for i0 in range(0, len(xb), 1 << 16):
        ix.add(xb[i0 : i0 + (1 << 16)])
    for rep in range(2):
        D, I = ix.search(xq, 10)
        bad = np.nonzero((I[:32] != Io).any(axis=1))[0]
        print(name, "rep", rep, ix.last_kernel_info()["name"], "queries differing from the oracle (of 32):", len(bad), "collect", ix.collect_stats(), flush=True)
        for q in bad[:2]:
            print("   q", q, "got", I[q].tolist(), "want", Io[q].tolist(), "D", D[q][:3].tolist(), Do[q][:3].tolist())
