#!/usr/bin/env python3
"""C3 shape with inner product: which kernel serves it, how long a batch takes, candidates per query (option sweeps)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "duckdb-faiss-ext_amd", "pyhost"))
import numpy as np, torch
import mi355_faiss as mf
n, d, nq, k = int(os.environ.get("N", 10_000_000)), 128, 10_000, 10
metric = mf.METRIC_INNER_PRODUCT if os.environ.get("METRIC", "IP") == "IP" else mf.METRIC_L2
ix = mf.index_factory(d, "IVF4096,Flat", metric)
xb = mf.synth_clustered_torch(n, d, 1234, n_centers=1024, sigma=0.1)
ix.train(xb[: 256 * 4096 * 2].cpu().numpy())
for s0 in range(0, n, 1 << 20):
    ix.add_torch(xb[s0 : s0 + (1 << 20)])
torch.cuda.synchronize()
xq = mf.synth_clustered_torch(nq, d, 4321, n_centers=1024, sigma=0.1)
D = torch.empty((nq, k), dtype=torch.float32, device="cuda:0"); I = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
for opts in ({}, {"ivf_exact_ties": 0}, {"ivf_collect": 0}):
    for key, v in opts.items():
        ix.set_option(key, v)
    ix.search_torch(xq, k, D=D, I=I, nprobe=32); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ix.search_torch(xq, k, D=D, I=I, nprobe=32)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(opts, ix.last_kernel_info()["name"], f"{ms:.2f} ms per batch", flush=True)
    for key in opts:
        ix.set_option(key, {"ivf_exact_ties": 1, "ivf_collect": -1}[key])
