#!/usr/bin/env python3
"""IVF list lengths: ms per batch of IVF4096,Flat (C3's rows: clustered sigma 0.1, nprobe 32) by k -- which kernel serves which k.
env: N (10 000 000), NQ (2048), KS ("10 32 33 64 100 256 1000"), NLIST (4096), NPROBE (32)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "duckdb-faiss-ext_amd", "pyhost"))
import torch
import mi355_faiss as mf

d, n, nq = 128, int(os.environ.get("N", 10_000_000)), int(os.environ.get("NQ", 2048))
nlist, nprobe = int(os.environ.get("NLIST", 4096)), int(os.environ.get("NPROBE", 32))
ks = [int(v) for v in os.environ.get("KS", "10 32 33 64 100 256 1000").split()]
dev = "cuda:0"
rows = lambda m, seed, row0: mf.synth_clustered_torch(m, d, seed, row0=row0, n_centers=1024, sigma=0.1)
ix = mf.index_factory(d, f"IVF{nlist},Flat", mf.METRIC_L2)
ix.train(rows(min(n, 1 << 20), 1234, 0).cpu().numpy())
for s0 in range(0, n, 1 << 20):
    ix.add_torch(rows(min(1 << 20, n - s0), 1234, s0))
xq = rows(nq, 4321, 0).contiguous()
print(f"# IVF{nlist},Flat L2 d={d} N={n} nq={nq} nprobe={nprobe}, clustered sigma 0.1")
print(f"{'k':>5} {'ms':>9} kernel")
for k in ks:
    D = torch.empty((nq, k), dtype=torch.float32, device=dev); I = torch.empty((nq, k), dtype=torch.int64, device=dev)
    for _ in range(2):
        ix.search_torch(xq, k, D=D, I=I, nprobe=nprobe); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ix.search_torch(xq, k, D=D, I=I, nprobe=nprobe)
    torch.cuda.synchronize()
    print(f"{k:>5} {(time.perf_counter() - t0) / 3 * 1e3:>9.2f} {ix.last_kernel_info()['name']}", flush=True)
