#!/usr/bin/env python3
"""flat_bf16_big_kernel pipeline variants (option cl_big_mode 0..3): results against the exact f32 kernel on a small d = 768
index, then (only if equal) the kernel time on a 2M-row index.  One process per mode: a GPU fault must not take the others down."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "duckdb-faiss-ext_amd", "pyhost"))
import numpy as np, torch
import mi355_faiss as mf

mode, d = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 768
metric = mf.METRIC_INNER_PRODUCT
rs = np.random.RandomState(1)
ok = True
for n, nq in ((40_000, 300), (70_001, 1000)):
    xb = rs.rand(n, d).astype(np.float32) - 0.5
    xq = rs.rand(nq, d).astype(np.float32) - 0.5
    a, b = mf.index_factory(d, "Flat", metric), mf.index_factory(d, "Flat", metric)
    a.set_option("prefilter", 2); a.set_option("cl_big_mode", mode); b.set_option("prefilter", 0)
    a.add(xb); b.add(xb)
    D1, I1 = a.search(xq, 10); name = a.last_kernel_info()["name"]
    D0, I0 = b.search(xq, 10)
    same = bool(np.array_equal(I1, I0) and np.array_equal(D1.view(np.uint32), D0.view(np.uint32)))
    print(f"mode {mode} d={d} N={n} nq={nq}: {name} equals the exact kernel: {same} (rows differing: {(I1 != I0).any(axis=1).sum()})", flush=True)
    ok &= same
if not ok:
    sys.exit(3)
n, nq = 2_000_000, 10_000
ix = mf.index_factory(d, "Flat", metric)
ix.set_option("cl_big_mode", mode)
for s0 in range(0, n, 1 << 18):
    x = mf.synth_clustered_torch(min(1 << 18, n - s0), d, 1234, row0=s0, n_centers=1024, sigma=1.0); x /= x.norm(dim=1, keepdim=True)
    ix.add_torch(x); torch.cuda.synchronize()
xq = mf.synth_clustered_torch(nq, d, 4321, n_centers=1024, sigma=1.0); xq /= xq.norm(dim=1, keepdim=True)
D = torch.empty((nq, 10), dtype=torch.float32, device="cuda:0"); I = torch.empty((nq, 10), dtype=torch.int64, device="cuda:0")
ix.search_torch(xq, 10, D=D, I=I); torch.cuda.synchronize()
ix.set_kernel_timing(True)
for _ in range(3):
    ix.search_torch(xq, 10, D=D, I=I)
torch.cuda.synchronize()
nl, ms = ix.kernel_time_stats()
ki = ix.last_kernel_info()
print(f"mode {mode} d={d} N={n}: {ki['name']} {ms / nl:.2f} ms per launch, {2.0 * nq * n * d / (ms / nl * 1e-3) / 1e12:.0f} TFLOP/s = {2.0 * nq * n * d / (ms / nl * 1e-3) / 2.5e15:.3f} of the bf16 peak", flush=True)
