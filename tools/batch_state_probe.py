#!/usr/bin/env python3
"""Why does a DIFFERENT batch take 12 % longer than the benchmark's?  Headline index; per search: ms, dominant
kernel ms (HIP events), rows admitted per query.  Sequences: the benchmark batch, the midpoint batch three times in a row (a first-use
effect would fade), a fresh uniform batch of another seed, a clone of the benchmark batch in new memory, midpoints again."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "duckdb-faiss-ext_amd", "pyhost"))
import torch
import mi355_faiss as mf
n, d, nq, k = int(os.environ.get("N", 10_000_000)), 128, 10_000, 10
ix = mf.index_factory(d, "Flat", mf.METRIC_L2)
for s0 in range(0, n, 1 << 20):
    ix.add_torch(mf.synth_uniform_torch(min(1 << 20, n - s0), d, 1234, row0=s0))
xq = mf.synth_uniform_torch(nq, d, 4321, row0=0)
mid = (0.5 * (xq + xq.roll(1, 0))).contiguous()
other = mf.synth_uniform_torch(nq, d, 999, row0=0)
clone = xq.clone()
# midpoints have a smaller spread around the data's centre: the same spread with the benchmark's marginals = shrink towards 0.5
shrunk = (0.5 + (xq - 0.5) * 0.7071).contiguous()
D = torch.empty((nq, k), dtype=torch.float32, device="cuda:0"); I = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
ix.set_kernel_timing(True)
for _ in range(3):
    ix.search_torch(xq, k, D=D, I=I)
torch.cuda.synchronize()
print(f"{'batch':<28} {'ms':>7} {'kernel ms':>9} {'admitted/q':>10}")
for name, x in [("benchmark", xq), ("midpoints", mid), ("midpoints", mid), ("midpoints", mid), ("benchmark", xq), ("uniform seed 999", other),
                ("uniform seed 999", other), ("benchmark, new memory", clone), ("shrunk towards the centre", shrunk), ("shrunk towards the centre", shrunk),
                ("midpoints", mid), ("benchmark", xq), ("benchmark after 0.5 s idle", xq), ("benchmark", xq), ("benchmark after 0.05 s idle", xq),
                ("midpoints after 0.5 s idle", mid), ("midpoints", mid)]:
    torch.cuda.synchronize()
    if "idle" in name:  # (the bench's extra searches come after host-side work: is it the DEVICE that is cold?)
        time.sleep(0.5 if "0.5" in name else 0.05)
    t0 = time.perf_counter()
    ix.search_torch(x, k, D=D, I=I); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    try:
        adm = ix.ivf_probe_stats()["admitted"] / nq
    except Exception:
        adm = float("nan")
    print(f"{name:<30} {ms:>7.2f} {ix.last_kernel_info()['last_ms']:>9.3f} {adm:>10.1f}", flush=True)
