#!/usr/bin/env python3
"""What the in-library sharding (csrc/sharded.hip) costs on ONE device: the headline index as G virtual shards on device 0
(host-pointer API: per-shard H2D of the queries, raw shard searches on G streams, D2H, host k-way merge) next to the
unsharded index through the same host-pointer API.  Not a scaling number -- there is one GPU -- but it bounds the fixed
cost the exchange + merge add per batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "duckdb-faiss-ext_amd", "pyhost"))
import numpy as np, torch
import mi355_faiss as mf

n, d, nq, k = int(os.environ.get("N", 10_000_000)), 128, 10_000, 10
ix = mf.index_factory(d, "Flat", mf.METRIC_L2)
for s0 in range(0, n, 1 << 20):
    ix.add_torch(mf.synth_uniform_torch(min(1 << 20, n - s0), d, 1234, row0=s0)); torch.cuda.synchronize()
xq = mf.synth_uniform_torch(nq, d, 4321).cpu().numpy()
def timeit(ix, reps=4):
    ix.search(xq, k)
    t0 = time.perf_counter()
    for _ in range(reps):
        D, I = ix.search(xq, k)
    return (time.perf_counter() - t0) / reps * 1e3, D, I
t1, D1, I1 = timeit(ix)
print(f"unsharded, host-pointer API: {t1:.2f} ms per 10k batch ({nq / t1 * 1e3:.0f} QPS)")
for G in (2, 4, 8):
    sh = ix.clone_to_gpu(0)
    sh.shard_to_gpus([0] * G)
    tg, Dg, Ig = timeit(sh)
    same = np.array_equal(I1, Ig) and np.array_equal(D1.view(np.uint32), Dg.view(np.uint32))
    print(f"{G} virtual shards on device 0: {tg:.2f} ms per batch ({nq / tg * 1e3:.0f} QPS), results identical: {same}, rows/shard {sh.shard_info()['rows_per_shard'][:2]}...")
    del sh
