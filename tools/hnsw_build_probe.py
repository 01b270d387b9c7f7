"""HNSW32 build on UNIFORM rows (the ingest line's hardest case, 10 k rows/s in round 5): builds of `rows` x `d` in the glue's 2048-row
chunks -- rows/s and recall@10 (efSearch 128, 1000 queries, ground truth = Flat) per setting of
  hnsw_build_wg     wavefronts that share one insertion's back links (round 6; 1 = round 5's one wavefront per point)
  hnsw_build_waves  concurrent insertions (0 = auto: min(1024, graph / 32))
usage: python tools/hnsw_build_probe.py [rows=100000] [d=768] [efc=40]   env: WGS="1 4 8", WAVES="0", KIND=uniform|clustered"""
import os
import sys
import time

import torch

import mi355_faiss as mf

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
efc = int(sys.argv[3]) if len(sys.argv) > 3 else 40
kind = os.environ.get("KIND", "uniform")
if kind == "uniform":
    x, xq = mf.synth_uniform_torch(n, d, 99, row0=0), mf.synth_uniform_torch(1000, d, 77, row0=0)
else:  # C5's rows
    x = mf.synth_clustered_torch(n, d, 1234, row0=0, n_centers=1024, sigma=1.0)
    xq = mf.synth_clustered_torch(1000, d, 4321, row0=0, n_centers=1024, sigma=1.0)
    x, xq = x / x.norm(dim=1, keepdim=True), (xq / xq.norm(dim=1, keepdim=True)).contiguous()
flat = mf.index_factory(d, "Flat", mf.METRIC_L2)
flat.add_torch(x)
_, gt = flat.search_torch(xq, 10)
gt = gt.cpu().numpy()
del flat
print(f"# HNSW32 efConstruction {efc}: {n} {kind} rows x {d} dims added 2048 at a time")
for waves in [int(v) for v in os.environ.get("WAVES", "0").split()]:
    for wg in [int(v) for v in os.environ.get("WGS", "1 4 8").split()]:
        ix = mf.index_factory(d, "HNSW32", mf.METRIC_L2)
        ix.set_ef_construction(efc)
        ix.set_option("hnsw_build_wg", wg)
        if waves:
            ix.set_option("hnsw_build_waves", waves)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i0 in range(0, n, 2048):  # the glue's DataChunks
            ix.add_torch(x[i0 : i0 + 2048])
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        _, I = ix.search_torch(xq, 10, efSearch=128)
        I = I.cpu().numpy()
        rec = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(I, gt)) / (10.0 * len(gt))
        print(f"hnsw_build_wg {wg} hnsw_build_waves {waves}: {t:.2f} s = {n / t:.0f} rows/s; recall@10 {rec:.4f}", flush=True)
        del ix
