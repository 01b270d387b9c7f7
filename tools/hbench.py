#!/usr/bin/env python3
"""HNSW micro-benchmark: device build time + search kernel time / recall on synthetic data.
    python tools/hbench.py --n 200000 --d 768 --M 32 --nq 10000 --efs 16,64,128 [--waves 1] [--normalize]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "duckdb-faiss-ext_amd", "pyhost"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200_000)
    ap.add_argument("--nq", type=int, default=10_000)
    ap.add_argument("--d", type=int, default=768)
    ap.add_argument("--M", type=int, default=32)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--efs", default="16,64,128")
    ap.add_argument("--efc", type=int, default=40)
    ap.add_argument("--metric", default="L2")
    ap.add_argument("--waves", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=65536)
    ap.add_argument("--data", default="clustered")
    ap.add_argument("--normalize", action="store_true")
    ap.add_argument("--sigma", type=float, default=0.1)
    ap.add_argument("--centers", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import numpy as np
    import torch

    import mi355_faiss as mf

    metric = mf.METRIC_L2 if args.metric == "L2" else mf.METRIC_INNER_PRODUCT
    if args.data == "uniform":
        gen = mf.synth_uniform_torch
    else:
        def gen(n, d, seed, row0=0):
            return mf.synth_clustered_torch(n, d, seed, row0=row0, n_centers=args.centers, sigma=args.sigma)

    def prep(x):
        if args.normalize:
            x /= x.norm(dim=1, keepdim=True)
        return x

    ix = mf.index_factory(args.d, "HNSW%d" % args.M, metric)
    ix.set_ef_construction(args.efc)
    if args.waves:
        ix.set_option("hnsw_build_waves", args.waves)
    fl = mf.index_factory(args.d, "Flat", metric)
    t0 = time.time()
    tb = 0.0
    for s0 in range(0, args.n, args.chunk):
        xb = prep(gen(min(args.chunk, args.n - s0), args.d, 1234, row0=s0))
        torch.cuda.synchronize()
        t1 = time.time()
        ix.add_torch(xb)
        torch.cuda.synchronize()
        tb += time.time() - t1
        fl.add_torch(xb)
        if (s0 // args.chunk) % 4 == 0:
            print("  built %d rows, %.1f s (%.0f rows/s)" % (s0 + xb.shape[0], tb, (s0 + xb.shape[0]) / tb), flush=True)
    print("build: n=%d d=%d M=%d efC=%d -> %.2f s (%.0f rows/s)" % (args.n, args.d, args.M, args.efc, tb, args.n / tb))
    g = ix.hnsw_graph()
    deg0 = np.mean([(g["neighbors"][g["offsets"][v] : g["offsets"][v] + 2 * args.M] >= 0).sum() for v in range(0, args.n, max(1, args.n // 2000))])
    print("graph: max_level=%d mean level-0 degree=%.1f" % (g["max_level"], deg0))
    xq = prep(gen(args.nq, args.d, 4321))
    ns = min(args.nq, 1000)
    _, Igt = fl.search_torch(xq[:ns].contiguous(), args.k)
    torch.cuda.synchronize()
    Igt = Igt.cpu().numpy()
    nl0, ms0 = 0, 0.0  # kernel_time_stats() is cumulative over the life of the index
    for ef in [int(e) for e in args.efs.split(",")]:
        D, I = ix.search_torch(xq, args.k, efSearch=ef)
        torch.cuda.synchronize()
        ix.set_kernel_timing(True)
        t1 = time.time()
        for _ in range(args.reps):
            ix.search_torch(xq, args.k, D=D, I=I, efSearch=ef)
        torch.cuda.synchronize()
        wall = (time.time() - t1) / args.reps
        ix.set_kernel_timing(False)
        nl1, ms1 = ix.kernel_time_stats()
        nl, ms = nl1 - nl0, ms1 - ms0
        nl0, ms0 = nl1, ms1
        ki = ix.last_kernel_info()
        Ih = I[:ns].cpu().numpy()
        rec = np.mean([len(set(a.tolist()) & set(b.tolist())) / args.k for a, b in zip(Ih, Igt)])
        kms = ms / max(nl, 1)
        print(
            "efSearch=%4d: kernel %.3f ms  wall %.3f ms  %.0f QPS  recall@%d %.4f  dist/query %.0f  expanded/query %d  %.0f GB/s (alg)  grid %d"
            % (ef, kms, wall * 1e3, args.nq / wall, args.k, rec, ki["bytes"] / (4 * args.d + 4) / args.nq, ki["nsplit"],
               ki["bytes"] / (kms * 1e-3) / 1e9, ki["grid"]),
            flush=True,
        )


if __name__ == "__main__":
    main()
