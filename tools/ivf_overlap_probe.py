#!/usr/bin/env python3
"""Does a C3 step gain from running two half batches CONCURRENTLY (two index objects, two host threads, two streams)?
The stages of one IVF search are a dependency chain of kernels bound by different resources (coarse matrix: f32 MFMA; list scan: HBM;
exact re-scoring: gather latency); two half batches in flight overlap them.  Probe with a deep copy of the index:
    python tools/ivf_overlap_probe.py [--rows 10000000] [--reps 40]
"""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "duckdb-faiss-ext_amd", "pyhost"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--nq", type=int, default=10_000)
    ap.add_argument("--index", default="IVF4096,Flat")
    args = ap.parse_args()
    import torch

    import mi355_faiss as mf

    d, n, nq = 128, args.rows, args.nq
    dev = torch.device("cuda", 0)
    ix = mf.index_factory(d, args.index, mf.METRIC_L2)
    ivf = "IVF" in args.index
    if ivf:
        xb = mf.synth_clustered_torch(n, d, 1234, row0=0, n_centers=1024, sigma=0.1, device=dev)
        ix.train(xb[: min(n, 2_000_000)].cpu().numpy())
        xq = mf.synth_clustered_torch(nq, d, 4321, row0=0, n_centers=1024, sigma=0.1, device=dev)
    else:
        xb = mf.synth_uniform_torch(n, d, 1234)
        xq = mf.synth_uniform_torch(nq, d, 4321)
    for s0 in range(0, n, 1 << 20):
        ix.add_torch(xb[s0 : s0 + (1 << 20)])
    torch.cuda.synchronize()
    del xb
    twin = ix.clone_to_gpu(0)
    k = 10
    kw = {"nprobe": 32} if ivf else {}

    def run(index, q, reps, out):
        st = torch.cuda.Stream(device=dev)  # (the caller's stream: one per thread, or the two searches serialise on it)
        D = torch.empty((q.shape[0], k), dtype=torch.float32, device=dev)
        I = torch.empty((q.shape[0], k), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        for _ in range(3):
            index.search_torch(q, k, D=D, I=I, stream=st.cuda_stream, **kw)
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            index.search_torch(q, k, D=D, I=I, stream=st.cuda_stream, **kw)
        st.synchronize()
        out.append((time.perf_counter() - t0) / reps * 1e3)

    o = []
    run(ix, xq, args.reps, o)
    print("one object, %d queries per call: %.3f ms per batch" % (nq, o[0]))
    o = []
    run(ix, xq[: nq // 2].contiguous(), args.reps, o)
    print("one object, %d queries per call: %.3f ms per half batch" % (nq // 2, o[0]))
    for parts in (2, 4):
        objs = [ix, twin] + [ix.clone_to_gpu(0) for _ in range(parts - 2)]
        per = nq // parts
        outs = [[] for _ in range(parts)]
        th = [threading.Thread(target=run, args=(objs[i], xq[i * per : (i + 1) * per].contiguous(), args.reps, outs[i])) for i in range(parts)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        print("%d objects x %d queries concurrently: %s ms per part-batch each; all %d queries every %.3f ms" % (
            parts, per, ["%.3f" % v[0] for v in outs], nq, max(v[0] for v in outs)))


if __name__ == "__main__":
    main()
