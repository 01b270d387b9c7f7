#!/usr/bin/env python3
"""Kernel micro-benchmark: times the dominant flat-search kernel (HIP events on its stream) on synthetic data.
    python tools/kbench.py --n 2000000 --nq 10000 --d 128 --k 10 --metric L2 --reps 5 [--opt key=value ...]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "duckdb-faiss-ext_amd", "pyhost"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2_000_000)
    ap.add_argument("--nq", type=int, default=10_000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric", default="L2")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--check", action="store_true", help="compare a query subsample with the oracle")
    ap.add_argument("--sweep", default="", help="key=v1,v2,...: repeat the timing for each value of one option (same index)")
    ap.add_argument("--sel-frac", type=float, default=0.0, help="> 0: IDSelectorBitmap keeping about this fraction of the rows")
    args = ap.parse_args()
    import torch

    import mi355_faiss as mf

    metric = mf.METRIC_L2 if args.metric == "L2" else mf.METRIC_INNER_PRODUCT
    ix = mf.index_factory(args.d, "Flat", metric)
    for o in args.opt:
        key, v = o.split("=")
        ix.set_option(key, int(v))
    slab = 1 << 20
    for s0 in range(0, args.n, slab):
        xb = mf.synth_uniform_torch(min(slab, args.n - s0), args.d, 1234, row0=s0)
        ix.add_torch(xb)
        torch.cuda.synchronize()
    xq = mf.synth_uniform_torch(args.nq, args.d, 4321)
    sel = None
    if args.sel_frac > 0:
        import numpy as np

        rs = np.random.RandomState(7)
        sel = ("bitmap", np.packbits(rs.rand((args.n + 7) // 8 * 8) < args.sel_frac, bitorder="little"))
    import time

    sweep = [None]
    if args.sweep:
        skey, svals = args.sweep.split("=")
        sweep = [int(v) for v in svals.split(",")]
    for sv in sweep:
        if sv is not None:
            ix.set_option(skey, sv)
        D, I = ix.search_torch(xq, args.k, sel=sel)
        torch.cuda.synchronize()
        n0, ms0 = ix.kernel_time_stats()
        ix.set_kernel_timing(True)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            ix.search_torch(xq, args.k, D=D, I=I, sel=sel)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / args.reps * 1e3
        ix.set_kernel_timing(False)
        n, ms = ix.kernel_time_stats()
        n, ms = n - n0, ms - ms0
        ki = ix.last_kernel_info()
        avg = ms / n
        tf = ki["flops"] / (avg * 1e-3) / 1e12
        gbs = ki["bytes"] / (avg * 1e-3) / 1e9
        extra = ""
        if ki["name"].startswith("flat_bf16"):
            extra = f"  prefilter={ix.prefilter_stats()} collect={ix.collect_stats()}"
        print(
            f"{ki['name']} opts={args.opt} {args.sweep.split('=')[0]}={sv} n={args.n} nq={args.nq} d={args.d} k={args.k} {args.metric}: "
            f"{avg:.3f} ms/launch  wall {wall:.3f} ms/search  {tf:.2f} TFLOP/s algorithmic  {gbs:.1f} GB/s algorithmic  "
            f"grid={ki['grid']} lds={ki['lds_bytes']} nsplit={ki['nsplit']}  qps={args.nq/(wall*1e-3):.0f}{extra}",
            flush=True,
        )
    if args.check:
        import numpy as np

        from oracle import oracle as orc

        xb_h = orc.synth_uniform(args.n, args.d, 1234)
        sub = np.arange(0, args.nq, max(1, args.nq // 256))[:256]
        Do, Io = orc.flat_search(metric, xb_h, xq[sub].cpu().numpy(), args.k, sel=sel,
                                 force_path=orc.PATH_AUTO if (sel is not None or args.nq < 20) else orc.PATH_BLAS)
        ok = np.array_equal(I.cpu().numpy()[sub], Io) and np.array_equal(D.cpu().numpy()[sub], Do)
        print("check vs oracle (", len(sub), "queries ):", "BIT-EXACT" if ok else "MISMATCH")


if __name__ == "__main__":
    main()
