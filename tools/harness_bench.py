#!/usr/bin/env python3
"""Secondary report (last row): the shapes of the reference's Go harness (go/benches_c.go) on synthetic
data -- index IDMap,HNSW128,Flat, d=1536, default metric inner product, efSearch/efConstruction at FAISS's defaults
(the harness's parameter keys are misspelt and ignored), query batches of 43 (all TREC-DL19 topics in one chunk:
run_post / run_sel) or 1 (run_post_one), k swept from 11 to 2000 (main_test.go:26-32).
    python tools/harness_bench.py --n 1000000 [--sel-frac 0.1]
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
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--d", type=int, default=1536)
    ap.add_argument("--index", default="IDMap,HNSW128,Flat")
    ap.add_argument("--ks", default="11,20,50,100,200,500,1000,2000")
    ap.add_argument("--sel-frac", type=float, default=0.1, help="selectivity of the run_sel bitmap")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--opt", action="append", default=[], help="index option key=value (e.g. hnsw_reg_lists=0)")
    args = ap.parse_args()
    import numpy as np
    import torch

    import mi355_faiss as mf

    d, n = args.d, args.n
    ix = mf.index_factory(d, args.index, mf.METRIC_INNER_PRODUCT)
    for o in args.opt:
        key, v = o.split("=")
        ix.set_option(key, int(v))
    fl = mf.index_factory(d, "IDMap,Flat", mf.METRIC_INNER_PRODUCT)
    tb = 0.0
    for s0 in range(0, n, 65536):
        m = min(65536, n - s0)
        xb = mf.synth_clustered_torch(m, d, 1234, row0=s0, n_centers=1024, sigma=1.0)
        xb /= xb.norm(dim=1, keepdim=True)
        ids = torch.arange(s0, s0 + m, dtype=torch.int64, device=xb.device)
        torch.cuda.synchronize()
        t0 = time.time()
        ix.add_torch(xb, ids=ids)
        torch.cuda.synchronize()
        tb += time.time() - t0
        fl.add_torch(xb, ids=ids)
    print("build %s d=%d N=%d: %.1f s (%.0f rows/s)" % (args.index, d, n, tb, n / tb), flush=True)
    xq = mf.synth_clustered_torch(43, d, 4321, n_centers=1024, sigma=1.0)
    xq /= xq.norm(dim=1, keepdim=True)
    rs = np.random.RandomState(3)
    bitmap = np.packbits(rs.rand((n + 7) // 8 * 8) < args.sel_frac, bitorder="little")
    batch_rows = {}
    for name, nq, sel in (("run_post (43 queries)", 43, None), ("run_post_one (1 query)", 1, None),
                          ("run_sel (43 queries, %.0f%% bitmap)" % (100 * args.sel_frac), 43, ("bitmap", bitmap))):
        print(name)
        for k in [int(v) for v in args.ks.split(",")]:
            q = xq[:nq].contiguous()
            D, I = ix.search_torch(q, k, sel=sel)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                ix.search_torch(q, k, D=D, I=I, sel=sel)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.reps
            Dg, Ig = fl.search_torch(q, k, sel=sel)
            torch.cuda.synchronize()
            got, want = I.cpu().numpy(), Ig.cpu().numpy()
            per_q = [len(set(a[a >= 0].tolist()) & set(b[b >= 0].tolist())) / max(1, (b >= 0).sum()) for a, b in zip(got, want)]
            rec = np.mean(per_q)
            extra = ""
            if nq == 43 and sel is None:
                batch_rows[k] = (got[0].copy(), per_q[0])
                extra = "  (query 0 alone: %.3f, min %.3f, max %.3f over the 43)" % (per_q[0], min(per_q), max(per_q))
            elif nq == 1 and k in batch_rows:
                # the single query IS query 0 of the batch: same graph, same walk -> the same rows; its recall is that query's, not the
                # batch mean (0.000 next to a batch mean of 0.27 at efSearch 16 on 8.8 M rows)
                extra = "  (row 0 of the 43-query batch returned the same labels: %s; its recall there: %.3f)" % (
                    bool(np.array_equal(got[0], batch_rows[k][0])), batch_rows[k][1])
            print("   k=%4d: %8.3f ms per batch  %9.0f queries/s  recall@k %.3f%s" % (k, dt * 1e3, nq / dt, rec, extra), flush=True)


if __name__ == "__main__":
    main()
