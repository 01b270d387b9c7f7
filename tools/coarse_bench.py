"""IVF coarse quantiser A/B at C3's shape (nq x nlist x d = 10 000 x 4 096 x 128, nprobe 32): whole-search step with
ivf_coarse_bf16 = 1 (csrc/coarse_bf16.hip) vs 0 (distance matrix + selection, csrc/coarse_select.hip); results must be bit-equal.
usage: python tools/coarse_bench.py [rows] [nq] [nlist] [nprobe]"""
import sys
import time

import torch

import mi355_faiss as mf

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
nlist = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
nprobe = int(sys.argv[4]) if len(sys.argv) > 4 else 32
d, k = 128, 10
dev = torch.device("cuda", 0)
xb = mf.synth_clustered_torch(n, d, 1234, row0=0, n_centers=1024, sigma=0.1, device=dev)
xq = mf.synth_clustered_torch(nq, d, 4321, row0=0, n_centers=1024, sigma=0.1, device=dev)
ix = mf.index_factory(d, f"IVF{nlist},Flat", mf.METRIC_L2)
ix.train(xb[: min(n, 256 * nlist)].cpu().numpy())
for i0 in range(0, n, 1 << 20):
    ix.add_torch(xb[i0 : i0 + (1 << 20)])
res = {}
for mode in (1, 0, 1, 0):
    ix.set_option("ivf_coarse_bf16", mode)
    for _ in range(3):
        D, I = ix.search_torch(xq, k, nprobe=nprobe)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        D, I = ix.search_torch(xq, k, nprobe=nprobe)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"ivf_coarse_bf16={mode}: {ms:.3f} ms per {nq}-query search (N={n}, nlist={nlist}, nprobe={nprobe})", flush=True)
    res.setdefault(mode, (D.clone(), I.clone()))
same = torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][0].view(torch.int32), res[1][0].view(torch.int32))
ix.set_option("ivf_coarse_bf16", 1)
ix.search_torch(xq, k, nprobe=nprobe)
print("bit-equal:", same, " coarse_bf16_queries:", ix.get_stat("coarse_bf16_queries"), " exhaustive:", ix.get_stat("coarse_bf16_exhaustive"),
      " candidates per query (last search):", ix.get_stat("coarse_bf16_candidates") / nq)
sys.exit(0 if same else 1)
