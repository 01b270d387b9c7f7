"""Is C5's recall@10 (0.80 at FAISS's default efConstruction 40) the DATA's or the concurrent device build's?  
The same rows -- C5's kind: clustered, sigma 1.0, L2-normalised, d = 768 -- are built into an HNSW32 graph twice:
  oracle   oracle/orc_hnsw.c, FAISS's single-thread insertion order (IndexHNSW::add -> hnsw_add_vertices, one thread)
  device   csrc/hnsw.hip, the default concurrent build (one wavefront per inserted point, per-vertex locks)
and searched at efSearch 128 for recall@10 against the exact Flat result.
usage: python tools/hnsw_graph_recall.py [rows=100000] [d=768] [nq=2000]"""
import sys
import time

import numpy as np
import torch

import mi355_faiss as mf
from oracle import oracle as orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
k, efs = 10, 128
dev = torch.device("cuda", 0)
xb = mf.synth_clustered_torch(n, d, 1234, row0=0, n_centers=1024, sigma=1.0, device=dev)
xb = xb / xb.norm(dim=1, keepdim=True)
xq = mf.synth_clustered_torch(nq, d, 4321, row0=0, n_centers=1024, sigma=1.0, device=dev)
xq = (xq / xq.norm(dim=1, keepdim=True)).contiguous()
xb_h, xq_h = xb.cpu().numpy(), xq.cpu().numpy()
flat = mf.index_factory(d, "Flat", mf.METRIC_L2)
flat.add_torch(xb)
_, gt = flat.search_torch(xq, k)
gt = gt.cpu().numpy()


def recall(I):
    return float(np.mean([len(set(a.tolist()) & set(b.tolist())) / k for a, b in zip(I, gt)]))


print(f"# HNSW32 recall@{k} at efSearch {efs}: N={n} d={d} nq={nq}, clustered sigma 1.0, normalised (C5's rows); ground truth = Flat L2")
for efc in (40, 200):
    t0 = time.perf_counter()
    o = orc.Index(d, "HNSW32", orc.METRIC_L2)
    o.hnsw_set_ef_construction(efc)
    o.add(xb_h)
    t_o = time.perf_counter() - t0
    _, Io = o.search(xq_h, k, efSearch=efs)
    t0 = time.perf_counter()
    g = mf.index_factory(d, "HNSW32", mf.METRIC_L2)
    g.set_ef_construction(efc)
    g.add_torch(xb)
    torch.cuda.synchronize()
    t_g = time.perf_counter() - t0
    _, Ig = g.search_torch(xq, k, efSearch=efs)
    g1 = mf.index_factory(d, "HNSW32", mf.METRIC_L2)
    g1.set_ef_construction(efc)
    g1.set_option("hnsw_build_waves", 1)  # FAISS's single-thread insertion order on the device: the oracle's graph bit for bit
    t0 = time.perf_counter()
    g1.add_torch(xb)
    torch.cuda.synchronize()
    t_g1 = time.perf_counter() - t0
    _, Ig1 = g1.search_torch(xq, k, efSearch=efs)
    print(f"efConstruction {efc:3d}: oracle single-thread graph recall {recall(Io):.4f} (built in {t_o:.1f} s on the host) | "
          f"device single-wave graph {recall(Ig1.cpu().numpy()):.4f} ({t_g1:.1f} s) | device concurrent graph {recall(Ig.cpu().numpy()):.4f} ({t_g:.2f} s)", flush=True)
