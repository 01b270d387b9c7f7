#!/bin/bash
# CPU-only: how FAISS's BLAS branch on OpenBLAS (oracle PATH_OPENBLAS) behaves on the GPU box's host for different query-block
# sizes and OpenBLAS thread counts (the sgemm of a 1024-row block is small; 64 pthreads + 128 OpenMP threads take turns)
O=$1
python3 - <<'PY' 2>&1 | tee $O/openblas_threads.txt
import os, sys, time
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
sys.path.insert(0, ".")
import numpy as np
from oracle import oracle as orc
print(orc.openblas_load(), "omp threads", orc.num_threads(), "cpus", os.cpu_count())
n, d = 2_000_000, 128
xb = orc.synth_uniform(n, d, 1234); xq = orc.synth_uniform(4096, d, 4321)
t = time.perf_counter(); orc.flat_search(orc.METRIC_L2, xb, xq, 10, force_path=orc.PATH_BLAS); print("port 4096 q: %.2f s" % (time.perf_counter() - t))
for thr in (64, 32, 16, 8):
    orc.openblas_set_num_threads(thr)
    for nq in (1024, 4096):
        t = time.perf_counter(); orc.flat_search(orc.METRIC_L2, xb, xq[:nq], 11, force_path=orc.PATH_OPENBLAS); dt = time.perf_counter() - t
        print("openblas threads=%d nq=%d: %.2f s -> %.0f q/s at N=2M, %.1f q/s scaled to N=10M" % (thr, nq, dt, nq / dt, nq / dt / 5))
for omp in (32, 16):
    orc.set_num_threads(omp); orc.openblas_set_num_threads(32)
    t = time.perf_counter(); orc.flat_search(orc.METRIC_L2, xb, xq, 11, force_path=orc.PATH_OPENBLAS); dt = time.perf_counter() - t
    print("openblas threads=32 omp=%d nq=4096: %.2f s -> %.1f q/s scaled to N=10M" % (omp, dt, 4096 / dt / 5))
PY
