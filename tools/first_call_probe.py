#!/usr/bin/env python3
"""What the FIRST search of a freshly filled Flat index pays on top of a steady one: run under
  rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/first_call_probe.py
then  python3 tools/first_call_probe.py --report <dir>  lists the kernels of the first search's window and the time no kernel ran.
The probe prints wall-clock marks (ns, the tracer's clock) around the first and the fourth search."""
import glob, csv, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    marks = {}
    for line in open(os.path.join(sys.argv[2], "marks.txt")):
        k, v = line.split()
        marks[k] = int(v)
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    tr = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
    for tag in ("first", "steady"):
        a, b = marks[tag + "_begin"], marks[tag + "_end"]
        ks = [(s, e, n) for s, e, n in tr if s >= a and e <= b]
        busy = sum(e - s for s, e, _ in ks)
        print(f"{tag} search: {(b - a) / 1e6:.2f} ms wall, {len(ks)} kernels, {busy / 1e6:.2f} ms of kernel time, {(b - a - busy) / 1e6:.2f} ms with no kernel running")
        agg = {}
        for s, e, n in ks:
            agg[n[:70]] = agg.get(n[:70], 0) + (e - s)
        for n, t in sorted(agg.items(), key=lambda kv: -kv[1])[:8]:
            print(f"    {t / 1e6:8.3f} ms  {n}")
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "duckdb-faiss-ext_amd", "pyhost"))
import torch
import mi355_faiss as mf
n, d, nq, k = int(os.environ.get("N", 10_000_000)), 128, 10_000, 10
ix = mf.index_factory(d, "Flat", mf.METRIC_L2)
for s0 in range(0, n, 1 << 20):
    ix.add_torch(mf.synth_uniform_torch(min(1 << 20, n - s0), d, 1234, row0=s0))
xq = mf.synth_uniform_torch(nq, d, 4321, row0=0)
D = torch.empty((nq, k), dtype=torch.float32, device="cuda:0"); I = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
torch.cuda.synchronize()
now = time.clock_gettime_ns  # (rocprofv3 timestamps: CLOCK_BOOTTIME on this stack -- both are written, the report picks what fits)
out = []
for i in range(4):
    torch.cuda.synchronize()
    t0 = now(time.CLOCK_BOOTTIME)
    ix.search_torch(xq, k, D=D, I=I); torch.cuda.synchronize()
    t1 = now(time.CLOCK_BOOTTIME)
    print(f"search {i}: {(t1 - t0) / 1e6:.2f} ms", flush=True)
    if i == 0:
        out += [("first_begin", t0), ("first_end", t1)]
    if i == 3:
        out += [("steady_begin", t0), ("steady_end", t1)]
dst = os.environ.get("MARKS_DIR")
if dst:
    open(os.path.join(dst, "marks.txt"), "w").write("".join(f"{k_} {v}\n" for k_, v in out))
