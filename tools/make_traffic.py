#!/usr/bin/env python3
"""profiles/<prev>_traffic.json + this round's PMC passes (tools/pmc_all.sh outputs, one <tag>_pmc.txt per workload) -> profiles/<new>_traffic.json.
usage: python tools/make_traffic.py profiles/r5_traffic.json profiles/r6_traffic.json h=profiles/r6_pmc_h.txt n8=profiles/r6_pmc_n8.txt c3=... c4=... c5=...
An entry is updated only when its workload's pass exists: FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE, KiB -> bytes,
`last` dispatch of the dominant kernel (the steady one); mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024)."""
import json, re, sys

prev, new = sys.argv[1], sys.argv[2]
passes = dict(a.split("=", 1) for a in sys.argv[3:])
tag_of = {"N=10000000 nq=10000 k=10|flat_bf16_collect_kernel": "h", "N=1250000|flat_bf16_collect_kernel": "n8", "IVF4096|ivf_bf16_collect_kernel": "c3",
          "d=768|flat_bf16_big_kernel": "c4", "HNSW|hnsw_search_kernel": "c5"}
j = json.load(open(prev))
for w in j["workloads"]:
    tag = next((t for k, t in tag_of.items() if all(p in (w["metric"] + "|" + w["kernel"]) for p in k.split("|"))), None)
    if tag == "h" and "N=1250000" in w["metric"]:
        tag = "n8"
    path = passes.get(tag)
    if not path:
        continue
    vals = {}
    for ln in open(path):
        m = re.match(r"(\w+) kernel='([^']*)' dispatches=(\d+) grid_size=(\d*) mean=([\d.e+]+) last=([\d.e+]+)", ln)
        if m and w["kernel"] in m.group(2):
            vals[m.group(1)] = (float(m.group(5)), float(m.group(6)), int(m.group(3)))
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        continue
    # `last` = the last dispatch of the run: the steady main launch (IVF: the main pass, not the pre-pass in front of it)
    f, wr = vals["FETCH_SIZE"][1], vals["WRITE_SIZE"][1]
    w["FETCH_SIZE_KiB_per_launch"], w["WRITE_SIZE_KiB_per_launch"] = f, wr
    w["hbm_bytes_per_launch"] = (2.0 * f + wr) * 1024.0
    if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "GRBM_GUI_ACTIVE" in vals:
        i = 1
        w["mfma_busy_frac"] = round(vals["SQ_VALU_MFMA_BUSY_CYCLES"][i] / (vals["GRBM_GUI_ACTIVE"][i] / 8.0 * 1024.0), 4)
        w["mfma_busy_source"] = "%s: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)" % path
    w["source"] = re.sub(r"profiles/r\d+_pmc_\w+\.txt", path, w.get("source", path)) if "profiles/" in w.get("source", "") else path
    print(tag, w["kernel"], "hbm bytes per launch %.4g" % w["hbm_bytes_per_launch"], "mfma_busy", w.get("mfma_busy_frac"))
json.dump(j, open(new, "w"), indent=1)
