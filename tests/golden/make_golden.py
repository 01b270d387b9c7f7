#!/usr/bin/env python3
"""Regenerates tests/golden/reference_goldens.json from the reference checkout.

Runs ONLY in the dev container (needs /root/reference).  It extracts the EXPECTED
RESULT ROWS (data) of the reference's sqllogictest files for the hot path:

  test/sql/faiss.test:19-38    Flat (default metric inner product), k=2: 20 distances
  test/sql/faiss2.test:22-41   IDMap,Flat: 20 labels (join order unspecified -> multiset)
  test/sql/faiss3.test:25-44   IDMap,Flat: 20 x (rank, label, distance)
  test/sql/faiss3.test:49-68   same with filter 'column0>100' (IDSelectorBitmap): 20 rows
  test/sql/faiss4.test:22, faiss6.test:10,30   user-visible error strings

and copies the two CSV data fixtures (training.csv: 1000 x (id, 8 floats);
queries.csv: 10 x (id, 8 floats)) verbatim.  No SQL / source text is kept.
"""
import json
import os
import re
import shutil

REF = "/root/reference/test/sql"
HERE = os.path.dirname(os.path.abspath(__file__))


def result_blocks(path):
    """Return the list of expected-result blocks (lists of lines) that follow '----'."""
    blocks, cur, on = [], None, False
    for line in open(path):
        line = line.rstrip("\n")
        if line.strip() == "----":
            cur, on = [], True
            continue
        if on:
            if line.strip() == "":
                blocks.append(cur)
                on = False
            else:
                cur.append(line)
    if on:
        blocks.append(cur)
    return blocks


def main():
    out = {}
    b = result_blocks(os.path.join(REF, "faiss.test"))
    out["flat_ip_k2_distances"] = [float(x) for x in b[0]]
    b = result_blocks(os.path.join(REF, "faiss2.test"))
    out["idmap_flat_ip_k2_labels_multiset"] = sorted(int(re.split(r"\s+", l.strip())[0]) for l in b[0])
    b = result_blocks(os.path.join(REF, "faiss3.test"))
    rows = [l.split("\t") for l in b[0]]
    out["idmap_flat_ip_k2"] = [[int(r[0]), int(r[1]), float(r[2])] for r in rows]
    rows = [l.split("\t") for l in b[1]]
    out["idmap_flat_ip_k2_filter_id_gt_100"] = [[int(r[0]), int(r[1]), float(r[2])] for r in rows]
    b = result_blocks(os.path.join(REF, "faiss4.test"))
    out["error_add_ids_on_flat"] = b[0][0]
    b = result_blocks(os.path.join(REF, "faiss6.test"))
    out["error_unknown_metric"] = b[0][0]
    out["_source"] = {
        "flat_ip_k2_distances": "test/sql/faiss.test:19-38",
        "idmap_flat_ip_k2_labels_multiset": "test/sql/faiss2.test:22-41",
        "idmap_flat_ip_k2": "test/sql/faiss3.test:25-44",
        "idmap_flat_ip_k2_filter_id_gt_100": "test/sql/faiss3.test:49-68",
        "error_add_ids_on_flat": "test/sql/faiss4.test:22",
        "error_unknown_metric": "test/sql/faiss6.test:10",
        "setup": "d=8, N=1000 rows of training.csv (col0=id, col1..8=vector), nq=10 rows of queries.csv, "
        "k=2, default metric INNER_PRODUCT (src/faiss_extension.cpp:105); the round(distance,5) "
        "columns are compared approximately by sqllogictest",
    }
    json.dump(out, open(os.path.join(HERE, "reference_goldens.json"), "w"), indent=1)
    for f in ("training.csv", "queries.csv"):
        shutil.copyfile(os.path.join(REF, f), os.path.join(HERE, f))
    print("wrote reference_goldens.json")


if __name__ == "__main__":
    main()
