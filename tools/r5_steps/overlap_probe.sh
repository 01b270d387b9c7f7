#!/bin/bash
(timeout 1200 python3 tools/ivf_overlap_probe.py 2>&1 | tail -6; timeout 600 python3 tools/ivf_overlap_probe.py --index Flat --rows 1250000 2>&1 | tail -6) | tee $1/overlap_probe.txt
