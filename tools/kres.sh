#!/bin/bash
# usage: tools/kres.sh <file.hip> [name filter]  -- VGPRs / spills / scratch / occupancy of every kernel instance of one source (CPU only)
cd "$(dirname "$0")/../duckdb-faiss-ext_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage $3 2>&1 |
  awk '/Function Name/ {name=$(NF-1)} /VGPRs:/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /Occupancy/ {o=$(NF-1)} /SGPRs Spill/ {ss=$(NF-1)} /VGPRs Spill/ {vs=$(NF-1)} /LDS Size/ {print name, "VGPR", v, "AGPR", a, "scratch", s, "occ", o, "sgpr_spill", ss, "vgpr_spill", vs}' |
  { if [ -n "$2" ]; then grep "$2"; else cat; fi; } | while read n rest; do echo "$(echo $n | c++filt | cut -c1-110) $rest"; done
