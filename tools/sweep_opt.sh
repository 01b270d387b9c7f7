#!/bin/bash
# usage: tools/sweep_opt.sh <option> "<v1 v2 ...>" [bench args]: one bench line per value of an index option
opt=$1; vals=$2; shift; shift
cd $GRAFT_REPO_ROOT
for v in $vals; do
  timeout 400 python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --opt $opt=$v "$@" 2>/dev/null | tail -1 > /tmp/sweep_line.json
  python3 - "$opt=$v" <<'PY'
import json, sys
j = json.loads(open("/tmp/sweep_line.json").read())
print(sys.argv[1], j["value"], j["ms_per_step"], j["roofline"]["avg_launch_ms"])
PY
done
