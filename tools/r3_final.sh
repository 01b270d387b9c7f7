#!/bin/bash
# round 3: the committed evidence -- full GPU suite, then the profile pass
out=gpurun_out/r3p; mkdir -p $out
timeout 1500 python3 -m pytest tests -x -q -m gpu > $out/full_suite.txt 2>&1; tail -5 $out/full_suite.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
tools/profile_round3.sh headline chunk c2 c3 c3ip c4 c5 shard sens ingest pmc
