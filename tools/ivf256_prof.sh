cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for o in -1 0; do python3 bench.py --index IVF256,Flat --rows 1000000 --data clustered --no-cpu-baseline --steps 3 --warmup 1 --opt ivf_collect=$o 2>/dev/null | cut -c1-230; done
mkdir -p gpurun_out/ivf256
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ivf256 -o t -- python3 bench.py --index IVF256,Flat --rows 1000000 --data clustered --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
f=$(find gpurun_out/ivf256 -name "*kernel_stats.csv" | head -1)
python3 tools/kstats_print.py $f | head -12 | cut -c1-140
