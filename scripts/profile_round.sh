#!/bin/bash
# Refresh the evidence under gpurun_out/prof (copy into profiles/ afterwards): run on the GPU box from the repo root.
#   1. bench.py with the per-entry-point table and the CPU baseline
#   2. rocprofv3 kernel trace + stats of the same command
#   3. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate counter passes (kernel trace only)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
mkdir -p $O
cd $R
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --breakdown > $O/bench_line.json 2> $O/entry_point_breakdown.txt
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/kt.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc/f -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_f.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc/w -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_w.log 2>&1
python3 $R/scripts/pmc_traffic.py $O/pmc $O/hbm_traffic_per_kernel.json > $O/hbm_traffic.txt
cat $O/bench_line.json
