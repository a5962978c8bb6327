#!/bin/bash
# HBM traffic per kernel of the bf16 inference forward (64 clips): FETCH_SIZE and WRITE_SIZE in separate passes
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/itraffic
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
I="$R/scripts/bench_infer.py A 64 bf16"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o p -- python3 $I > $O/f.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o p -- python3 $I > $O/w.log 2>&1
python3 $R/scripts/pmc_traffic.py $O $O/per_kernel.json | head -6
