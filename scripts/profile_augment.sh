#!/bin/bash
# rocprofv3 evidence for the waveform augmentation (256 pairs x 10 s, 1 s impulse responses): kernel stats and HBM traffic
# (FETCH_SIZE and WRITE_SIZE in separate counter passes).  Run on the GPU box from the repo root.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_aug
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C="$R/scripts/bench_augment.py 256 1.0"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 $C > $O/kt.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc/f -o p -- python3 $C > $O/pmc_f.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc/w -o p -- python3 $C > $O/pmc_w.log 2>&1
python3 $R/scripts/pmc_traffic.py $O/pmc $O/hbm_traffic_per_kernel.json > $O/hbm_traffic.txt
cat $O/hbm_traffic.txt
