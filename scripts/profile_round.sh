#!/bin/bash
# Refresh the evidence under gpurun_out/prof (copy into profiles/ afterwards): run on the GPU box from the repo root.
#   1. bench.py with the per-entry-point table and the CPU baseline
#   2. rocprofv3 kernel trace + stats of the same command
#   3. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate counter passes (kernel trace only)
#   4. matrix-core counters of the training step (MFMA instructions, MFMA busy cycles)
#   5. the bf16 inference forward (BASELINE configs[1]): entry-point table, kernel stats, HBM traffic
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
mkdir -p $O
cd $R
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --breakdown > $O/bench_line.json 2> $O/entry_point_breakdown.txt
timeout -k 10 200 python3 scripts/bench_infer.py A 256 both > $O/infer_A.txt 2>&1
timeout -k 10 200 python3 scripts/bench_infer.py B 256 both > $O/infer_B.txt 2>&1
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-other-configs"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 $B --steps 5 --warmup 2 > $O/kt.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc/f -o p -- python3 $B --steps 2 --warmup 1 > $O/pmc_f.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc/w -o p -- python3 $B --steps 2 --warmup 1 > $O/pmc_w.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc/m -o p -- python3 $B --steps 2 --warmup 1 > $O/pmc_m.log 2>&1
python3 $R/scripts/pmc_traffic.py $O/pmc $O/hbm_traffic_per_kernel.json > $O/hbm_traffic.txt
python3 $R/scripts/pmc_mfma.py $O/pmc/m > $O/mfma_utilisation.txt
I="$R/scripts/bench_infer.py A 64 bf16"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ikt -o p -- python3 $I > $O/ikt.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/ipmc/f -o p -- python3 $I > $O/ipmc_f.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/ipmc/w -o p -- python3 $I > $O/ipmc_w.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/ipmc/m -o p -- python3 $I > $O/ipmc_m.log 2>&1
python3 $R/scripts/pmc_traffic.py $O/ipmc $O/infer_hbm_traffic_per_kernel.json > $O/infer_hbm_traffic.txt
python3 $R/scripts/pmc_mfma.py $O/ipmc/m > $O/infer_mfma_utilisation.txt
cat $O/bench_line.json
