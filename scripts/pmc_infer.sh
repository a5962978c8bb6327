#!/bin/bash
# usage: scripts/pmc_infer.sh <tag> <bench_infer args...> ; counter passes over scripts/bench_infer.py (kernel trace only)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
run() { timeout -s KILL 150 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/$2 -o p -- python3 $GRAFT_REPO_ROOT/scripts/bench_infer.py "${@:3}" > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.$2.log 2>&1 || exit 1; }
run "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU" a "$@"
run "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" b "$@"
