#!/bin/bash
# usage: scripts/pmc.sh <tag> <kbench args...> ; separate counter passes
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
run() { timeout -s KILL 120 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/$2 -o p -- python3 $GRAFT_REPO_ROOT/scripts/kbench.py "${@:3}" > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.$2.log 2>&1 || exit 1; }
run "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU" a "$@"
run "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" b "$@"
run "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" c "$@"
