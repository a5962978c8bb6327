#!/bin/bash
# usage: scripts/pmc.sh <tag> <kbench args...> ; two separate counter passes
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/a -o p -- python3 $GRAFT_REPO_ROOT/scripts/kbench.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.a.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/b -o p -- python3 $GRAFT_REPO_ROOT/scripts/kbench.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.b.log 2>&1 || exit 1
