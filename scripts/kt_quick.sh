#!/bin/bash
# Kernel trace + stats of a short training-step run: gpurun_out/kt/p_kernel_stats.csv
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/kt
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --steps 5 --warmup 2 > $O/kt.log 2>&1
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/p_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:60]:
    n = r["Name"].split("(")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    print(f'{n[:60]:60s} calls {int(r["Calls"]):4d}  avg {float(r["AverageNs"])/1e3:9.2f} us  total/step {float(r["TotalDurationNs"])/7e6:7.3f} ms')
PY
