"""Per-kernel summary of a rocprofv3 --kernel-trace --stats run (gpurun_out/kt by default): name, calls, average, ms per step."""
import csv
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/kt"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
pat = sys.argv[3:] 
rows = list(csv.DictReader(open(f"{d}/p_kernel_stats.csv")))
for r in rows:
    n = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    n = n.split("(")[0]
    if pat and not any(p in n for p in pat):
        continue
    print(f'{n[:64]:64s} calls {int(r["Calls"]):4d}  avg {float(r["AverageNs"]) / 1e3:9.2f} us  min {float(r["MinNs"]) / 1e3:8.2f}  '
          f'per step {float(r["TotalDurationNs"]) / steps / 1e6:7.3f} ms')
