"""Matrix-core counters per kernel from one rocprofv3 --pmc pass: MFMA instructions, MFMA busy cycles and the share of
the kernel's SIMD-cycles the matrix pipe was busy (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs));
GRBM_GUI_ACTIVE is reported summed over the 8 XCDs."""
import collections
import csv
import glob
import sys

f = glob.glob(f"{sys.argv[1]}/p_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    key = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[key] += 1
print(f"{'kernel':34s} {'launches':>8s} {'MFMA instr/launch':>18s} {'MFMA busy share':>16s} {'VALU issue share':>17s}")
rows = []
for k, d in agg.items():
    n = max(cnt[k], 1)
    mf = sum(v for c, v in d.items() if c.startswith("SQ_INSTS_VALU_MFMA_") and "MOPS" not in c)
    cyc = d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(cyc * 1024, 1.0)
    valu = 4.0 * d.get("SQ_ACTIVE_INST_VALU", 0.0) / max(cyc * 1024, 1.0)     # quad-cycles -> cycles
    rows.append((cyc, k, n, mf / n, busy, valu))
for cyc, k, n, mf, busy, valu in sorted(rows, reverse=True)[:24]:
    print(f"{k:34s} {n:8d} {mf:18.0f} {100 * busy:15.1f}% {100 * valu:16.1f}%")
