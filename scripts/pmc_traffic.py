"""Aggregate FETCH_SIZE / WRITE_SIZE (two separate rocprofv3 --pmc passes over bench.py) into HBM bytes per kernel.
gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is doubled; both counters are in KiB."""
import csv, glob, collections, json, sys
root = sys.argv[1]
out = {}
for tag in ('f', 'w'):
    f = glob.glob(f'{root}/{tag}/p_counter_collection.csv')[0]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        key = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0].split('<')[0]
        a = agg[key]; a[0] += float(r['Counter_Value']); a[1] += 1
    out[tag] = agg
res = {}
for k in out['f']:
    fs, n = out['f'][k]; ws, _ = out['w'].get(k, [0, 0])
    res[k] = {"dispatches": n, "fetch_kib": fs / n, "write_kib": ws / n, "hbm_bytes_per_dispatch": (2 * fs + ws) * 1024 / n}
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_dispatch"] * kv[1]["dispatches"])[:20]:
    print(f"{k:34s} n={v['dispatches']:4d}  HBM {v['hbm_bytes_per_dispatch']/1e6:9.1f} MB/dispatch")
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], 'w'), indent=1)
