import csv, glob, collections, sys
tag, pat = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(f'/root/repo/gpurun_out/pmc_{tag}/*/p_counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            meta = (r['VGPR_Count'], r['Accum_VGPR_Count'], r['LDS_Block_Size'], r['Scratch_Size'], r['Grid_Size'])
    for c, v in agg.items():
        print(f"{c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
    if agg: print("vgpr,agpr,lds,scratch,grid:", meta)
