import csv, collections, glob, sys
tag = sys.argv[1]
for f in glob.glob(f"gpurun_out/pmc_{tag}/*/*counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'branch_kernel' in k and '51' in k.split('(')[0]:
            agg[r['Counter_Name']]['v'].append(float(r['Counter_Value']))
            agg[r['Counter_Name']]['t'].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    for c, v in sorted(agg.items()):
        print(tag, c, round(sum(v['v']) / len(v['v'])), 'dur_us', round(sum(v['t']) / len(v['t']) / 1e3))
