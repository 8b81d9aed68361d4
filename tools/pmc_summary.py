import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv') + glob.glob(sys.argv[1] + '/*counter_collection.csv')
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    k = r['Kernel_Name']
    k = k[k.find('wf_'):][:28] if 'wf_' in k else k[:28]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    agg[k]['_vgpr'] = float(r['VGPR_Count']); agg[k]['_scratch'] = float(r['Scratch_Size'])
for k, d in sorted(agg.items()):
    if 'wf_' in k or 'film' in k:
        print(k, {n: f'{v:.3g}' for n, v in sorted(d.items())})
