import csv, glob, json, sys, collections
out = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        out[k][r['Counter_Name']] += float(r['Counter_Value'])
print(json.dumps({k: dict(v) for k, v in out.items() if 'paint_kernel_r' in k}, indent=1))
