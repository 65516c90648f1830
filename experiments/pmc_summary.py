import csv, sys, collections, glob
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:60]; agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k, v in agg.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k: continue
    print(k, "launches", max(n[(k, c)] for c in v))
    for c, x in sorted(v.items()): print(f"   {c:32s} {x:.4g}")
