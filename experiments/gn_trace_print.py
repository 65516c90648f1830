import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
for r in csv.DictReader(open(f)):
    if "gn_" in r["Name"]:
        print(f'{float(r["AverageNs"])/1e3:8.1f} us  min {float(r["MinNs"])/1e3:6.1f}  x{r["Calls"]}  {r["Name"][:70]}')
