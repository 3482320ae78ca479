import csv,glob,sys,collections
d=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[(r["Kernel_Name"][:60], r["Counter_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
for k,v in sorted(d.items()):
    print(k, "n=%d avg=%.1f KB (raw counter)" % (len(v), sum(v)/len(v)))
