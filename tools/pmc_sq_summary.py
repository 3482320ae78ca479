import csv, collections, sys, glob
d = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        e = d[r["Kernel_Name"]][r["Counter_Name"]]
        e[0] += 1; e[1] += float(r["Counter_Value"])
for k, c in d.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k: continue
    print(k[:80])
    for n, (cnt, v) in sorted(c.items()):
        print("   %-28s launches=%d  per-launch=%.4g" % (n, cnt, v / cnt))
