"""Summarise a rocprofv3 --pmc CSV directory: per kernel, sum of each counter over dispatches."""
import csv, glob, sys, collections
d = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(float))
counts = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
        counts[(k, r["Counter_Name"])] += 1
for k, cs in rows.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s %18.0f  (n=%d)" % (c, v, counts[(k, c)]))
