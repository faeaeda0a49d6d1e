"""Summarise rocprofv3 --pmc CSV output: per kernel, mean of each counter over dispatches (skipping warm-up)."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-40:]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        v = v[len(v) // 4:]
        print(f"   {c:28s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
