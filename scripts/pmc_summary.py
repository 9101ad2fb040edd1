"""Summarise rocprofv3 --pmc CSV output: per kernel name, per counter: sum over dispatches / number of dispatches."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        print("   %-28s total %.6g  per-dispatch %.6g  (n=%d)" % (c, acc[k][c], acc[k][c] / cnt[k][c], cnt[k][c]))
