"""Summarise rocprofv3 --pmc CSV output: per kernel name, mean of each counter over dispatches."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-32s n=%3d mean=%.6g" % (c, len(v), sum(v) / len(v)))
