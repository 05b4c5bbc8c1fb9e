"""Summarises a rocprofv3 --pmc counter_collection.csv: per kernel name, the mean of each counter per dispatch."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(path)):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
only = sys.argv[2:] 
for k, cs in acc.items():
    if only and not any(o in k for o in only):
        continue
    print(k[:90])
    for c, v in sorted(cs.items()):
        print("   %-28s n=%-4d mean=%.4g" % (c, len(v), sum(v) / len(v)))
