"""Summarise rocprofv3 --pmc csv output: per kernel name, mean of each counter over dispatches."""
import csv
import glob
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "conv" not in k and "score" not in k and "wgrad" not in k:
                continue
            acc[k[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        print(k)
        for c, v in sorted(cs.items()):
            print(f"   {c:32s} {sum(v) / len(v):16.0f}  (n={len(v)})")
