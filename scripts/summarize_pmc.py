"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel name: mean counter value per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row.get("Kernel_Name", "?")[:60]
                c = row.get("Counter_Name", "?")
                v = float(row.get("Counter_Value", 0) or 0)
                a = acc[k][c]
                a[0] += v
                a[1] += 1
        print("==", path)
        for k, cs in sorted(acc.items()):
            print(k, {c: (round(a[0] / max(a[1], 1), 1), a[1]) for c, a in cs.items()})
