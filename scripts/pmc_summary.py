"""Summarises rocprofv3 --pmc counter_collection.csv files: mean per dispatch of each counter for kernels matching a substring."""
import csv, sys, collections
key = sys.argv[1]
for path in sys.argv[2:]:
    acc = collections.defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if key in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
