"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per dispatch for each kernel."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            if filt and filt not in k:
                continue
            acc[k[:90]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):4d} mean={sum(v) / len(v):16.1f}")
