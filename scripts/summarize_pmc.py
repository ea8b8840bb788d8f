"""Summarises rocprofv3 csv output (kernel trace + pmc) per kernel name."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for sub in sorted(os.listdir(root)):
    d = os.path.join(root, sub)
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(lambda: defaultdict(int))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "?")[:60]
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                cnt[k][row["Counter_Name"]] += 1
        print("==", f)
        for k in acc:
            for c in acc[k]:
                print("  %-60s %-22s sum=%.6g  dispatches=%d  per_dispatch=%.6g" % (k, c, acc[k][c], cnt[k][c], acc[k][c] / cnt[k][c]))
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        dur = defaultdict(list)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                dur[row["Kernel_Name"][:60]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        print("==", f)
        for k, v in dur.items():
            print("  %-60s calls=%d avg_ms=%.3f min_ms=%.3f max_ms=%.3f" % (k, len(v), sum(v) / len(v) / 1e6, min(v) / 1e6, max(v) / 1e6))
