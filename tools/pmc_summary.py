#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc run (counter_collection.csv) per kernel: mean counter value per launch."""
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    print(k, " ".join(f"{c}={acc[k][c] / cnt[k][c]:.4g}" for c in sorted(acc[k])), f"launches={max(cnt[k].values())}")
