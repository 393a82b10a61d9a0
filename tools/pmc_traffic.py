#!/usr/bin/env python3
"""Fold two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; counter_collection.csv) into the per-kernel HBM traffic
table bench.py reads for roofline.traffic.  Units and the gfx950 correction follow MI355X_MICROARCH.md's HBM section:
both counters are in KB; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950, so it is doubled."""
import csv, glob, json, sys, collections


def mean_per_kernel(d, counter):
    acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
            acc[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return {k: acc[k] / cnt[k] for k in acc}


fetch = mean_per_kernel(sys.argv[1], "FETCH_SIZE")
write = mean_per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 "
                 "--warmup 0 --no-cpu-baseline, 1 GiB b=8, MI355X; mean KB per launch as reported; corrected_bytes = "
                 "2*FETCH_SIZE*1024 (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md HBM section) + WRITE_SIZE*1024",
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("k_"):
        continue
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    out["kernels"][k] = {"fetch_kb": round(f, 1), "write_kb": round(w, 1), "corrected_bytes": int(2 * f * 1024 + w * 1024)}
print(json.dumps(out, indent=1))
