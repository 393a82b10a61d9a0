#!/usr/bin/env python3
"""Print the last bursts of a rocprofv3 --kernel-trace CSV as a timeline (start, end, duration in microseconds
relative to the burst's first kernel, queue id): what overlaps what inside one compress / decompress call.
A burst = kernels separated from the previous ones by more than `gap_us` of idle device."""
import csv, glob, sys
d = sys.argv[1]
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 4
gap_us = float(sys.argv[3]) if len(sys.argv) > 3 else 150.0
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r.get("Queue_Id", "?")))
rows.sort()
cuts, end = [0], rows[0][1] if rows else 0
for i in range(1, len(rows)):
    if rows[i][0] - end > gap_us * 1e3:
        cuts.append(i)
    end = max(end, rows[i][1])
cuts.append(len(rows))
for a, b in list(zip(cuts, cuts[1:]))[-nlast:]:
    t0 = rows[a][0]
    print("---- burst of", b - a, "kernels, total", round((max(r[1] for r in rows[a:b]) - t0) / 1e3, 1), "us")
    for r in rows[a:b]:
        print(f"{(r[0]-t0)/1e3:9.1f} {(r[1]-t0)/1e3:9.1f} {(r[1]-r[0])/1e3:8.1f}  q{r[3]} {r[2]}")
