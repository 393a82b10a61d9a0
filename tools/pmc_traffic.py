#!/usr/bin/env python3
"""Fold two rocprofv3 PMC passes over tools/pmc_pass.py (FETCH_SIZE, WRITE_SIZE; counter_collection.csv) into the per-kernel HBM
traffic table bench.py reads for roofline.traffic.

Per kernel: launches per pass, bytes per PASS (sum over all launches of the run / passes: a compress pass launches most kernels
twice, once per lane, on different amounts of data -- the mean over launches says nothing), bytes per launch for kernels that
run once per pass.  Units and the gfx950 correction follow MI355X_MICROARCH.md's HBM section: both counters are in KB;
FETCH_SIZE reports half of the bytes of a wide coalesced streaming read (16 B per lane) and is doubled for the kernels
that read that way (WIDE below: checked on the calibration row, k_erase_bits, which reads exactly 1 GiB); kernels that
stage dwords (k_blk_count: 4 B per lane; its FETCH_SIZE equals the compressed bytes it must read) are taken as reported.
usage: pmc_traffic.py <fetch dir> <write dir> <passes>"""
import csv, glob, json, sys, collections

WIDE = {"k_tile_summary", "k_histogram", "k_emit", "k_merge_segments", "k_scan_candidates", "k_erase_bits", "k_validate_wave", "k_validate_candidates"}


def sum_per_kernel(d, counter):
    acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
            acc[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return acc, cnt


passes = int(sys.argv[3]) if len(sys.argv) > 3 else 3
fetch, nf = sum_per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = sum_per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs) -- python3 tools/pmc_pass.py: 1 GiB N(10,3) b=8, {passes} compress + "
                 f"{passes} decompress passes, MI355X.  bytes_per_pass = (sum over all launches) / {passes}; FETCH_SIZE doubled for the kernels with 16-B-per-lane "
                 "streaming loads (fetch_factor 2), as reported for the others",
       "passes": passes, "kernels": {}}
COMPRESS = {"k_tile_summary", "k_stream_scan", "k_histogram", "k_block_reduce", "k_block_index", "k_huffman", "k_huffman_hdr", "k_stream_layout", "k_pair_bits",
            "k_pair_offsets", "k_container", "k_zero_records", "k_emit", "k_emit_headers"}
tot = {"compress": 0, "decompress": 0}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("k_"):
        continue
    fac = 2 if k in WIDE else 1
    launches = max(nf.get(k, 0), nw.get(k, 0))
    per_pass_div = 1 if k == "k_erase_bits" else passes
    b = (fac * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0
    row = {"launches_per_pass": round(launches / per_pass_div, 2), "fetch_factor": fac, "fetch_bytes_per_pass": int(fac * fetch.get(k, 0.0) * 1024 / per_pass_div),
           "write_bytes_per_pass": int(write.get(k, 0.0) * 1024 / per_pass_div), "bytes_per_pass": int(b / per_pass_div)}
    if launches == per_pass_div:
        row["bytes_per_launch"] = row["bytes_per_pass"]
    out["kernels"][k] = row
    if k != "k_erase_bits":
        tot["compress" if k in COMPRESS else "decompress"] += row["bytes_per_pass"]
out["bytes_per_compress_pass"] = tot["compress"]
out["bytes_per_decompress_pass"] = tot["decompress"]
print(json.dumps(out, indent=1))
