#!/usr/bin/env python3
"""Fold the SQ counter passes of tools/pmc_sq.sh (rocprofv3 counter_collection.csv, one directory per pass) into one
per-kernel table: mean counter value per launch plus the derived shares that say what bounds a kernel.

SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md, cycle
constants table); the derived figures below are ratios, so the unit cancels:
  valu_share   SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES     share of a wave's lifetime spent issuing VALU
  lds_share    SQ_ACTIVE_INST_LDS  / SQ_WAVE_CYCLES
  wait_share   SQ_WAIT_ANY         / SQ_WAVE_CYCLES     parked on s_waitcnt / barrier
  stall_share  SQ_WAIT_INST_ANY    / SQ_WAVE_CYCLES     ready but not issued (pipe busy, dependency)
  valu_busy    SQ_ACTIVE_INST_VALU * 4 / (SQ_BUSY_CU_CYCLES-like: SQ_BUSY_CYCLES) is not used: per-SIMD issue occupancy
               is instead given as  insts_valu * 2 cycles (wave64 on SIMD-32) / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs)
  lds_conflict SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE share of LDS-array cycles that are conflict replays
"""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
        if not k.startswith("k_"):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
out = {"source": "rocprofv3 --kernel-trace --pmc <8 SQ counters per pass> -- python3 bench.py --steps 1 --warmup 0 "
                 "--no-cpu-baseline (tools/pmc_sq.sh), MI355X; mean per launch",
       "kernels": {}}
for k in sorted(acc):
    m = {c: acc[k][c] / cnt[k][c] for c in acc[k]}
    row = {c: (int(v) if v == int(v) else round(v, 1)) for c, v in sorted(m.items())}
    row["launches"] = max(cnt[k].values())
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    der = {}
    if wc:
        for name, c in (("valu_share", "SQ_ACTIVE_INST_VALU"), ("lds_share", "SQ_ACTIVE_INST_LDS"), ("sca_share", "SQ_ACTIVE_INST_SCA"),
                        ("vmem_share", "SQ_ACTIVE_INST_VMEM"), ("any_share", "SQ_ACTIVE_INST_ANY"), ("wait_share", "SQ_WAIT_ANY"),
                        ("stall_share", "SQ_WAIT_INST_ANY"), ("lds_stall_share", "SQ_WAIT_INST_LDS")):
            if c in m:
                der[name] = round(m[c] / wc, 4)
    if m.get("SQ_LDS_IDX_ACTIVE"):
        der["lds_conflict"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"], 4)
    if m.get("GRBM_GUI_ACTIVE") and "SQ_INSTS_VALU" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0  # sum over 8 XCDs -> shader cycles of the launch
        der["kernel_cycles"] = int(cyc)
        der["valu_issue_occupancy"] = round(m["SQ_INSTS_VALU"] * 2.0 / (cyc * 1024.0), 4)  # 2 cyc per wave64 VALU op, 1024 SIMDs
        if "SQ_LDS_IDX_ACTIVE" in m:
            der["lds_array_occupancy"] = round(m["SQ_LDS_IDX_ACTIVE"] / (cyc * 256.0), 4)   # per-CU LDS array cycles / (cycles * 256 CUs)
    row["derived"] = der
    out["kernels"][k] = row
print(json.dumps(out, indent=1))
