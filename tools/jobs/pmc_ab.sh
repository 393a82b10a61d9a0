#!/bin/bash
# SQ counters of k_blk_count for several builds of the kernels (datacompressionfloat_amd/lib/ab/<name>.so):
#   bash tools/jobs/pmc_ab.sh name1 name2 ...     (through gpurun, from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_ab
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_IFETCH GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU"
for v in "$@"; do
  export MRCZ_LIB_PATH=$R/datacompressionfloat_amd/lib/ab/$v.so
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i + 1))
    rm -rf $O/$v.p$i
    timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc $P -d $O/$v.p$i -- python3 $R/tools/decompress_trace.py > $O/$v.p$i.out 2> $O/$v.p$i.err || { echo "$v pass $i failed"; tail -3 $O/$v.p$i.err; exit 1; }
  done
  python3 - $O $v <<'PY'
import csv, glob, sys, collections
o, v = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob(f"{o}/{v}.p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_blk_count" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
print(v, " ".join(f"{k}={acc[k]/cnt[k]/1e6:.1f}M" for k in sorted(acc)))
PY
  for i in 1 2 3; do rm -rf $O/$v.p$i; done
done
