#!/bin/bash
# SQ counter passes for the three entropy kernels (k_blk_count, k_emit, k_huffman) and everything else bench.py launches:
#   tools/pmc_sq.sh r02            (through gpurun, from the repo root)
# Each pass is its own rocprofv3 run with --kernel-trace only (8 SQ slots per pass on gfx950, MI355X_MICROARCH.md
# "rocprofv3 PMC slots"); the program itself follows "--".  Folded into <round>_pmc_sq.json by tools/pmc_sq_fold.py.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}
shift
EXTRA="$@"
O=$R/gpurun_out/sq_$TAG
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS"
P4="GRBM_GUI_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_LDS_ATOMIC_RETURN SQ_IFETCH"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
    i=$((i + 1))
    echo "pass $i: $P"
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $P -d $O/p$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline $EXTRA > $O/p$i.out 2> $O/p$i.err
    rc=$?
    echo "pass $i rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $i timed out: stopping"; exit $rc; fi
done
python3 $R/tools/pmc_sq_fold.py $O > $O/${TAG}_pmc_sq.json
# keep only the folded table and the stderr tails
for i in 1 2 3 4; do rm -rf $O/p$i; done
ls -la $O
