#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES -d /tmp/p_a -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAIT_INST_LDS -d /tmp/q_a -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY -d /tmp/r_a -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
for d in p_a q_a r_a; do python3 $R/tools/pmc_summary.py /tmp/$d | grep -E "^k_(blk_count|emit|histogram) "; done
