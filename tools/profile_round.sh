#!/bin/bash
# Regenerate the profiles/ artifacts of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01
# 1. bench.py JSON line, 2. rocprofv3 --kernel-trace --stats of the same command, 3. two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE) over tools/pmc_pass.py (three compress + three decompress passes), folded into
# <round>_pmc_traffic.json (bytes per PASS) by tools/pmc_traffic.py.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
O=$R/gpurun_out/prof_$TAG
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/${TAG}_bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline > $O/${TAG}_bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch -- python3 $R/tools/pmc_pass.py > /dev/null 2> $O/fetch.err
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write -- python3 $R/tools/pmc_pass.py > /dev/null 2> $O/write.err
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/${TAG}_kernel_stats_bench_1GiB_b8.csv
python3 $R/tools/pmc_traffic.py $O/fetch $O/write 3 > $O/${TAG}_pmc_traffic.json
rm -rf $O/stats $O/fetch $O/write
ls -la $O
