#!/bin/bash
# HBM traffic per compress / decompress pass: bash tools/jobs/pmc_traffic.sh <tag>   (through gpurun, from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
O=$R/gpurun_out/traffic_$TAG
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $C -d $O/$C -- python3 $R/tools/pmc_pass.py > $O/$C.out 2> $O/$C.err || { echo "$C pass failed"; tail -5 $O/$C.err; exit 1; }
done
python3 $R/tools/pmc_traffic.py $O/FETCH_SIZE $O/WRITE_SIZE 3 > $O/${TAG}_pmc_traffic.json
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
ls -la $O
