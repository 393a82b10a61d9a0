# 64 MiB calls: kernel timelines of one compress and one decompress call (Gaussian and Poisson volumes)
set -e
R=$(pwd); O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for K in gauss poisson; do
  rm -rf /tmp/ts_$K && rocprofv3 --kernel-trace --output-format csv -d /tmp/ts_$K -- python3 $R/tools/small_trace.py $K > $O/r02_small_trace_$K.log 2>&1
  python3 $R/tools/trace_fold.py /tmp/ts_$K 2 100 > $O/r02_timeline_small_$K.txt
  tail -2 $O/r02_small_trace_$K.log
done
