set -e
R=$(pwd); O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for S in 0 1; do
  export MRCZ_STAGGER=$S
  rm -rf /tmp/tr$S && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$S -- python3 $R/tools/compress_trace.py > $O/r02_compress_trace_$S.log 2>&1
  python3 $R/tools/trace_fold.py /tmp/tr$S 1 100 > $O/r02_timeline_compress_stagger$S.txt
  tail -2 $O/r02_compress_trace_$S.log
done
