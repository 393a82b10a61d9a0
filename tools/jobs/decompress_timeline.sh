set -e
R=$(pwd); O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trd && rocprofv3 --kernel-trace --output-format csv -d /tmp/trd -- python3 $R/tools/decompress_trace.py > $O/r02_decompress_trace.log 2>&1
python3 $R/tools/trace_fold.py /tmp/trd 1 100 > $O/r02_timeline_decompress.txt
tail -2 $O/r02_decompress_trace.log
