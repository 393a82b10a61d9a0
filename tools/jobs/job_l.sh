set -e
R=$(pwd); O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/small_trace.py gauss > $O/r02_l_small.log 2>&1
python3 $R/tools/small_trace.py poisson >> $O/r02_l_small.log 2>&1
BATCH=4 python3 $R/tools/small_trace.py gauss >> $O/r02_l_small.log 2>&1
rm -rf /tmp/tr && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 $R/tools/small_trace.py gauss > /dev/null 2>&1
python3 $R/tools/trace_fold.py /tmp/tr 2 100 > $O/r02_l_small_timeline.txt
cat $O/r02_l_small.log
