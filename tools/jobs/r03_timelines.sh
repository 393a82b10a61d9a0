# kernel timelines of a 1 GiB compress and decompress call + the SQ counter table of the round (through gpurun, from the repo root)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trc && rocprofv3 --kernel-trace --output-format csv -d /tmp/trc -- python3 $R/tools/compress_trace.py > $O/r03_compress_trace.log 2>&1
python3 $R/tools/trace_fold.py /tmp/trc 1 100 > $O/r03_timeline_compress_1GiB_b8.txt
rm -rf /tmp/trd && rocprofv3 --kernel-trace --output-format csv -d /tmp/trd -- python3 $R/tools/decompress_trace.py > $O/r03_decompress_trace.log 2>&1
python3 $R/tools/trace_fold.py /tmp/trd 1 100 > $O/r03_timeline_decompress_1GiB_b8.txt
cd $R && bash tools/pmc_sq.sh r03 > $O/r03_pmc_sq.log 2>&1
tail -3 $O/r03_pmc_sq.log
