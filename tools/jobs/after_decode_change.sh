# after a decoder change: GPU parity tests, 64 MiB timelines, the 1 GiB decompress timeline, the bench line
set -e
R=$(pwd); O=$R/gpurun_out
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -1 $O/gpu_tests.log
bash tools/jobs/small_timeline.sh
bash tools/jobs/decompress_timeline.sh
cd $R && python bench.py --no-cpu-baseline > $O/bench_now.json 2>$O/bench_now.err; cat $O/bench_now.json
