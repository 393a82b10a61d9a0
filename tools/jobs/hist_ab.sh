# k_histogram with one and with four waves per segment: parity suite with each forced, small-call times, bench with each
set -e
R=$(pwd); O=$R/gpurun_out
for H in 1 4; do
  MRCZ_HIST_WAVES=$H python -m pytest tests/test_gpu_parity.py -x -q > $O/gpu_tests_h$H.log 2>&1 || { tail -30 $O/gpu_tests_h$H.log; exit 1; }
  echo "MRCZ_HIST_WAVES=$H: $(tail -1 $O/gpu_tests_h$H.log)"
  MRCZ_HIST_WAVES=$H python tools/small_trace.py gauss | tail -1
  MRCZ_HIST_WAVES=$H python tools/small_trace.py poisson | tail -1
  MRCZ_HIST_WAVES=$H python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['compress_GBps'], d['decompress_GBps'], d['roofline']['kernel_ms']['k_histogram'])"
done
python -m pytest tests -m gpu -x -q 2>&1 | tail -1
