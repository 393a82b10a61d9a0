# the two header validators side by side: GPU tests with the wave version, timelines and bench with both
set -e
R=$(pwd); O=$R/gpurun_out
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -1 $O/gpu_tests.log
for V in 0 1; do
  export MRCZ_VALIDATE_WAVE=$V
  bash tools/jobs/small_timeline.sh > /dev/null 2>&1
  bash tools/jobs/decompress_timeline.sh > /dev/null 2>&1
  echo "== MRCZ_VALIDATE_WAVE=$V"
  grep -E "k_valid" $O/r02_timeline_small_gauss.txt | tail -1
  grep -E "k_valid|k_blk_count|burst" $O/r02_timeline_decompress.txt | tail -3
  cd $R && python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['compress_GBps'], d['decompress_GBps'])"
done
unset MRCZ_VALIDATE_WAVE
timeout -k 10 200 python tests/tools_soak_parity.py 60 11 | tail -1
