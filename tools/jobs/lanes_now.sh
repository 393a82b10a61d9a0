for cfg in "MRCZ_LANES=2" "MRCZ_LANES=3" "MRCZ_LANES=2 MRCZ_SPLIT=60" "MRCZ_LANES=2 MRCZ_SPLIT=50"; do
  env $cfg python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['compress_GBps'], d['decompress_GBps'])"
done
