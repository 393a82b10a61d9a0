for S in 45 50 55 45 50 55; do
  MRCZ_SPLIT=$S python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split $S', d['value'], d['compress_GBps'], d['decompress_GBps'])"
done
