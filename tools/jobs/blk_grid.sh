for G in 256 512 768; do
  MRCZ_BLK_GRID=$G python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('grid $G', d['value'], d['decompress_GBps'], d['roofline']['kernel_ms']['k_blk_count'])"
done
