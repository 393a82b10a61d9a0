for cfg in "MRCZ_HUFF_SPLIT=1" "MRCZ_HUFF_SPLIT=0 MRCZ_HT=48" "MRCZ_HUFF_SPLIT=0 MRCZ_HT=16"; do
  echo "== $cfg"; env $cfg python tools/small_trace.py gauss | tail -2
done
