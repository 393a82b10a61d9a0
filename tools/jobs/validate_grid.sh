set -e
R=$(pwd); O=$R/gpurun_out
export MRCZ_VALIDATE_WAVE=1
for G in 2048 8192 16384 32768; do
  export MRCZ_VALIDATE_GRID=$G
  bash tools/jobs/decompress_timeline.sh > /dev/null 2>&1
  echo "== grid $G"; grep -E "k_valid" $O/r02_timeline_decompress.txt | tail -1
done
