set -e
R=$(pwd); O=$R/gpurun_out
for H in 1 4; do
  export MRCZ_HIST_WAVES=$H
  bash tools/jobs/small_timeline.sh > /dev/null 2>&1
  echo "== MRCZ_HIST_WAVES=$H"
  grep -E "k_histogram" $O/r02_timeline_small_gauss.txt | tail -2
  grep -E "k_histogram" $O/r02_timeline_small_poisson.txt | tail -2
  python tools/small_trace.py poisson | tail -4
done
