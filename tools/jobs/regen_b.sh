# round artifacts, part B: side measurements and timelines
set -e
R=$(pwd); O=$R/gpurun_out
python3 tools/sweep.py > $O/r02_sweep.json 2> $O/sweep.err || { tail -20 $O/sweep.err; exit 1; }
bash tools/jobs/small_timeline.sh
bash tools/jobs/decompress_timeline.sh
MRCZ_STAGGER=1 bash tools/jobs/compress_timeline.sh || true
cd $R
python3 tools/jobs/decode_by_mask.py > $O/r02_decode_by_mask.txt 2>&1
python3 tools/jobs/chain_validate_phases.py 16 8 > $O/r02_chain_validate_phases.txt 2>&1
python3 tests/tools_phase_profile.py 2 8 > $O/r02_blk_count_phases_b8.txt 2>&1
python3 bench.py --gib-per-gpu 64 --no-cpu-baseline > $O/r02_bench_streaming_64GiB.json 2> $O/stream.err
tail -c 600 $O/r02_bench_streaming_64GiB.json
