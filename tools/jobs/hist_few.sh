python -m pytest tests -m gpu -x -q 2>&1 | tail -1
python tools/jobs/hist_probe.py 256 2>&1 | grep -E "^poisson  |^gauss|4 values|64 values|^zeros|^random  " | sed 's/tile_summary=[0-9]* //; s/block_reduce.*//'
python tools/jobs/hist_probe.py 16 2>&1 | grep -E "^poisson  |^gauss" | sed 's/block_reduce.*//'
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['compress_GBps'], d['decompress_GBps'], d['roofline']['kernel_ms']['k_histogram'])"
timeout -k 10 200 python tests/tools_soak_parity.py 60 71 | tail -1
