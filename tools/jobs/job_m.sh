set -e
python -m pytest tests -m gpu -x -q > gpurun_out/r02_m_pytest.log 2>&1 || { tail -20 gpurun_out/r02_m_pytest.log; exit 1; }
tail -2 gpurun_out/r02_m_pytest.log
python tests/tools_huff_profile.py
python bench.py --steps 10 --warmup 3 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms']; print(d['value'], d['ms_per_step'], 'emit', k['k_emit'], 'huff', k['k_huffman'], 'hist', k['k_histogram'])"
python tools/tune.py compress 2>/dev/null | head -1
python tools/small_trace.py gauss | tail -2
