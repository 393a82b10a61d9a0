#!/usr/bin/env python3
"""Driver for the HBM-traffic counter passes (run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and again under
`--pmc WRITE_SIZE`): the SURVEY 8(d) 1 GiB volume, b = 8, exactly REPS compress and REPS decompress calls in the codec's normal
mode (two compress lanes), plus one k_erase_bits over the volume as the calibration row (it reads and writes 1 GiB, 16 B per
lane).  tools/pmc_traffic.py divides every kernel's SUM over its launches by REPS: bytes per compress / decompress PASS."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from datacompressionfloat_amd import MrcZipCodec
REPS = 3
n = 1 << 28
dev = torch.device("cuda", 0)
w = torch.from_numpy(bench.make_volume(n, 1234, True).view("int32")).to(dev)
codec = MrcZipCodec(0, max_batch_chunks=43)
rec_buf = torch.empty(codec.records_bound(n), dtype=torch.uint8, device=dev)
out_buf = torch.empty(n, dtype=torch.int32, device=dev)
for _ in range(REPS):
    rec, _ = codec.compress_device(w, 8, 0, out=rec_buf)
    codec.uncompress_device(rec, n, out=out_buf)
codec.erase_bits_device(out_buf, 8, 0)
torch.cuda.synchronize()
print("passes", REPS, "zbytes", rec.numel())
