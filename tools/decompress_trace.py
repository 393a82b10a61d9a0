#!/usr/bin/env python3
"""1 GiB b=8 volume, HBM resident: a few decompress calls (run under `rocprofv3 --kernel-trace`, fold with tools/trace_fold.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from datacompressionfloat_amd import MrcZipCodec
dev = torch.device("cuda", 0)
n = 1 << 28
bits = int(os.environ.get("BITS", "8"))
g = torch.Generator(device=dev).manual_seed(1234)
w = torch.empty(n, dtype=torch.float32, device=dev).normal_(10.0, 3.0, generator=g).view(torch.int32)
codec = MrcZipCodec(0, max_batch_chunks=43)
rec, _ = codec.compress_device(w, bits, 0)
out_buf = torch.empty(n, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
time.sleep(0.01)
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    codec.uncompress_device(rec, n, out=out_buf)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"it{it}: decompress {1e3*(t1-t0):.3f} ms", flush=True)
    time.sleep(0.005)
codec.close()
