#!/usr/bin/env python3
"""64 MiB volume (configs 1 and 3), HBM resident: a few compress + decompress calls, wall time of each.  Run it under
`rocprofv3 --kernel-trace` and fold with tools/trace_fold.py to see where a small call's time goes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from datacompressionfloat_amd import MrcZipCodec

dev = torch.device("cuda", 0)
n = 16 * 1048576 + 256
kind = sys.argv[1] if len(sys.argv) > 1 else "gauss"
g = torch.Generator(device=dev).manual_seed(7)
if kind == "poisson":
    w = torch.cat([torch.zeros(256, device=dev), torch.poisson(torch.full((n - 256,), 8.0, device=dev), generator=g)]).view(torch.int32).contiguous()
else:
    w = torch.empty(n, dtype=torch.float32, device=dev).normal_(10.0, 3.0, generator=g).view(torch.int32)
codec = MrcZipCodec(0, max_batch_chunks=int(os.environ.get("BATCH", "43")))
rec_buf = torch.empty(codec.records_bound(n), dtype=torch.uint8, device=dev)
out_buf = torch.empty(n, dtype=torch.int32, device=dev)
for it in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rec, _ = codec.compress_device(w, 8, 0, out=rec_buf)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    out, _ = codec.uncompress_device(rec, n, out=out_buf)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{kind} it{it}: compress {1e3*(t1-t0):.3f} ms ({4*n/(t1-t0)/1e9:.1f} GB/s)  decompress {1e3*(t2-t1):.3f} ms ({4*n/(t2-t1)/1e9:.1f} GB/s)", flush=True)
    time.sleep(0.002)
codec.close()
