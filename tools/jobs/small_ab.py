#!/usr/bin/env python3
"""64 MiB decompress, wall time per call, two contexts side by side: header validator one lane per candidate / one wave per candidate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from datacompressionfloat_amd import MrcZipCodec
dev = torch.device("cuda", 0)
n = 16 * 1048576 + 256
g = torch.Generator(device=dev).manual_seed(7)
w = torch.empty(n, dtype=torch.float32, device=dev).normal_(10.0, 3.0, generator=g).view(torch.int32)
cs = {}
for v in ("0", "1"):
    os.environ["MRCZ_VALIDATE_WAVE"] = v
    cs[v] = MrcZipCodec(0, max_batch_chunks=43)
rec_buf = torch.empty(cs["0"].records_bound(n), dtype=torch.uint8, device=dev)
out_buf = torch.empty(n, dtype=torch.int32, device=dev)
rec, _ = cs["0"].compress_device(w, 8, 0, out=rec_buf)
for rep in range(3):
    for v in ("0", "1"):
        ts = []
        for it in range(20):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            cs[v].uncompress_device(rec, n, out=out_buf)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ts.sort()
        print(f"validate_wave={v}: median {1e3*ts[10]:.3f} ms, min {1e3*ts[0]:.3f} ms ({4*n/ts[10]/1e9:.1f} GB/s)", flush=True)
