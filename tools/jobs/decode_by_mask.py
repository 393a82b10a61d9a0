"""Developer tool (GPU): decode time and fallback count of the 1 GiB volume at several mask levels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from datacompressionfloat_amd import MrcZipCodec
dev = torch.device("cuda", 0)
n = 1 << 28
g = torch.Generator(device=dev).manual_seed(1234)
w = torch.empty(n, dtype=torch.float32, device=dev).normal_(10.0, 3.0, generator=g).view(torch.int32)
w[:256] = 0
codec = MrcZipCodec(0, max_batch_chunks=43)
for bits in (0, 8, 12, 16, 23):
    rec, _ = codec.compress_device(w, bits, 0)
    out_buf = torch.empty(n, dtype=torch.int32, device=dev)
    best = 1e9
    for it in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out, _ = codec.uncompress_device(rec, n, out=out_buf)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    exp = w.clone(); codec.erase_bits_device(exp, bits, 0)
    print(bits, f"{1e3*best:.3f} ms  {4*n/best/1e9:.1f} GB/s  fallbacks {codec.last_fallbacks()}  exact {bool(torch.equal(out, exp))}", flush=True)
