#!/usr/bin/env python3
"""Developer tool (GPU): per-kernel times of one 64 MiB compress call (timing mode) for a few synthetic volumes, to see what
k_histogram and k_emit wait for in small batches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from datacompressionfloat_amd import MrcZipCodec
dev = torch.device("cuda", 0)
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 16) * 1048576 + 256   # argv[1]: MiFloats (16 = 64 MiB, 256 = 1 GiB)
g = torch.Generator(device=dev).manual_seed(7)
pois = torch.poisson(torch.full((n,), 8.0, device=dev), generator=g)
vols = {
    "poisson": pois.view(torch.int32),
    "poisson + random low 16 bits": (pois.view(torch.int32) | torch.randint(0, 65536, (n,), device=dev, dtype=torch.int32, generator=g)),
    "poisson >> 16 replicated in the low half": (pois.view(torch.int32) | ((pois.view(torch.int32) >> 16) & 0xffff)),
    "gauss": torch.empty(n, dtype=torch.float32, device=dev).normal_(10.0, 3.0, generator=g).view(torch.int32),
    "zeros": torch.zeros(n, dtype=torch.int32, device=dev),
    "random": torch.randint(-2**31, 2**31 - 1, (n,), device=dev, dtype=torch.int32, generator=g),
    "random & 0x3f3f3f3f (64 values per plane)": torch.randint(-2**31, 2**31 - 1, (n,), device=dev, dtype=torch.int32, generator=g) & 0x3f3f3f3f,
    "random & 0x03030303 (4 values per plane)": torch.randint(-2**31, 2**31 - 1, (n,), device=dev, dtype=torch.int32, generator=g) & 0x03030303,
}
c = MrcZipCodec(0, max_batch_chunks=43)
for name, w in vols.items():
    w = w.contiguous()
    for bits in (0,):
        c.compress_device(w, bits, 0)
        c.set_timing(True)
        c.compress_device(w, bits, 0)
        t = c.last_timings()
        c.set_timing(False)
        print(f"{name:45s} " + " ".join(f"{k[2:]}={1e3*v:.0f}" for k, v in t.items() if v > 0.02), flush=True)
