"""Developer check (GPU): the two Huffman paths (per-thread headers / k_huffman_hdr) must write identical containers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from datacompressionfloat_amd import MrcZipCodec
dev = torch.device("cuda", 0)
os.environ["MRCZ_HUFF_SPLIT"] = "0"; mono = MrcZipCodec(0, max_batch_chunks=8)
os.environ["MRCZ_HUFF_SPLIT"] = "1"; split = MrcZipCodec(0, max_batch_chunks=8)
rng = np.random.default_rng(2026)
bad = 0
for case in range(60):
    n = int(rng.integers(1, 3_000_000))
    kind = case % 6
    if kind == 0: w = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    elif kind == 1: w = rng.normal(10, 3, n).astype(np.float32).view(np.uint32)
    elif kind == 2: w = rng.poisson(8.0, n).astype(np.float32).view(np.uint32)
    elif kind == 3: w = (rng.integers(0, 4, n, dtype=np.uint64) ** 3).astype(np.uint32) * np.uint32(0x01010101)
    elif kind == 4: w = np.repeat(rng.integers(0, 2**32, n // 97 + 1, dtype=np.uint64).astype(np.uint32), 97)[:n]
    else: w = rng.geometric(0.02, n).astype(np.uint32) | (rng.integers(0, 3, n, dtype=np.uint64).astype(np.uint32) << 16)
    bits = int(rng.choice([0, 4, 8, 12, 16, 20, 23, 28, 32]))
    t = torch.from_numpy(w.view(np.int32)).to(dev)
    a, _ = mono.compress_device(t, bits, 0); a = a.cpu().numpy().tobytes()
    b, _ = split.compress_device(t, bits, 0); b = b.cpu().numpy().tobytes()
    if a != b:
        bad += 1
        print("MISMATCH case", case, "kind", kind, "n", n, "bits", bits, len(a), len(b), flush=True)
print("cases 60, mismatches", bad)
