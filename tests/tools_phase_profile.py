"""Developer tool (GPU): in-kernel phase breakdown of the parallel inflate for one heavy stream."""
import ctypes, sys, os
import sys as _s
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from datacompressionfloat_amd import MrcZipCodec
from datacompressionfloat_amd.codec import _LIB
n = 16 * 6291456
g = torch.Generator(device="cuda").manual_seed(1234)
x = torch.empty(n, dtype=torch.float32, device="cuda").normal_(10.0, 3.0, generator=g).view(torch.int32)
c = MrcZipCodec(0, 16)
BITS = int(_s.argv[2]) if len(_s.argv) > 2 else 8
rec, _ = c.compress_device(x, BITS, 1)
_LIB.mrcz_debug_inflate_phases.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p]
MODE = int(_s.argv[1]) if len(_s.argv) > 1 else 1   # 1 = sequential-chain kernel, 2 = block-parallel kernels (count rows, then write rows)
_LIB.mrcz_debug_inflate_phases(c._ctx, MODE, 0, None)
out, _ = c.uncompress_device(rec, n)
names = ["hdr", "stage", "P1 exitfn", "P2 compose", "P3 walk", "P3 scans", "P4 scatter", "P4 wait", "fill+flush", "-"]
for s in ((4, 5, 6, 7) if MODE == 1 else (5, 6, 7)):
    buf = (ctypes.c_uint64 * 20)()
    _LIB.mrcz_debug_inflate_phases(c._ctx, MODE, s, buf)
    v = list(buf)
    tot = sum(v[:10]) + sum(v[12:20])
    print(f"stream {s} (plane {s % 4}): blocks={v[10]} windows={v[11]} total={tot / 1e6:.1f} M shader clocks (thread 0 of every block, summed) " +
          " ".join(f"{names[i]}={100.0 * v[i] / max(tot, 1):.1f}%" for i in range(9)) + " | hdr: " + " ".join(f"{n}={100.0 * v[12 + i] / max(tot, 1):.1f}%" for i, n in enumerate(["first", "blcode", "lens-write+sync", "lit", "dist", "lens-exitfn", "lens-compose", "lens-count"])) + f" rest={100.0 * v[0] / max(tot, 1):.1f}%")

cnt = (ctypes.c_uint64 * 2)()
_LIB.mrcz_debug_candidates.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
_LIB.mrcz_debug_candidates(c._ctx, cnt)
print(f"candidates: {cnt[0]} passed the signature scan, {cnt[1]} validated ({n // 6291456} chunks)")
