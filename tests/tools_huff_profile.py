"""Developer tool (GPU): phase clocks of k_huffman (one thread = one tree; the kernel lasts as long as its slowest tree)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from datacompressionfloat_amd import MrcZipCodec
from datacompressionfloat_amd.codec import _LIB
n = 16 * 6291456
g = torch.Generator(device="cuda").manual_seed(1234)
x = torch.empty(n, dtype=torch.float32, device="cuda").normal_(10.0, 3.0, generator=g).view(torch.int32)
c = MrcZipCodec(0, 16)
_LIB.mrcz_debug_inflate_phases.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p]
c.compress_device(x, 8, 1)
_LIB.mrcz_debug_inflate_phases(c._ctx, 3, 0, None)
c.compress_device(x, 8, 1)
buf = (ctypes.c_uint64 * 20)()
v = []
for s in (0, 1):
    _LIB.mrcz_debug_inflate_phases(c._ctx, 3, s, buf)
    v += list(buf)
names = ["load freqs + heap fill", "merge loop (lit/len)", "gen_bitlen", "overflow repair", "gen_codes + code rows", "scan_tree", "bit-length tree", "header bits + meta"]
trees = max(v[32], 1)
print(f"trees {v[32]}  (shader clocks per tree, thousands)")
for i, nm in enumerate(names):
    print(f"  {nm:28s} max {v[i] / 1000.0:8.1f} k   mean {v[16 + i] / trees / 1000.0:8.1f} k")
print(f"  sum of max {sum(v[:8]) / 1000.0:.1f} k, sum of mean {sum(v[16:24]) / trees / 1000.0:.1f} k")
