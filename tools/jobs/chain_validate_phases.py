#!/usr/bin/env python3
"""Developer tool (GPU): shader-clock stamps inside k_chain (per stream) and k_validate_candidates (first 64 workgroups)
for one decompress call (mrcz_debug_inflate_phases mode 4).  argv: chunks (default 16), mask bits (default 8)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from datacompressionfloat_amd import MrcZipCodec
from datacompressionfloat_amd.codec import _LIB
NCH = int(sys.argv[1]) if len(sys.argv) > 1 else 16
BITS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = NCH * 6291456
g = torch.Generator(device="cuda").manual_seed(1234)
x = torch.empty(n, dtype=torch.float32, device="cuda").normal_(10.0, 3.0, generator=g).view(torch.int32)
c = MrcZipCodec(0, NCH)
rec, _ = c.compress_device(x, BITS, 1)
_LIB.mrcz_debug_inflate_phases.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p]
out, _ = c.uncompress_device(rec, n)  # warm
_LIB.mrcz_debug_inflate_phases(c._ctx, 4, 0, None)
out, _ = c.uncompress_device(rec, n)
ns = 4 * NCH
words = []
for k in range((ns * 8 + 64 * 8 + 19) // 20):
    buf = (ctypes.c_uint64 * 20)()
    _LIB.mrcz_debug_inflate_phases(c._ctx, 4, k, buf)
    words += list(buf)
_LIB.mrcz_debug_inflate_phases(c._ctx, 0, 0, None)
print("k_chain, per stream (shader clocks, thousands): load+hash, walk, segments | blocks, candidates")
rows = []
for s in range(ns):
    v = words[s * 8:s * 8 + 8]
    if v[3] == 0:
        continue
    rows.append((s, (v[1] - v[0]) / 1e3, (v[2] - v[1]) / 1e3, (v[3] - v[2]) / 1e3, v[5], v[6]))
for plane in range(4):
    r = [x for x in rows if x[0] % 4 == plane]
    if r:
        worst = max(r, key=lambda x: x[1] + x[2] + x[3])
        print(f"  plane {plane}: worst stream {worst[0]}: {worst[1]:.1f} {worst[2]:.1f} {worst[3]:.1f} | {worst[4]} {worst[5]}   mean total {sum(x[1]+x[2]+x[3] for x in r)/len(r):.1f}")
print("k_validate_candidates, first 64 workgroups (thousands of clocks): prologue->stage, zero rows, CL table, symbols, write-back, total")
base = ns * 8
tot = []
for b in range(64):
    v = words[base + b * 8: base + b * 8 + 8]
    if v[6] == 0 or v[1] == 0:
        continue
    tot.append(((v[1] - v[0]) / 1e3, (v[2] - v[1]) / 1e3, (v[3] - v[2]) / 1e3, (v[4] - v[3]) / 1e3, (v[5] - v[4]) / 1e3, (v[6] - v[0]) / 1e3))
if tot:
    for i, name in enumerate(["stage", "zero", "cl-table", "symbols", "write-back", "total"]):
        col = [t[i] for t in tot]
        print(f"  {name:10s} mean {sum(col)/len(col):8.1f}  max {max(col):8.1f}")
