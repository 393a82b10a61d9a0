"""Shared helpers of the test-suite: input generators, oracle / emulator bindings."""
import ctypes
import hashlib
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
SIM_DIR = os.path.join(ROOT, "tests", "sim")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
GOLDEN = os.path.join(ROOT, "tests", "golden")
CHUNK = 6 * 1048576


def sha256(b) -> str:
    return hashlib.sha256(bytes(b)).hexdigest()


def kat_words(n: int) -> np.ndarray:
    """SURVEY App. D integer generator (two LCG steps per word, 4096-word stripes of 10.0f)."""
    A = np.uint32(1664525)
    C = np.uint32(1013904223)
    with np.errstate(over="ignore"):
        a = np.cumprod(np.full(2 * n, A, dtype=np.uint32), dtype=np.uint32)
        g = np.empty(2 * n, dtype=np.uint32)
        g[0] = 1
        g[1:] = np.cumsum(a[:-1], dtype=np.uint32) + np.uint32(1)
        s = a * np.uint32(0x9E3779B9) + C * g
    r, r2 = s[0::2], s[1::2]
    w = (((r2 >> 27) & 1) << 31) | ((np.uint32(124) + ((r2 >> 28) & 7)) << 23) | (r >> 9)
    i = np.arange(n, dtype=np.uint32)
    return np.where(((i >> 12) & 3) == 3, np.uint32(0x41200000), w).astype(np.uint32)


def gauss_words(n: int, seed: int = 1234, header: bool = True) -> np.ndarray:
    """SURVEY 8(d) config 1/2 shape: 256 header words + N(10, 3^2) float32."""
    rng = np.random.default_rng(seed)
    w = rng.normal(10.0, 3.0, n).astype(np.float32).view(np.uint32).copy()
    if header and n >= 256:
        w[:256] = 0
        w[0], w[1], w[2], w[3] = 4096, 4096, 1, 2
    return w


def poisson_words(n: int, seed: int = 7) -> np.ndarray:
    """SURVEY 8(d) config 3 shape: MRC-like header + Poisson(8) detector counts as float32."""
    rng = np.random.default_rng(seed)
    w = rng.poisson(8.0, n).astype(np.float32).view(np.uint32).copy()
    if n >= 256:
        w[:256] = 0
        w[0], w[1], w[2], w[3] = 1024, 1024, 16, 2
    return w


def int_mode_words(n: int, seed: int = 21) -> np.ndarray:
    """Inputs for the "-s int" mode (workers.c:125-175): 256 header words, then floats that exercise the quantiser
    (char)round(x): detector-like counts, exact .5 ties of both signs, values beyond [-128, 127] (the low byte of the
    int32 conversion survives), values beyond int32, infinities, NaNs, denormals, and the int32 edge itself."""
    rng = np.random.default_rng(seed)
    f = rng.normal(20.0, 60.0, n).astype(np.float32)
    f[::7] = np.round(f[::7]) + np.float32(0.5)          # ties: half away from zero
    f[3::11] = -(np.abs(np.round(f[3::11])) + np.float32(0.5))
    special = np.array([0.5, -0.5, 1.5, -1.5, 2.5, 126.5, 127.49, 127.5, 128.0, -128.5, -129.0, 255.0, 256.0, 300.7, -300.7,
                        65535.6, 2147483520.0, 2147483648.0, -2147483648.0, -2147483904.0, 4294967296.0, 1e10, -1e10, 1e38,
                        np.inf, -np.inf, np.nan, 1e-40, -1e-40, 0.0, -0.0, 0.49999997, -0.49999997, 8388607.5, 8388608.0], np.float32)
    k = min(n, len(special) * 8)
    if k:
        f[-k:] = np.resize(special, k)
    w = f.view(np.uint32).copy()
    if n >= 256:
        w[:256] = kat_words(256)                           # arbitrary header bits: copied verbatim, never quantised
    return w


def int_mode_expected(words: np.ndarray) -> np.ndarray:
    """What unzip(-s int) of zip(-s int) returns (workers.c:444-511): header words verbatim, every other word
    (float)(signed char)(char)round(x) with the x86-64 double -> int32 -> low byte conversion."""
    f = words.view(np.float32).astype(np.float64)
    with np.errstate(invalid="ignore", over="ignore"):
        r = np.where(f >= 0, np.floor(f + 0.5), np.ceil(f - 0.5))       # round(): half away from zero (exact in double)
        ok = (r >= -2147483648.0) & (r < 2147483648.0)
        i = np.where(ok, r, -2147483648.0).astype(np.int64)
    q = (i & 0xff).astype(np.uint8).view(np.int8).astype(np.float32)
    out = q.view(np.uint32).copy()
    out[:256] = words[:256]
    return out


def runs_words(n: int, lens, alpha: int, seed: int = 3) -> np.ndarray:
    rng = np.random.default_rng(seed)
    out, tot = [], 0
    while tot < n:
        l = int(rng.choice(lens))
        v = int(rng.integers(0, alpha))
        out.append(np.full(l, (v * 0x01010101) & 0xFFFFFFFF, np.uint32))
        tot += l
    return np.concatenate(out)[:n]


def erase_expected(words: np.ndarray, bits: int) -> np.ndarray:
    """erasebytes semantics (src/tool/erasebytes.c:109-134) in numpy."""
    m = np.uint32(0) if bits >= 32 else np.uint32((0xFFFFFFFF << bits) & 0xFFFFFFFF)
    e = words.copy()
    e[256:] &= m
    return e


def aligned_empty(nbytes: int, align: int = 64) -> np.ndarray:
    raw = np.zeros(nbytes + align, np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + nbytes]


# ------------------------------------------------------------------ oracle
class Oracle:
    def __init__(self, lib):
        self.lib = lib
        vp, u64, u32, i32, i64 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, ctypes.c_int64
        lib.mrcz_oracle_bound.restype = u64
        lib.mrcz_oracle_bound.argtypes = [u64]
        lib.mrcz_oracle_compress.restype = i64
        lib.mrcz_oracle_compress.argtypes = [vp, u64, i32, vp, u64]
        lib.mrcz_oracle_compress_mt.restype = i64
        lib.mrcz_oracle_compress_mt.argtypes = [vp, u64, i32, i32, vp, u64]
        lib.mrcz_oracle_uncompress.restype = i64
        lib.mrcz_oracle_uncompress.argtypes = [vp, u64, vp, u64]
        for f in ("mrcz_oracle_deflate_rle", "mrcz_oracle_deflate_zlib"):
            getattr(lib, f).restype = i64
            getattr(lib, f).argtypes = [vp, u32, vp, u64]
        lib.mrcz_oracle_inflate.restype = i64
        lib.mrcz_oracle_inflate.argtypes = [vp, u64, vp, u64]
        lib.mrcz_oracle_compress_int.restype = i64
        lib.mrcz_oracle_compress_int.argtypes = [vp, u64, vp, u64]
        lib.mrcz_oracle_uncompress_int.restype = i64
        lib.mrcz_oracle_uncompress_int.argtypes = [vp, u64, vp, u64]
        lib.mrcz_oracle_erasebytes.restype = None
        lib.mrcz_oracle_erasebytes.argtypes = [vp, u64, i32]

    def compress(self, data, bits: int, threads: int = 0) -> bytes:
        data = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8)
        cap = int(self.lib.mrcz_oracle_bound(len(data)))
        out = np.empty(cap, np.uint8)
        if threads:
            n = self.lib.mrcz_oracle_compress_mt(data.ctypes.data, len(data), bits, threads, out.ctypes.data, cap)
        else:
            n = self.lib.mrcz_oracle_compress(data.ctypes.data, len(data), bits, out.ctypes.data, cap)
        if n < 0:
            raise RuntimeError("oracle compress failed")
        return out[:n].tobytes()

    def compress_int(self, data) -> bytes:
        """run_compress with -s int (workers.c:782-787)"""
        data = np.frombuffer(bytes(data), dtype=np.uint8)
        cap = int(self.lib.mrcz_oracle_bound(len(data)))
        out = np.empty(cap, np.uint8)
        n = self.lib.mrcz_oracle_compress_int(data.ctypes.data, len(data), out.ctypes.data, cap)
        if n < 0:
            raise RuntimeError("oracle compress_int failed")
        return out[:n].tobytes()

    def uncompress(self, z: bytes, int_mode: bool = False) -> bytes:
        z = np.frombuffer(bytes(z), dtype=np.uint8)
        import struct
        fsz = struct.unpack("<Q", z[:8].tobytes())[0]
        out = np.empty(fsz // 4 * 4, np.uint8)
        f = self.lib.mrcz_oracle_uncompress_int if int_mode else self.lib.mrcz_oracle_uncompress
        n = f(z.ctypes.data, len(z), out.ctypes.data, len(out))
        if n < 0:
            raise RuntimeError("oracle uncompress failed")
        return out[:n].tobytes()

    def deflate(self, plane: np.ndarray, use_zlib: bool = False) -> bytes:
        plane = np.ascontiguousarray(plane, dtype=np.uint8)
        cap = len(plane) + len(plane) // 8 + 4096
        out = np.empty(cap, np.uint8)
        f = self.lib.mrcz_oracle_deflate_zlib if use_zlib else self.lib.mrcz_oracle_deflate_rle
        n = f(plane.ctypes.data, len(plane), out.ctypes.data, cap)
        if n < 0:
            raise RuntimeError("deflate failed")
        return out[:n].tobytes()

    def inflate(self, z: bytes, outlen: int) -> bytes:
        z = np.frombuffer(bytes(z), dtype=np.uint8)
        out = np.empty(outlen, np.uint8)
        n = self.lib.mrcz_oracle_inflate(z.ctypes.data, len(z), out.ctypes.data, outlen)
        return out[:max(n, 0)].tobytes()


def load_oracle() -> Oracle:
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    src = os.path.join(ORACLE_DIR, "mrcz_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return Oracle(ctypes.CDLL(so))


def ref_binary(name: str):
    """oracle/_ref/<name> (the reference's own sources compiled in place) or None."""
    p = os.path.join(REF_DIR, name)
    if os.path.exists(p):
        return p
    if os.path.isdir("/root/reference/src/core"):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)
        return p if os.path.exists(p) else None
    return None


# ------------------------------------------------------------------ SIMT emulator build of the product sources
class SimCodec:
    """Same C ABI as libmrcz_hip.so, kernels executed by tests/sim on the CPU ("device" = host memory)."""

    def __init__(self, lib, max_batch_chunks: int = 2):
        self.lib = lib
        vp, u64, u32, i32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
        lib.mrcz_create.argtypes = [ctypes.POINTER(vp), i32, u32]
        lib.mrcz_destroy.argtypes = [vp]
        lib.mrcz_records_bound.restype = u64
        lib.mrcz_records_bound.argtypes = [u64]
        lib.mrcz_compress_chunks.argtypes = [vp, vp, u64, u64, i32, vp, u64, ctypes.POINTER(u64), vp]
        lib.mrcz_uncompress_chunks.argtypes = [vp, vp, u64, u64, u32, vp, ctypes.POINTER(u64)]
        lib.mrcz_compress_chunks_int8.argtypes = [vp, vp, u64, u64, vp, u64, ctypes.POINTER(u64), vp]
        lib.mrcz_uncompress_chunks_int8.argtypes = [vp, vp, u64, u64, u32, u64, vp, ctypes.POINTER(u64)]
        lib.mrcz_generate_kat_words.argtypes = [vp, vp, u64, u64]
        lib.mrcz_set_ztypes.argtypes = [vp, ctypes.c_char_p]
        lib.mrcz_last_error.restype = ctypes.c_char_p
        lib.mrcz_last_error.argtypes = [vp]
        lib.mrcz_debug_fallbacks.restype = ctypes.c_int64
        lib.mrcz_debug_fallbacks.argtypes = [vp]
        lib.mrcz_debug_chain_fallbacks.restype = ctypes.c_int64
        lib.mrcz_debug_chain_fallbacks.argtypes = [vp]
        self.ctx = vp()
        assert lib.mrcz_create(ctypes.byref(self.ctx), 0, max_batch_chunks) == 0

    def compress_records(self, words: np.ndarray, bits: int, first_chunk: int = 0, int_mode: bool = False) -> bytes:
        n = len(words)
        din = aligned_empty(4 * n).view(np.uint32)
        din[:] = words
        cap = int(self.lib.mrcz_records_bound(n))
        dout = aligned_empty(cap + 8)
        olen = ctypes.c_uint64()
        if int_mode:
            rc = self.lib.mrcz_compress_chunks_int8(self.ctx, din.ctypes.data, n, first_chunk, dout.ctypes.data, cap, ctypes.byref(olen), None)
        else:
            rc = self.lib.mrcz_compress_chunks(self.ctx, din.ctypes.data, n, first_chunk, bits, dout.ctypes.data, cap, ctypes.byref(olen), None)
        if rc != 0:
            raise RuntimeError(f"sim compress rc={rc}: {self.lib.mrcz_last_error(self.ctx)}")
        return dout[:olen.value].tobytes()

    def set_ztypes(self, ztypes) -> int:
        return self.lib.mrcz_set_ztypes(self.ctx, bytes(bytearray(z & 0xff for z in ztypes)))

    def generate_kat(self, first: int, n: int) -> np.ndarray:
        out = aligned_empty(4 * n).view(np.uint32)
        assert self.lib.mrcz_generate_kat_words(self.ctx, out.ctypes.data, first, n) == 0
        return out.copy()

    def uncompress_records(self, rec: bytes, nfloats: int, chk: int = CHUNK, int_mode: bool = False) -> np.ndarray:
        r = aligned_empty(len(rec) + 8)
        r[:len(rec)] = np.frombuffer(rec, np.uint8)
        out = aligned_empty(4 * nfloats).view(np.uint32)
        cons = ctypes.c_uint64()
        if int_mode:
            rc = self.lib.mrcz_uncompress_chunks_int8(self.ctx, r.ctypes.data, len(rec), nfloats, chk, 0, out.ctypes.data, ctypes.byref(cons))
        else:
            rc = self.lib.mrcz_uncompress_chunks(self.ctx, r.ctypes.data, len(rec), nfloats, chk, out.ctypes.data, ctypes.byref(cons))
        if rc != 0:
            raise RuntimeError(f"sim uncompress rc={rc}: {self.lib.mrcz_last_error(self.ctx)}")
        self.fallbacks = int(self.lib.mrcz_debug_fallbacks(self.ctx))
        self.chain_fallbacks = int(self.lib.mrcz_debug_chain_fallbacks(self.ctx))
        return out.copy()


def load_sim() -> SimCodec:
    so = os.path.join(SIM_DIR, "libmrcz_sim.so")
    subprocess.check_call(["make", "-C", SIM_DIR], stdout=subprocess.DEVNULL)
    return SimCodec(ctypes.CDLL(so))
