"""ctypes binding of the C ABI declared in include/mrcz_hip.h (no torch types cross it)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MRCZ_LIB_PATH: developer knob (tools/ab_*.py time several builds of the kernels in one GPU call); it still is a HIP build
LIB_PATH = os.environ.get("MRCZ_LIB_PATH") or os.path.join(_HERE, "lib", "libmrcz_hip.so")


class MrczLibraryMissing(ImportError):
    pass


def load():
    if not os.path.exists(LIB_PATH):
        raise MrczLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the codec."
        )
    lib = ctypes.CDLL(LIB_PATH)
    vp, u64, u32, i32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
    lib.mrcz_create.restype = i32
    lib.mrcz_create.argtypes = [ctypes.POINTER(vp), i32, u32]
    lib.mrcz_destroy.restype = None
    lib.mrcz_destroy.argtypes = [vp]
    lib.mrcz_last_error.restype = ctypes.c_char_p
    lib.mrcz_last_error.argtypes = [vp]
    lib.mrcz_stream.restype = vp
    lib.mrcz_stream.argtypes = [vp]
    lib.mrcz_records_bound.restype = u64
    lib.mrcz_records_bound.argtypes = [u64]
    lib.mrcz_compress_chunks.restype = i32
    lib.mrcz_compress_chunks.argtypes = [vp, vp, u64, u64, i32, vp, u64, ctypes.POINTER(u64), ctypes.POINTER(u64)]
    lib.mrcz_uncompress_chunks.restype = i32
    lib.mrcz_uncompress_chunks.argtypes = [vp, vp, u64, u64, u32, vp, ctypes.POINTER(u64)]
    lib.mrcz_compress_chunks_int8.restype = i32
    lib.mrcz_compress_chunks_int8.argtypes = [vp, vp, u64, u64, vp, u64, ctypes.POINTER(u64), ctypes.POINTER(u64)]
    lib.mrcz_uncompress_chunks_int8.restype = i32
    lib.mrcz_uncompress_chunks_int8.argtypes = [vp, vp, u64, u64, u32, u64, vp, ctypes.POINTER(u64)]
    lib.mrcz_generate_kat_words.restype = i32
    lib.mrcz_generate_kat_words.argtypes = [vp, vp, u64, u64]
    lib.mrcz_set_ztypes.restype = i32
    lib.mrcz_set_ztypes.argtypes = [vp, ctypes.c_char_p]
    lib.mrcz_erase_bits.restype = i32
    lib.mrcz_erase_bits.argtypes = [vp, vp, u64, u64, i32]
    lib.mrcz_set_timing.restype = i32
    lib.mrcz_set_timing.argtypes = [vp, i32]
    lib.mrcz_last_timings.restype = i32
    lib.mrcz_last_timings.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_float), i32]
    lib.mrcz_debug_fallbacks.restype = ctypes.c_int64
    lib.mrcz_debug_fallbacks.argtypes = [vp]
    lib.mrcz_debug_chain_fallbacks.restype = ctypes.c_int64
    lib.mrcz_debug_chain_fallbacks.argtypes = [vp]
    return lib


# every symbol include/mrcz_hip.h declares (checked by tests/test_abi.py without a GPU)
EXPORTS = [
    "mrcz_create", "mrcz_destroy", "mrcz_last_error", "mrcz_stream", "mrcz_records_bound",
    "mrcz_compress_chunks", "mrcz_uncompress_chunks", "mrcz_erase_bits", "mrcz_set_timing",
    "mrcz_last_timings", "mrcz_debug_blocks", "mrcz_debug_fallbacks", "mrcz_debug_chain_fallbacks", "mrcz_debug_inflate_phases", "mrcz_debug_candidates", "mrcz_device_count", "mrcz_dev_malloc", "mrcz_dev_free",
    "mrcz_host_malloc", "mrcz_host_free", "mrcz_copy_h2d", "mrcz_copy_d2h",
    "mrcz_event_create", "mrcz_event_destroy", "mrcz_event_record", "mrcz_stream_wait_event", "mrcz_event_sync",
    "mrcz_copy_h2d_async", "mrcz_copy_d2h_async", "mrcz_compress_chunks_async", "mrcz_uncompress_chunks_async",
    "mrcz_set_ztypes", "mrcz_generate_kat_words", "mrcz_err_hist", "mrcz_err_collect", "mrcz_compress_chunks_int8", "mrcz_uncompress_chunks_int8", "mrcz_compress_chunks_int8_async", "mrcz_uncompress_chunks_int8_async",
]
