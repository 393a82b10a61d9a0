"""MI355X-native float32 mask + byte-plane + DEFLATE(Z_RLE) codec (drop-in for the
apply_mask + zip/unzip hot path of ruanhuabin/DataCompressionFloat).

The compute path is the hand-written HIP library ``lib/libmrcz_hip.so`` (csrc/, C ABI in
include/mrcz_hip.h).  This package is the thin Python host mirror used by tests and bench.py:
PyTorch only supplies device memory, streams and torch.distributed.  There is no CPU fallback:
importing :mod:`datacompressionfloat_amd.codec` raises if the HIP library is missing.
"""
from .codec import (  # noqa: F401
    CHUNK_FLOATS,
    FILE_HEADER_BYTES,
    MrcZipCodec,
    MrczError,
    pack_file_header,
    unpack_file_header,
)
