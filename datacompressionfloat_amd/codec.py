"""Host mirror of the reference's chunk-codec seam for Python callers.

Mirrors /root/reference/src/include/workers.h:30-31 (run_compress / run_uncompress) and the file
header I/O of src/core/common.c:117-148 with the same argument meaning: `bits` is bitsToMask
(0..32), containers start with the 17-byte header, decode output is 4*floor(fsz/4) bytes.
Device memory comes from torch; all arithmetic happens in libmrcz_hip.so.
"""
import ctypes
import struct

import torch

from . import _lib

CHUNK_FLOATS = 6 * 1048576  # src/include/constant.h:25
FILE_HEADER_BYTES = 17      # src/core/common.c:137-148

_LIB = _lib.load()  # raises MrczLibraryMissing: no CPU fallback


class MrczError(RuntimeError):
    pass


def pack_file_header(fsz: int) -> bytes:
    """write_mrczip_header (src/core/common.c:137-148): u64 fsz, u32 chk, i8 type, i8 ztypes[4]."""
    return struct.pack("<QIb4b", fsz, CHUNK_FLOATS, 0, 0, 0, 0, 0)


def unpack_file_header(buf: bytes):
    """read_mrczip_header (src/core/common.c:117-134) -> (fsz, chk, type, ztypes)."""
    if len(buf) < FILE_HEADER_BYTES:
        raise MrczError("container shorter than the 17-byte header")
    fsz, chk, typ, z0, z1, z2, z3 = struct.unpack("<QIb4b", bytes(buf[:FILE_HEADER_BYTES]))
    return fsz, chk, typ, (z0, z1, z2, z3)


class MrcZipCodec:
    """One codec context (HIP stream + workspace) on one GPU."""

    def __init__(self, device=0, max_batch_chunks=64):
        if not torch.cuda.is_available():
            raise MrczError("no HIP device visible: the codec has no CPU path")
        self.device = torch.device("cuda", device if isinstance(device, int) else device.index)
        self._ctx = ctypes.c_void_p()
        rc = _LIB.mrcz_create(ctypes.byref(self._ctx), self.device.index, max_batch_chunks)
        if rc != 0:
            raise MrczError(f"mrcz_create failed ({rc})")

    def close(self):
        if self._ctx:
            _LIB.mrcz_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self, what, rc):
        msg = _LIB.mrcz_last_error(self._ctx)
        return MrczError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def set_timing(self, on: bool):
        _LIB.mrcz_set_timing(self._ctx, 1 if on else 0)

    def last_timings(self):
        names = (ctypes.c_char_p * 32)()
        ms = (ctypes.c_float * 32)()
        n = _LIB.mrcz_last_timings(self._ctx, names, ms, 32)
        return {names[i].decode(): float(ms[i]) for i in range(n)}

    def last_fallbacks(self) -> int:
        """streams of the last uncompress call that needed the sequential general-distance decoder"""
        return int(_LIB.mrcz_debug_fallbacks(self._ctx))

    @staticmethod
    def records_bound(nfloats: int) -> int:
        return int(_LIB.mrcz_records_bound(nfloats))

    # ---- device-resident API (what bench.py times) ----
    def compress_device(self, words: torch.Tensor, bits: int, first_chunk: int = 0, out: torch.Tensor = None, int_mode: bool = False):
        """words: cuda tensor of 32-bit elements (int32/float32/uint32 view), chunk-aligned start.
        int_mode = the reference's "-s int" (src/core/workers.c:125-175): bits is ignored.
        Returns (records uint8 cuda tensor view, plane_bytes[4])."""
        assert words.is_cuda and words.is_contiguous() and words.element_size() == 4
        n = words.numel()
        cap = self.records_bound(n)
        if out is None:
            out = torch.empty(cap, dtype=torch.uint8, device=words.device)
        assert out.is_cuda and out.numel() >= cap
        torch.cuda.current_stream(words.device).synchronize()
        olen = ctypes.c_uint64()
        planes = (ctypes.c_uint64 * 4)()
        if int_mode:
            rc = _LIB.mrcz_compress_chunks_int8(self._ctx, words.data_ptr(), n, first_chunk, out.data_ptr(), out.numel(),
                                                ctypes.byref(olen), planes)
        else:
            rc = _LIB.mrcz_compress_chunks(self._ctx, words.data_ptr(), n, first_chunk, bits, out.data_ptr(), out.numel(),
                                           ctypes.byref(olen), planes)
        if rc != 0:
            raise self._err("mrcz_compress_chunks", rc)
        return out[: olen.value], [int(p) for p in planes]

    def uncompress_device(self, records: torch.Tensor, nfloats: int, chk: int = CHUNK_FLOATS, out: torch.Tensor = None,
                          int_mode: bool = False, first_chunk: int = 0):
        assert records.is_cuda and records.dtype == torch.uint8 and records.is_contiguous()
        if out is None:
            out = torch.empty(nfloats, dtype=torch.int32, device=records.device)
        assert out.is_cuda and out.numel() >= nfloats and out.element_size() == 4
        torch.cuda.current_stream(records.device).synchronize()
        consumed = ctypes.c_uint64()
        if int_mode:
            rc = _LIB.mrcz_uncompress_chunks_int8(self._ctx, records.data_ptr(), records.numel(), nfloats, chk, first_chunk, out.data_ptr(),
                                                  ctypes.byref(consumed))
        else:
            rc = _LIB.mrcz_uncompress_chunks(self._ctx, records.data_ptr(), records.numel(), nfloats, chk, out.data_ptr(),
                                             ctypes.byref(consumed))
        if rc != 0:
            raise self._err("mrcz_uncompress_chunks", rc)
        return out[:nfloats], int(consumed.value)

    def erase_bits_device(self, words: torch.Tensor, bits: int, first_word_index: int = 0):
        assert words.is_cuda and words.element_size() == 4
        torch.cuda.current_stream(words.device).synchronize()
        rc = _LIB.mrcz_erase_bits(self._ctx, words.data_ptr(), words.numel(), first_word_index, bits)
        if rc != 0:
            raise self._err("mrcz_erase_bits", rc)
        return words

    def generate_kat_device(self, words: torch.Tensor, first_index: int = 0):
        """fill `words` (cuda, 32-bit elements) with words [first_index, ...) of the SURVEY App. D integer generator"""
        assert words.is_cuda and words.is_contiguous() and words.element_size() == 4
        torch.cuda.current_stream(words.device).synchronize()
        rc = _LIB.mrcz_generate_kat_words(self._ctx, words.data_ptr(), first_index, words.numel())
        if rc != 0:
            raise self._err("mrcz_generate_kat_words", rc)
        return words

    # ---- file-image API: same bytes as `mrc_tar_c -t zip|unzip` reads/writes ----
    def zip_bytes(self, data: bytes, bits: int, mode: str = "float") -> bytes:
        """run_compress on an in-memory file image (src/core/workers.c:690-881); mode = dataConvertedType ("float" | "int")."""
        if mode not in ("float", "int"):
            raise MrczError("mode must be 'float' or 'int' (mrc_tar -s)")
        if bits < 0 or bits > 32:
            raise MrczError("bits must be in 0..32 (src/core/workers.c:29-37 has 33 table entries)")
        fsz = len(data)
        nfl = fsz // 4
        if nfl == 0:
            return b""  # src/core/workers.c:757: nothing is written when the first read is empty
        host = torch.frombuffer(bytearray(data[: nfl * 4]), dtype=torch.int32)
        dev = host.to(self.device)
        rec, _ = self.compress_device(dev, bits, 0, int_mode=(mode == "int"))
        return pack_file_header(fsz) + rec.cpu().numpy().tobytes()

    def unzip_bytes(self, container: bytes, mode: str = "float") -> bytes:
        """read_mrczip_header + run_uncompress (src/core/workers.c:568-688); mode as for zip_bytes (the container does not
        record it: the reference needs -s int again on decode)."""
        fsz, chk, typ, ztypes = unpack_file_header(container)
        if any(z not in (0, 2, 4) for z in ztypes):
            raise MrczError("byte stream compressor types must be ZLIB_DEF (0), LZ4_DEF (2) or LZ4HC_DEF (4)")
        rc = _LIB.mrcz_set_ztypes(self._ctx, bytes(bytearray(z & 0xff for z in ztypes)))
        if rc != 0:
            raise self._err("mrcz_set_ztypes", rc)
        nfl = fsz // 4
        if chk == 0:
            raise MrczError("chunk size 0 in header (the reference divides by it, src/core/workers.c:589)")
        if nfl == 0:
            return b""
        rec = torch.frombuffer(bytearray(container[FILE_HEADER_BYTES:]), dtype=torch.uint8).to(self.device)
        out, _ = self.uncompress_device(rec, nfl, chk, int_mode=(mode == "int"))
        return out.cpu().numpy().tobytes()
