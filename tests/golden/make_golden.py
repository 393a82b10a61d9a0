#!/usr/bin/env python3
"""Regenerates tests/golden/*.  Run in the build container (needs /root/reference):

    make -C oracle ref && python tests/golden/make_golden.py

Inputs are produced by the integer generators in tests/util.py (nothing random is committed); the
outputs are what the REFERENCE ITSELF (oracle/_ref/mrc_tar_c = the reference's own sources compiled
in place against the image's zlib) writes for them.  Small containers are committed verbatim as
*.zip fixtures, large ones as sha256 + size in golden.json.  The App. D hashes recorded in SURVEY.md
(produced with the reference binary + its bundled zlib 1.2.8) are kept in golden.json under
"survey_appendix_d" and must agree with what this script regenerates for the same inputs.
"""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import util  # noqa: E402

import numpy as np  # noqa: E402

REF = util.ref_binary("mrc_tar_c")
assert REF, "oracle/_ref/mrc_tar_c missing: run `make -C oracle ref` in the build container"

SURVEY_D = {  # SURVEY.md Appendix D (reference + zlib 1.2.8): name -> {bits: (size, sha256 or 16-hex prefix)}
    "katA": {0: (769630, "c69720170255cb16341c414bc6ebb557bf65f708bc889b7f00b7e73ea8cec4b5"),
             8: (543250, "467f7d93d75ab67da5598259caf891dcb12fc632868a40137ab88598a0bdd11f"),
             12: (431884, "bf970b9a88a932f3"),
             16: (316879, "3cf830a3e2a1366c6034726e18dee40863543191de39ab0efce32963d70a5baf"),
             23: (131548, "faef79e4e7f94a12"),
             24: (90508, "52d1441020e5e40a5ab925a798a6f94df52df32c97ab537fb56875210463a868"),
             31: (43436, "cf20e3021fe454b1"),
             32: (2401, "1a2efd37f9c45df82f457a7e0b8af0b9e53be02c534b6a7e3b5f9f3cacb59b62")},
    "katB": {0: (16091267, "6b929c0d08f7f9b9a8d5682cc0220c41888aed8a36cecf560b25bade2eb6b501"),
             8: (11351150, "034c55c12b03e5384b7881ca60c52be7bf00d66819002e20b85ec8889527944a"),
             12: (9016521, "58915f1e8acb9f0c"),
             16: (6611166, "607fe6d887f3dec0b8c846d07bfbf4b1042ef70395890cb7646f48c0a5eda925"),
             23: (2725768, "54fcd2b5bad5000f"),
             24: (1871169, "b07aead5f1a1515c7a16b5e324fdb5e57e9d526f584eca2659b6cb5b6e092650"),
             31: (880449, "0ff647b9604410c3"),
             32: (25774, "25a6df30b1181913869a14ff3f5089ecae488e6573d52be46692d506a034556e")},
}


def ref_zip(data: bytes, bits: int, mode: str = "float") -> bytes:
    with tempfile.TemporaryDirectory() as d:
        i, o = os.path.join(d, "in"), os.path.join(d, "out.zip")
        open(i, "wb").write(data)
        subprocess.check_call([REF, "-i", i, "-o", o, "-b", str(bits), "-t", "zip", "-s", mode], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return open(o, "rb").read()


def ref_unzip(z: bytes, mode: str = "float") -> bytes:
    with tempfile.TemporaryDirectory() as d:
        i, o = os.path.join(d, "in.zip"), os.path.join(d, "out")
        open(i, "wb").write(z)
        subprocess.check_call([REF, "-i", i, "-o", o, "-t", "unzip", "-s", mode], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return open(o, "rb").read()


def lz4_container(data: bytes, hc: bool) -> bytes:
    """A container whose four byte streams are LZ4 (ztype 2) or LZ4HC (ztype 4) blocks, built the way run_compress would
    build it with LZ4_DEF / LZ4HC_DEF selected (zip.c:32-67 _lz4_def: COMPRESSED iff the compressor returned > 0, i.e. the
    block fits in inlen bytes) -- the reference's writer never selects them (workers.c:719), its reader accepts them
    (workers.c:584, zip.c:306-318).  The blocks come from the reference's own vendored LZ4 (src/core/lz4.c, lz4hc.c), compiled
    into oracle/_ref/libmrcref.so."""
    import ctypes, struct
    lib = ctypes.CDLL(os.path.join(util.REF_DIR, "libmrcref.so"))
    fn = lib.LZ4_compressHC_limitedOutput if hc else lib.LZ4_compress_limitedOutput
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    n = len(data) // 4
    w = np.frombuffer(data[: 4 * n], np.uint32)
    b = w.view(np.uint8).reshape(-1, 4)
    zt = 4 if hc else 2
    out = bytearray(struct.pack("<QIb4b", len(data), util.CHUNK, 0, zt, zt, zt, zt))
    for c0 in range(0, n, util.CHUNK):
        c1 = min(n, c0 + util.CHUNK)
        hdr, pay = bytearray(), bytearray()
        for j in range(4):
            plane = np.ascontiguousarray(b[c0:c1, j]).tobytes()
            buf = ctypes.create_string_buffer(len(plane) + 64)
            zl = fn(plane, buf, len(plane), len(plane))
            if zl > 0:
                hdr += struct.pack("<I", zl)
                pay += buf.raw[:zl]
            else:
                hdr += struct.pack("<I", len(plane) | 0x80000000)
                pay += plane
        out += hdr + pay
    return bytes(out)


def lz4_cases():
    """name -> (input bytes, hc): runs (long matches, overlapping copies), detector counts (short literals and matches), noise (RAW)"""
    return {"lz4_runs": (util.runs_words(30000, [1, 2, 3, 5, 17, 300, 1000, 5000], 3, seed=31).tobytes(), False),
            "lz4_poisson": (util.poisson_words(50000, seed=32).tobytes(), False),
            "lz4hc_poisson": (util.poisson_words(50000, seed=33).tobytes(), True),
            "lz4_gauss_tail1": (util.gauss_words(20001, seed=34).tobytes() + b"\x09", False)}


def int_cases():
    """name -> input bytes for the "-s int" mode (workers.c:125-175,444-511): committed with the reference's containers"""
    return {"int5000": util.int_mode_words(5000, seed=21).tobytes(),
            "int300": util.int_mode_words(300, seed=22).tobytes(),               # 44 quantised words after the header
            "int70001_tail2": util.int_mode_words(70001, seed=23).tobytes() + b"\x07\x08"}


def small_cases():
    """name -> (input bytes, bits): committed verbatim with their reference containers."""
    cases = {}
    cases["words100_b0"] = (util.kat_words(100).tobytes(), 0)                 # 400-byte input (SURVEY App. A example)
    cases["words5000_tail3_b8"] = (util.kat_words(5000).tobytes() + b"\x01\x02\x03", 8)  # 20,003 bytes: trailing bytes dropped
    cases["words300_b16"] = (util.kat_words(300).tobytes(), 16)               # 44 masked words after the 256-word header
    cases["words256_b32"] = (util.kat_words(256).tobytes(), 32)               # header only: nothing is masked
    cases["runs9000_b0"] = (util.runs_words(9000, [1, 2, 3, 4, 257, 258, 259, 260, 261, 516, 517, 518, 519], 3, seed=11).tobytes(), 0)
    cases["gauss20000_b12"] = (util.gauss_words(20000, seed=5).tobytes(), 12)
    cases["poisson40000_b0"] = (util.poisson_words(40000, seed=9).tobytes(), 0)
    return cases


def main():
    golden = {"survey_appendix_d": {k: {str(b): list(v) for b, v in d.items()} for k, d in SURVEY_D.items()},
              "kat_inputs": {"katA": 300000, "katB": 6303801}, "regenerated": {}, "small": {}, "large": {}}
    for name, n in golden["kat_inputs"].items():
        data = util.kat_words(n).tobytes()
        for b, (size, h) in SURVEY_D[name].items():
            z = ref_zip(data, b)
            hh = util.sha256(z)
            assert len(z) == size and hh.startswith(h), (name, b, len(z), hh)
            dec = ref_unzip(z)
            golden["regenerated"][f"{name}_b{b}"] = {"size": len(z), "sha256": hh, "decoded_sha256": util.sha256(dec)}
        print(name, "matches SURVEY App. D")
    for name, (data, b) in small_cases().items():
        z = ref_zip(data, b)
        open(os.path.join(HERE, name + ".zip"), "wb").write(z)
        golden["small"][name] = {"bits": b, "input_bytes": len(data), "input_sha256": util.sha256(data), "size": len(z), "sha256": util.sha256(z)}
    # "-s int" mode: containers and decoded outputs of the reference run with -s int
    golden["int_mode"] = {}
    for name, data in int_cases().items():
        z = ref_zip(data, 5, "int")                      # the mask level is ignored in this mode (workers.c:786)
        assert z == ref_zip(data, 0, "int")
        open(os.path.join(HERE, name + ".zip"), "wb").write(z)
        dec = ref_unzip(z, "int")
        n = len(data) // 4
        exp = util.int_mode_expected(np.frombuffer(data[: 4 * n], np.uint32)).tobytes()
        assert dec == exp, name                          # the numpy statement of the quantiser agrees with the reference binary
        golden["int_mode"][name] = {"input_bytes": len(data), "input_sha256": util.sha256(data), "size": len(z), "sha256": util.sha256(z),
                                    "decoded_sha256": util.sha256(dec)}
    # decoder tolerance: LZ4 / LZ4HC byte streams (ztypes 2 / 4); the reference binary must decode them to the input
    golden["lz4"] = {}
    for name, (data, hc) in lz4_cases().items():
        z = lz4_container(data, hc)
        assert ref_unzip(z) == data[: len(data) // 4 * 4], name
        open(os.path.join(HERE, name + ".zip"), "wb").write(z)
        golden["lz4"][name] = {"input_bytes": len(data), "input_sha256": util.sha256(data), "size": len(z), "sha256": util.sha256(z), "hc": hc}
    # larger seeded shapes (scaled to a few chunks): hashes only
    big = {
        "gauss_4Mi_b8": (util.gauss_words(4 * 1048576, seed=1234).tobytes(), 8),         # one partial chunk
        "gauss_16Mi_b8": (util.gauss_words(16 * 1048576, seed=1234).tobytes(), 8),       # config 1: 64 MiB, 3 chunks
        "poisson_7Mi_b0": (util.poisson_words(7 * 1048576, seed=7).tobytes(), 0),        # config 3 shape, 2 chunks
        # config 3 at its full size (mrc_small_full.sh shape): 1024-byte header + 1024 x 1024 x 16 detector counts, b = 0 and 8
        "poisson_mrc_small_b0": (util.poisson_words(256 + 16 * 1048576, seed=7).tobytes(), 0),
        "poisson_mrc_small_b8": (util.poisson_words(256 + 16 * 1048576, seed=7).tobytes(), 8),
        "int_mode_7Mi": (util.int_mode_words(7 * 1048576, seed=24).tobytes(), -1),       # -s int across a chunk boundary
    }
    for name, (data, b) in big.items():
        z = ref_zip(data, b) if b >= 0 else ref_zip(data, 0, "int")
        golden["large"][name] = {"bits": b, "input_bytes": len(data), "input_sha256": util.sha256(data), "size": len(z), "sha256": util.sha256(z)}
        print(name, len(z))
    json.dump(golden, open(os.path.join(HERE, "golden.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
