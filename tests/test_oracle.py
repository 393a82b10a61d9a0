"""CPU-only: pins the oracle (oracle/mrcz_oracle.c) against
 (1) the SURVEY App. D known-answer hashes produced by the reference binary with zlib 1.2.8,
 (2) containers written by oracle/_ref (the reference's own sources compiled in place) -- committed
     under tests/golden by tests/golden/make_golden.py, and re-run live when oracle/_ref exists,
 (3) system zlib called with the reference's deflateInit2 parameters (zip.c:106-123)."""
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

import util

G = json.load(open(os.path.join(util.GOLDEN, "golden.json")))


@pytest.mark.parametrize("name", ["katA", "katB"])
def test_oracle_matches_survey_appendix_d(oracle, name):
    data = util.kat_words(G["kat_inputs"][name]).tobytes()
    for b, (size, h) in G["survey_appendix_d"][name].items():
        z = oracle.compress(data, int(b))
        assert len(z) == size, (name, b)
        assert util.sha256(z).startswith(h), (name, b)
        dec = oracle.uncompress(z)
        assert util.sha256(dec) == G["regenerated"][f"{name}_b{b}"]["decoded_sha256"]
        exp = util.erase_expected(np.frombuffer(data, np.uint32), int(b)).tobytes()
        assert dec == exp  # the reference's one real test: unzip(zip(x,b)) == erasebytes(x,b) (run_full_test.sh:84-102)


def test_oracle_matches_committed_reference_containers(oracle):
    from golden.make_golden import small_cases
    for name, (data, bits) in small_cases().items():
        meta = G["small"][name]
        assert util.sha256(data) == meta["input_sha256"], "input generator drifted"
        ref = open(os.path.join(util.GOLDEN, name + ".zip"), "rb").read()
        assert util.sha256(ref) == meta["sha256"]
        assert oracle.compress(data, bits) == ref, name
        dec = oracle.uncompress(ref)
        assert len(dec) == len(data) // 4 * 4
        assert dec == util.erase_expected(np.frombuffer(data[: len(dec)], np.uint32), bits).tobytes()


def test_oracle_matches_reference_hashes_of_seeded_volumes(oracle):
    # gauss_4Mi_b8 only (one 16 MiB volume) to keep the CPU suite short; the other two are covered on the GPU box
    meta = G["large"]["gauss_4Mi_b8"]
    data = util.gauss_words(4 * 1048576, seed=1234).tobytes()
    assert util.sha256(data) == meta["input_sha256"], "numpy generator drifted"
    z = oracle.compress(data, meta["bits"])
    assert (len(z), util.sha256(z)) == (meta["size"], meta["sha256"])
    assert oracle.compress(data, meta["bits"], threads=4) == z  # chunk-parallel pthread variant


@pytest.mark.skipif(util.ref_binary("mrc_tar_c") is None, reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_matches_live_reference_binary(oracle):
    ref = util.ref_binary("mrc_tar_c")
    rng = np.random.default_rng(99)
    cases = [(util.gauss_words(70000, seed=3), 8), (util.runs_words(50000, [1, 2, 3, 258, 259, 1000], 2, seed=4), 0),
             (rng.integers(0, 2**32, 1000, dtype=np.uint64).astype(np.uint32), 5), (util.poisson_words(33000), 23)]
    with tempfile.TemporaryDirectory() as d:
        for i, (w, b) in enumerate(cases):
            src, dst, back = os.path.join(d, f"i{i}"), os.path.join(d, f"o{i}.zip"), os.path.join(d, f"b{i}")
            w.tofile(src)
            subprocess.check_call([ref, "-i", src, "-o", dst, "-b", str(b), "-t", "zip"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            z = open(dst, "rb").read()
            assert oracle.compress(w.tobytes(), b) == z
            # cross-decode: the reference decodes what the oracle wrote
            subprocess.check_call([ref, "-i", dst, "-o", back, "-t", "unzip"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            assert open(back, "rb").read() == oracle.uncompress(z)


def test_deflate_model_equals_zlib_on_fuzz(oracle):
    rng = np.random.default_rng(7)

    def check(x):
        x = np.ascontiguousarray(x, dtype=np.uint8)
        a, b = oracle.deflate(x), oracle.deflate(x, use_zlib=True)
        assert a == b, len(x)
        assert oracle.inflate(a, len(x)) == x.tobytes()

    def runs(n, maxrun, alpha):
        out, tot = [], 0
        while tot < n:
            l = int(rng.integers(1, maxrun + 1))
            out.append(np.full(l, int(rng.integers(0, alpha)), np.uint8))
            tot += l
        return np.concatenate(out)[:n]

    for n in [1, 2, 3, 4, 257, 258, 259, 260, 261, 262, 516, 519, 520, 32767, 32768, 65273, 65274, 65278, 65536, 65537, 98046, 98304, 131072]:
        check(np.zeros(n, np.uint8))
        check(rng.integers(0, 256, n, dtype=np.uint8))
        check(runs(n, 600, 3))
        check(runs(n, 5, 256))
    fib = [1, 1]
    while sum(fib) < 30000:
        fib.append(fib[-1] + fib[-2])
    x = np.concatenate([np.full(f, i, np.uint8) for i, f in enumerate(fib)])
    rng.shuffle(x)
    check(x)  # > 15-bit code lengths: overflow repair
    for _ in range(10):
        n = int(rng.integers(1, 300000))
        check(rng.choice(256, n, p=rng.dirichlet(np.full(256, 0.05))).astype(np.uint8))
    check(np.where(rng.random(6291456) < 0.97, rng.integers(0, 256, 6291456), 0).astype(np.uint8))  # full plane: stored/dynamic mix


def test_edge_cases(oracle):
    assert oracle.compress(b"", 0) == b""          # workers.c:757: empty input -> empty output
    assert oracle.compress(b"abc", 0) == b""       # < 1 float
    z = oracle.compress(util.kat_words(1).tobytes(), 8)
    assert len(z) == 17 + 16 + 4 and oracle.uncompress(z) == util.kat_words(1).tobytes()  # 4 RAW planes of 1 byte
    with pytest.raises(RuntimeError):
        oracle.compress(b"\0" * 64, 33)            # table has 33 entries (workers.c:29-37)


def test_int_mode_oracle_matches_committed_reference_containers(oracle):
    """"-s int" (workers.c:125-175, 444-511): containers the reference wrote with -s int (make_golden.py) and what it
    decodes them to; inputs hold ties, values beyond char / int32, NaN and infinities (the x86-64 conversion made explicit)."""
    from golden.make_golden import int_cases
    for name, data in int_cases().items():
        meta = G["int_mode"][name]
        assert util.sha256(data) == meta["input_sha256"], "input generator drifted"
        ref = open(os.path.join(util.GOLDEN, name + ".zip"), "rb").read()
        assert util.sha256(ref) == meta["sha256"]
        assert oracle.compress_int(data) == ref, name
        dec = oracle.uncompress(ref, int_mode=True)
        assert util.sha256(dec) == meta["decoded_sha256"]
        n = len(data) // 4
        assert dec == util.int_mode_expected(np.frombuffer(data[: 4 * n], np.uint32)).tobytes()


@pytest.mark.skipif(util.ref_binary("mrc_tar_c") is None, reason="oracle/_ref not built (needs /root/reference)")
def test_int_mode_oracle_matches_live_reference_binary(oracle):
    ref = util.ref_binary("mrc_tar_c")
    w = util.int_mode_words(40000, seed=77)
    with tempfile.TemporaryDirectory() as d:
        src, dst, back = os.path.join(d, "i"), os.path.join(d, "o.zip"), os.path.join(d, "b")
        w.tofile(src)
        subprocess.check_call([ref, "-i", src, "-o", dst, "-b", "9", "-t", "zip", "-s", "int"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        z = open(dst, "rb").read()
        assert oracle.compress_int(w.tobytes()) == z
        subprocess.check_call([ref, "-i", dst, "-o", back, "-t", "unzip", "-s", "int"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        assert open(back, "rb").read() == oracle.uncompress(z, int_mode=True) == util.int_mode_expected(w).tobytes()


def test_oracle_decodes_lz4_byte_streams(oracle):
    """ztypes 2 / 4 (LZ4_DEF / LZ4HC_DEF, mrczip.h:37-40): fixtures built with the reference's vendored LZ4 and decoded by the
    reference binary when they were generated (make_golden.py); the oracle's own LZ4 block decoder must give the input back."""
    from golden.make_golden import lz4_cases
    for name, (data, hc) in lz4_cases().items():
        meta = G["lz4"][name]
        z = open(os.path.join(util.GOLDEN, name + ".zip"), "rb").read()
        assert util.sha256(z) == meta["sha256"] and util.sha256(data) == meta["input_sha256"]
        assert z[13:17] == bytes([4 if hc else 2] * 4)
        assert oracle.uncompress(z) == data[: len(data) // 4 * 4], name
