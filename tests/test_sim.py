"""CPU-only: the product's HIP kernels (unmodified sources, compiled with g++ against the test-only
SIMT emulator in tests/sim) against the oracle.  This is how kernel logic is debugged without a GPU;
the GPU parity tests proper are in test_gpu_parity.py."""
import os

import numpy as np
import pytest

import util


def _check(sim, oracle, words, bits):
    words = np.ascontiguousarray(words, dtype=np.uint32)
    ref = oracle.compress(words.tobytes(), bits)[17:]
    got = sim.compress_records(words, bits)
    assert got == ref, (len(words), bits, len(got), len(ref))
    dec = sim.uncompress_records(ref, len(words))
    assert np.array_equal(dec, util.erase_expected(words, bits))
    assert sim.fallbacks == 0  # the parallel decoder handled every stream itself


@pytest.mark.parametrize("n", [1, 3, 255, 256, 257, 4095, 4096, 4097, 10000])
def test_ragged_sizes(simlib, oracle, n):
    rng = np.random.default_rng(n)
    w = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    for b in (0, 8, 32):
        _check(simlib, oracle, w, b)


@pytest.mark.parametrize("bits", [0, 8, 12, 20, 23, 31])
def test_gaussian_multiblock(simlib, oracle, bits):
    _check(simlib, oracle, util.gauss_words(140000, seed=2), bits)


def test_run_lengths_around_258(simlib, oracle):
    lens = [1, 2, 3, 4, 5, 257, 258, 259, 260, 261, 262, 515, 516, 517, 518, 519, 520, 63, 64, 65, 127, 128, 129, 4095, 4096, 4097]
    _check(simlib, oracle, util.runs_words(100000, lens, 3, seed=5), 0)
    _check(simlib, oracle, util.runs_words(60000, [1, 1, 1, 2, 3, 300, 1000, 5000], 2, seed=6), 0)
    _check(simlib, oracle, np.zeros(200000, np.uint32), 0)          # one run across 49 tiles, window slides
    _check(simlib, oracle, np.full(70000, 0x41200000, np.uint32), 0)


def test_skewed_alphabets_force_length_overflow(simlib, oracle):
    # Fibonacci byte frequencies push Huffman depths past 15 bits (overflow repair, and the cost recomputed from lengths)
    fib = [1, 1]
    while sum(fib) < 30000:
        fib.append(fib[-1] + fib[-2])
    rng = np.random.default_rng(3)
    p0 = np.concatenate([np.full(f, i, np.uint32) for i, f in enumerate(fib)])
    rng.shuffle(p0)
    w = p0 | (rng.integers(0, 4, len(p0), dtype=np.uint64).astype(np.uint32) << 8)
    _check(simlib, oracle, np.tile(w, 2), 0)


def test_planes_that_begin_with_stored_blocks(simlib, oracle):
    # noise, then constants: every plane's stream starts with a STORED block and goes on with coded ones, so the candidate at
    # bit 0 is a stored block (this is the input on which the emulator caught waves of the block decoder disagreeing on the
    # number of barriers: thread 0 changed sh.status while slower waves were still testing it)
    rng = np.random.default_rng(11)
    for nn, nz in ((34000, 40000), (70000, 70000)):
        w = np.concatenate([rng.integers(0, 2**32, nn, dtype=np.uint64).astype(np.uint32), np.full(nz, 0x41200000, np.uint32)])
        _check(simlib, oracle, w, 0)
    # coded blocks, a run of stored ones, coded blocks again: k_chain ranks the first run of dynamic blocks, walks the stored
    # ones (speculating on their common length) and ranks the second run from where they end
    low = lambda n: rng.integers(0, 7, n, dtype=np.uint64).astype(np.uint32) * np.uint32(0x01010101)
    w = np.concatenate([low(80000), rng.integers(0, 2**32, 100000, dtype=np.uint64).astype(np.uint32), low(90000)])
    _check(simlib, oracle, w, 0)


def test_tile_of_long_codes_exceeds_the_emit_staging_buffer(simlib, oracle):
    # a block of mostly four frequent byte values (2-bit codes) with a stretch of 250 rare values (10..12-bit codes) that fills whole
    # tiles: those tiles' bits (4096 x ~12) do not fit k_emit's 5 KiB staging buffer and are emitted in two halves
    rng = np.random.default_rng(5)
    common = lambda n: rng.integers(0, 4, n, dtype=np.uint64).astype(np.uint32)
    rare = (4 + rng.integers(0, 250, 5000, dtype=np.uint64)).astype(np.uint32)
    w = np.concatenate([common(12000), rare, common(40000)])
    _check(simlib, oracle, w, 0)


def test_kat_a(simlib, oracle):
    w = util.kat_words(300000)
    for b in (0, 23):
        _check(simlib, oracle, w, b)


def test_committed_reference_containers(simlib):
    from golden.make_golden import small_cases
    for name, (data, bits) in small_cases().items():
        ref = open(os.path.join(util.GOLDEN, name + ".zip"), "rb").read()
        n = len(data) // 4
        w = np.frombuffer(data[: 4 * n], np.uint32)
        assert simlib.compress_records(w, bits) == ref[17:], name


def test_int_mode_matches_the_reference_containers(simlib, oracle):
    """"-s int" (workers.c:125-175, 444-511): committed containers written by the reference with -s int, the oracle's
    restatement, and the decoded words (float)(signed char)(char)round(x)."""
    from golden.make_golden import int_cases
    for name, data in int_cases().items():
        ref = open(os.path.join(util.GOLDEN, name + ".zip"), "rb").read()
        assert oracle.compress_int(data) == ref, name
        n = len(data) // 4
        w = np.frombuffer(data[: 4 * n], np.uint32)
        assert simlib.compress_records(w, 0, int_mode=True) == ref[17:], name
        exp = util.int_mode_expected(w)
        assert np.array_equal(simlib.uncompress_records(ref[17:], n, int_mode=True), exp), name
        assert oracle.uncompress(ref, int_mode=True) == exp.tobytes(), name


def test_device_generator_equals_the_appendix_d_generator(simlib):
    """mrcz_generate_kat_words (the on-device source of the 64 GiB benchmark volume) == tests/util.py kat_words, at any offset"""
    ref = util.kat_words(70000)
    assert np.array_equal(simlib.generate_kat(0, 70000), ref)
    assert np.array_equal(simlib.generate_kat(12345, 40001), ref[12345:12345 + 40001])
    assert np.array_equal(simlib.generate_kat(69999, 1), ref[69999:])
    # the LCG has period 2^32 (two steps per word) and the stripe pattern period 2^14: word i + 2^31 == word i
    assert np.array_equal(simlib.generate_kat((1 << 31) + 5, 9000), ref[5:9005])
    assert np.array_equal(simlib.generate_kat((1 << 34) + (1 << 31) + 4090, 100), ref[4090:4190])


def test_lz4_byte_streams_decode(simlib, oracle):
    """Decoder tolerance: containers whose byte streams are LZ4 / LZ4HC blocks (header ztypes 2 / 4), written
    with the reference's own vendored LZ4 by tests/golden/make_golden.py and verified there with the reference binary."""
    from golden.make_golden import lz4_cases
    for name, (data, hc) in lz4_cases().items():
        z = open(os.path.join(util.GOLDEN, name + ".zip"), "rb").read()
        n = len(data) // 4
        assert oracle.uncompress(z) == data[: 4 * n], name
        assert simlib.set_ztypes(list(z[13:17])) == 0
        got = simlib.uncompress_records(z[17:], n)
        assert got.tobytes() == data[: 4 * n], name
    assert simlib.set_ztypes([0, 1, 0, 0]) != 0      # ZLIB_INF is not a stream type a file can carry
    assert simlib.set_ztypes([0, 0, 0, 0]) == 0
