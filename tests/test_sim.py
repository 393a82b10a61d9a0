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
