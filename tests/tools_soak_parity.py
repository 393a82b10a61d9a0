"""Developer tool (not collected by pytest): a timed soak of the GPU path against the CPU oracle on random, structured inputs.

    python tests/tools_soak_parity.py [seconds=240] [seed=1]

Every case builds a volume out of random segments (noise, byte runs of critical lengths, small and skewed alphabets, floats
of a few distributions, constant stretches that span block and chunk borders), picks a mask level, and checks
  - the container bytes against oracle.compress (byte identity),
  - the round trip against the erased input,
  - that no stream fell back to the sequential decoder;
one case in ten runs in "-s int" mode (container against oracle.compress_int, round trip against the quantised input).
Prints one line per failure (with the case's seed, so it can be rebuilt) and a summary; exit code 1 on any failure.
Lives under tests/ because it uses the oracle (test infrastructure); the product never does.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import util  # noqa: E402

RUNS = [1, 2, 3, 4, 5, 63, 64, 65, 257, 258, 259, 260, 515, 516, 517, 4095, 4096, 4097, 32766, 32767, 32768, 32769, 65535, 70000]


def segment(rng, n):
    kind = int(rng.integers(0, 9))
    if kind == 0:
        return rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    if kind == 1:
        return rng.normal(rng.uniform(-50, 50), rng.uniform(0.01, 100), n).astype(np.float32).view(np.uint32)
    if kind == 2:
        return rng.poisson(rng.uniform(0.05, 40), n).astype(np.float32).view(np.uint32)
    if kind == 3:  # a constant stretch
        return np.full(n, rng.integers(0, 2**32, dtype=np.uint64), np.uint32)
    if kind == 4:  # byte runs of critical lengths, in one to four planes
        out = np.zeros(n, np.uint32)
        for plane in rng.choice(4, int(rng.integers(1, 5)), replace=False):
            lens = rng.choice(RUNS, max(1, n // 200))
            vals = rng.integers(0, 256, len(lens), dtype=np.uint64).astype(np.uint32)
            out |= np.resize(np.repeat(vals, lens), n).astype(np.uint32) << np.uint32(8 * plane)
        return out
    if kind == 5:  # tiny alphabets (short codes, two-literal table entries in the decoder)
        k = int(rng.integers(2, 6))
        a = rng.integers(0, 2**32, k, dtype=np.uint64).astype(np.uint32)
        return a[rng.integers(0, k, n)]
    if kind == 6:  # skewed alphabet: geometric byte values (long codes, length overflow when combined with noise)
        p = rng.uniform(0.01, 0.6)
        g = np.minimum(rng.geometric(p, n), 255).astype(np.uint32)
        return g | (np.minimum(rng.geometric(p, n), 255).astype(np.uint32) << 8) | (rng.integers(0, 3, n, dtype=np.uint64).astype(np.uint32) << 24)
    if kind == 7:  # Fibonacci frequencies in plane 0
        fib = [1, 1]
        while sum(fib) < n:
            fib.append(fib[-1] + fib[-2])
        v = np.concatenate([np.full(f, i & 255, np.uint32) for i, f in enumerate(fib)])[:n]
        rng.shuffle(v)
        return v
    return np.arange(n, dtype=np.uint32) * np.uint32(rng.integers(1, 2**20))  # counters: period-256 low plane


def volume(rng):
    scale = rng.choice([200, 5000, 100000, 1500000, 7000000, 14000000], p=[0.1, 0.15, 0.3, 0.25, 0.15, 0.05])
    n = int(rng.integers(1, scale + 1))
    parts, left = [], n
    while left > 0:
        m = int(min(left, rng.integers(1, max(2, n // int(rng.integers(1, 9))) + 1)))
        parts.append(segment(rng, m))
        left -= m
    return np.ascontiguousarray(np.concatenate(parts)[:n], dtype=np.uint32)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from datacompressionfloat_amd import MrcZipCodec
    oracle = util.load_oracle()
    codec = MrcZipCodec(0, max_batch_chunks=8)
    t0, cases, fails, nbytes = time.time(), 0, 0, 0
    while time.time() - t0 < budget:
        seed = seed0 * 1000003 + cases
        rng = np.random.default_rng(seed)
        w = volume(rng)
        bits = int(rng.choice([0, 0, 0, 1, 4, 7, 8, 9, 12, 15, 16, 17, 20, 23, 24, 25, 28, 31, 32]))
        data = w.tobytes()
        int_mode = len(w) > 256 and rng.random() < 0.1  # "-s int": the quantiser instead of the mask (workers.c:125-175)
        try:
            what = None
            if int_mode:
                z = codec.zip_bytes(data, 0, mode="int")
                ref = oracle.compress_int(data)
                if z != ref:
                    what = "int-mode container differs (%d vs %d bytes)" % (len(z), len(ref))
                elif codec.unzip_bytes(z, mode="int") != util.int_mode_expected(w).tobytes():
                    what = "int-mode round trip differs"
            else:
                z = codec.zip_bytes(data, bits)
                ref = oracle.compress(data, bits)
                if z != ref:
                    what = "container differs (%d vs %d bytes)" % (len(z), len(ref))
                elif codec.unzip_bytes(z) != util.erase_expected(w, bits).tobytes():
                    what = "round trip differs"
            if what is None and codec.last_fallbacks() != 0:
                what = "%d streams fell back to the sequential decoder" % codec.last_fallbacks()
        except Exception as e:  # noqa: BLE001
            what = "exception %r" % (e,)
        if what:
            fails += 1
            print("FAIL case seed=%d n=%d bits=%d int=%d: %s" % (seed, len(w), bits, int(int_mode), what), flush=True)
        cases += 1
        nbytes += len(data)
        if cases % 50 == 0:
            print("... %d cases, %.1f MiB, %d failures, %.0f s" % (cases, nbytes / 1048576, fails, time.time() - t0), flush=True)
    print("soak: %d cases, %.1f MiB, %d failures, %.0f s" % (cases, nbytes / 1048576, fails, time.time() - t0))
    codec.close()
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
