"""GPU parity tests (the tests proper): the HIP path, called through the C ABI, against the CPU
oracle on seeded inputs, against the committed reference containers / hashes, and -- at sizes the
oracle would take too long for -- through size-independent properties (round trip == erasebytes,
header arithmetic).  Integer/byte work: everything is compared bit-exactly."""
import json
import os
import struct

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

G = json.load(open(os.path.join(util.GOLDEN, "golden.json")))


@pytest.fixture(scope="module")
def codec():
    import torch
    from datacompressionfloat_amd import MrcZipCodec
    assert torch.cuda.is_available()
    c = MrcZipCodec(0, max_batch_chunks=8)
    yield c
    c.close()


def _roundtrip(codec, oracle, words, bits, check_oracle=True):
    data = np.ascontiguousarray(words, dtype=np.uint32).tobytes()
    z = codec.zip_bytes(data, bits)
    if check_oracle:
        ref = oracle.compress(data, bits)
        assert len(z) == len(ref) and z == ref, (len(words), bits, len(z), len(ref))
    back = codec.unzip_bytes(z)
    assert back == util.erase_expected(np.frombuffer(data, np.uint32), bits).tobytes()
    assert codec.last_fallbacks() == 0  # decoded by the parallel kernel, not the sequential one
    return z


@pytest.mark.parametrize("n", [1, 2, 3, 5, 64, 255, 256, 257, 300, 4095, 4096, 4097, 32767, 32768, 32769, 65536, 100000])
def test_ragged_sizes_all_planes(codec, oracle, n):
    rng = np.random.default_rng(n)
    w = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    for b in (0, 8, 16, 24, 32):
        _roundtrip(codec, oracle, w, b)


@pytest.mark.parametrize("bits", list(range(0, 33)))
def test_all_33_mask_levels_roundtrip_equals_erasebytes(codec, oracle, bits):
    # the reference's own test loop (script/run_full_test.sh:84-102) + container byte identity
    _roundtrip(codec, oracle, util.gauss_words(300000, seed=11), bits)


def test_survey_appendix_d_known_answers(codec):
    for name in ("katA", "katB"):
        data = util.kat_words(G["kat_inputs"][name]).tobytes()
        for b, (size, h) in G["survey_appendix_d"][name].items():
            z = codec.zip_bytes(data, int(b))
            assert len(z) == size, (name, b, len(z))
            assert util.sha256(z).startswith(h), (name, b)
            back = codec.unzip_bytes(z)
            assert util.sha256(back) == G["regenerated"][f"{name}_b{b}"]["decoded_sha256"]


def test_committed_reference_containers(codec):
    from golden.make_golden import small_cases
    for name, (data, bits) in small_cases().items():
        ref = open(os.path.join(util.GOLDEN, name + ".zip"), "rb").read()
        assert codec.zip_bytes(data, bits) == ref, name
        back = codec.unzip_bytes(ref)  # decode what the REFERENCE wrote
        n = len(data) // 4 * 4
        assert back == util.erase_expected(np.frombuffer(data[:n], np.uint32), bits).tobytes()


def test_seeded_volumes_match_reference_hashes(codec):
    gens = {"gauss_4Mi_b8": lambda: util.gauss_words(4 * 1048576, seed=1234),
            "gauss_16Mi_b8": lambda: util.gauss_words(16 * 1048576, seed=1234),
            "poisson_7Mi_b0": lambda: util.poisson_words(7 * 1048576, seed=7),
            # BASELINE config 3 at its full size: 1024-byte header + 1024 x 1024 x 16 detector counts (mrc_small_full.sh shape)
            "poisson_mrc_small_b0": lambda: util.poisson_words(256 + 16 * 1048576, seed=7),
            "poisson_mrc_small_b8": lambda: util.poisson_words(256 + 16 * 1048576, seed=7)}
    for name, gen in gens.items():
        meta = G["large"][name]
        data = gen().tobytes()
        assert util.sha256(data) == meta["input_sha256"], "numpy generator drifted"
        z = codec.zip_bytes(data, meta["bits"])
        assert (len(z), util.sha256(z)) == (meta["size"], meta["sha256"]), name
        assert codec.unzip_bytes(z) == util.erase_expected(np.frombuffer(data, np.uint32), meta["bits"]).tobytes()


def test_int_mode(codec, oracle):
    """"-s int" of the reference (workers.c:125-175, 444-511): committed containers the reference wrote with -s int, the
    oracle on a fresh input, a two-chunk volume by hash; decode = (float)(signed char)(char)round(x) past the header."""
    from golden.make_golden import int_cases
    for name, data in int_cases().items():
        ref = open(os.path.join(util.GOLDEN, name + ".zip"), "rb").read()
        assert codec.zip_bytes(data, 11, mode="int") == ref, name          # the mask level is ignored in this mode
        n = len(data) // 4
        exp = util.int_mode_expected(np.frombuffer(data[: 4 * n], np.uint32)).tobytes()
        assert codec.unzip_bytes(ref, mode="int") == exp, name
        assert util.sha256(exp) == G["int_mode"][name]["decoded_sha256"]
    w = util.int_mode_words(500000, seed=31)
    z = codec.zip_bytes(w.tobytes(), 0, mode="int")
    assert z == oracle.compress_int(w.tobytes())
    assert codec.unzip_bytes(z, mode="int") == util.int_mode_expected(w).tobytes()
    meta = G["large"]["int_mode_7Mi"]
    big = util.int_mode_words(7 * 1048576, seed=24)
    assert util.sha256(big.tobytes()) == meta["input_sha256"]
    z = codec.zip_bytes(big.tobytes(), 0, mode="int")
    assert (len(z), util.sha256(z)) == (meta["size"], meta["sha256"])
    assert codec.unzip_bytes(z, mode="int") == util.int_mode_expected(big).tobytes()


def test_run_structures(codec, oracle):
    lens = [1, 2, 3, 4, 5, 257, 258, 259, 260, 261, 262, 515, 516, 517, 518, 519, 520, 63, 64, 65, 127, 128, 129, 4095, 4096, 4097]
    _roundtrip(codec, oracle, util.runs_words(1000000, lens, 3, seed=5), 0)
    _roundtrip(codec, oracle, util.runs_words(500000, [1, 1, 1, 2, 3, 300, 1000, 5000, 70000], 2, seed=6), 0)
    _roundtrip(codec, oracle, np.zeros(6291456 + 5, np.uint32), 0)        # single-block planes, full chunk + 5 floats
    _roundtrip(codec, oracle, np.full(3000000, 0x41200000, np.uint32), 7)


def test_skewed_alphabets_force_length_overflow(codec, oracle):
    # Fibonacci byte frequencies in plane 0 push Huffman depths past 15 bits (App. B.3 overflow repair)
    fib = [1, 1]
    while sum(fib) < 30000:
        fib.append(fib[-1] + fib[-2])
    rng = np.random.default_rng(3)
    p0 = np.concatenate([np.full(f, i, np.uint32) for i, f in enumerate(fib)])
    rng.shuffle(p0)
    w = p0 | (rng.integers(0, 4, len(p0), dtype=np.uint64).astype(np.uint32) << 8)
    _roundtrip(codec, oracle, np.tile(w, 5), 0)


def test_tile_of_long_codes_exceeds_the_emit_staging_buffer(codec, oracle):
    # see tests/test_sim.py: tiles whose bits do not fit k_emit's staging buffer are emitted in two halves
    rng = np.random.default_rng(5)
    common = lambda n: rng.integers(0, 4, n, dtype=np.uint64).astype(np.uint32)
    rare = (4 + rng.integers(0, 250, 5000, dtype=np.uint64)).astype(np.uint32)
    _roundtrip(codec, oracle, np.concatenate([common(12000), rare, common(40000), rare, common(300000)]), 0)


def test_huffman_header_paths_agree(monkeypatch):
    # the dynamic headers come from k_huffman_hdr (one wave per tree, runs of equal lengths in closed form) by default and from the
    # per-thread walk of k_huffman with MRCZ_HUFF_SPLIT=0: the two must write the same bytes on any input
    import torch
    from datacompressionfloat_amd import MrcZipCodec
    monkeypatch.setenv("MRCZ_HUFF_SPLIT", "0")
    mono = MrcZipCodec(0, max_batch_chunks=4)
    monkeypatch.setenv("MRCZ_HUFF_SPLIT", "1")
    split = MrcZipCodec(0, max_batch_chunks=4)
    rng = np.random.default_rng(2026)
    for case in range(18):
        n = int(rng.integers(1, 1_500_000))
        kind = case % 6
        if kind == 0: w = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
        elif kind == 1: w = rng.normal(10, 3, n).astype(np.float32).view(np.uint32)
        elif kind == 2: w = rng.poisson(8.0, n).astype(np.float32).view(np.uint32)
        elif kind == 3: w = (rng.integers(0, 4, n, dtype=np.uint64) ** 3).astype(np.uint32) * np.uint32(0x01010101)
        elif kind == 4: w = np.repeat(rng.integers(0, 2**32, n // 97 + 1, dtype=np.uint64).astype(np.uint32), 97)[:n]
        else: w = rng.geometric(0.02, n).astype(np.uint32) | (rng.integers(0, 3, n, dtype=np.uint64).astype(np.uint32) << 16)
        bits = int(rng.choice([0, 4, 8, 12, 16, 20, 23, 28, 32]))
        t = torch.from_numpy(np.ascontiguousarray(w).view(np.int32)).cuda()
        a, _ = mono.compress_device(t, bits, 0)
        b, _ = split.compress_device(t, bits, 0)
        assert torch.equal(a, b), (case, kind, n, bits, a.numel(), b.numel())
    mono.close(); split.close()


def test_header_validators_agree(monkeypatch, oracle):
    # candidate headers are parsed one wave each by default, one lane each with MRCZ_VALIDATE_WAVE=0: both must hand every
    # block of every stream to the parallel decoder (no fallbacks) and decode the same bytes
    from datacompressionfloat_amd import MrcZipCodec
    monkeypatch.setenv("MRCZ_VALIDATE_WAVE", "0")
    lanes = MrcZipCodec(0, max_batch_chunks=4)
    monkeypatch.setenv("MRCZ_VALIDATE_WAVE", "1")
    waves = MrcZipCodec(0, max_batch_chunks=4)
    for w, bits in ((util.gauss_words(2_000_000, seed=3), 8), (util.poisson_words(1_500_000, seed=4), 0),
                    (util.runs_words(1_000_000, [1, 2, 3, 300, 5000], 3, seed=8), 0), (util.kat_words(700_000), 12)):
        for c in (lanes, waves):
            _roundtrip(c, oracle, w, bits)
    lanes.close(); waves.close()


def test_planes_that_begin_with_stored_blocks(codec, oracle):
    # noise, then constants: the stream of every plane starts with STORED blocks and goes on with coded ones (see tests/test_sim.py)
    rng = np.random.default_rng(11)
    w = np.concatenate([rng.integers(0, 2**32, 300000, dtype=np.uint64).astype(np.uint32), np.full(400000, 0x41200000, np.uint32)])
    _roundtrip(codec, oracle, w, 0)
    assert codec.last_fallbacks() == 0
    # coded blocks, a run of stored ones, coded blocks again (see tests/test_sim.py)
    low = lambda n: rng.integers(0, 7, n, dtype=np.uint64).astype(np.uint32) * np.uint32(0x01010101)
    w = np.concatenate([low(800000), rng.integers(0, 2**32, 500000, dtype=np.uint64).astype(np.uint32), low(900000)])
    _roundtrip(codec, oracle, w, 0)


def test_chunk_boundaries(codec, oracle):
    C = util.CHUNK
    for n in (C - 1, C, C + 1, 2 * C + 12345):
        w = util.poisson_words(n, seed=n & 0xffff)
        _roundtrip(codec, oracle, w, 8)


def test_batches_smaller_than_the_volume(codec):
    """A volume of 5 chunks + tail through contexts whose batch is 2 chunks (three batches; the running record offset
    and the chunk-header walk carry over on the device) gives the same bytes as one batch."""
    n = 5 * util.CHUNK + 4321
    w = util.gauss_words(n, seed=5)
    small = type(codec)(0, max_batch_chunks=2)
    z_big = codec.zip_bytes(w.tobytes(), 8)
    assert small.zip_bytes(w.tobytes(), 8) == z_big
    exp = util.erase_expected(w, 8).tobytes()
    assert small.unzip_bytes(z_big) == exp
    assert codec.unzip_bytes(z_big) == exp
    small.close()


def test_two_lane_batches_with_a_short_last_batch(codec):
    """Two compress lanes (batches of >= 8 chunks run as two streams) over several batches whose last one is shorter:
    workspace rows must never be shared by two streams at a time.  26 chunks through a 16-chunk context = one batch of
    16 (lanes of 8 + 8) and one of 10 (5 + 5); a noisy first part and an all-zero second part make the lanes finish at
    very different times.  Same bytes as ONE batch, and as 7-chunk batches (single lane)."""
    import torch
    n = 26 * util.CHUNK - 777
    g = torch.Generator(device="cuda").manual_seed(99)
    words = torch.empty(n, dtype=torch.float32, device="cuda").normal_(10.0, 3.0, generator=g).view(torch.int32)
    words[13 * util.CHUNK + 5:] = 0
    one = type(codec)(0, max_batch_chunks=26)
    ref, ref_planes = one.compress_device(words, 4, 0)
    ref = ref.clone()
    one.close()
    for mb in (16, 12, 7):
        c = type(codec)(0, max_batch_chunks=mb)
        for _ in range(3):
            rec, planes = c.compress_device(words, 4, 0)
            assert planes == ref_planes and rec.numel() == ref.numel() and torch.equal(rec, ref), mb
        out, _ = c.uncompress_device(rec, n)
        exp = words.clone()
        c.erase_bits_device(exp, 4, 0)
        assert torch.equal(out, exp) and c.last_fallbacks() == 0
        c.close()


def test_empty_and_invalid(codec):
    from datacompressionfloat_amd import MrczError
    assert codec.zip_bytes(b"", 0) == b""
    assert codec.zip_bytes(b"abc", 5) == b""      # workers.c:757
    with pytest.raises(MrczError):
        codec.zip_bytes(b"\0" * 4096, 33)
    with pytest.raises(MrczError):
        codec.zip_bytes(b"\0" * 4096, -1)
    z = codec.zip_bytes(util.gauss_words(5000).tobytes(), 8)
    with pytest.raises(MrczError):
        codec.unzip_bytes(z[: len(z) // 2])       # truncated container is rejected, not read out of bounds
    with pytest.raises(MrczError):
        codec.unzip_bytes(z[:10])


def test_header_and_accounting(codec):
    data = util.gauss_words(70001).tobytes() + b"xyz"
    z = codec.zip_bytes(data, 8)
    fsz, chk, typ, zt = struct.unpack("<QIb", z[:13]) + (z[13:17],)
    assert (fsz, chk, typ, zt) == (len(data), 6291456, 0, b"\0\0\0\0")   # common.c:137-148
    assert len(codec.unzip_bytes(z)) == 4 * (len(data) // 4)               # trailing fsz%4 bytes dropped (workers.c:744,854)


def test_one_gib_roundtrip_property(codec):
    """BASELINE config 2 size: 1 GiB, b=8; too big for the oracle inside a test -> properties only."""
    import torch
    n = 268435456
    g = torch.Generator(device="cuda").manual_seed(1234)
    x = torch.empty(n, dtype=torch.float32, device="cuda").normal_(10.0, 3.0, generator=g)
    words = x.view(torch.int32)
    big = type(codec)(0, max_batch_chunks=43)
    rec, planes = big.compress_device(words, 8, 0)
    nchunks = (n + util.CHUNK - 1) // util.CHUNK
    assert sum(planes) == rec.numel()                          # zfsz sums include the 4-byte plane headers = 16 B per chunk
    out, consumed = big.uncompress_device(rec, n)
    assert consumed == rec.numel() and big.last_fallbacks() == 0
    exp = words.clone()
    big.erase_bits_device(exp, 8, 0)
    assert torch.equal(out, exp)
    # plane 0 is all zero after masking 8 bits: ~6 KiB per chunk; plane 1 is incompressible -> RAW
    assert rec.numel() < 0.6 * 4 * n
    big.close()


def test_eight_gib_multi_batch_roundtrip(codec, oracle):
    """BASELINE config 4 shape on one GPU, scaled to 8 GiB: the SURVEY App. D volume generated on the device, 342 chunks through
    a 128-chunk context = three batches inside ONE call each way (running record offset, header walk, scratch reuse, two
    compress lanes per batch).  Too big for the oracle -> properties, plus the first chunk record against the oracle."""
    import torch
    n = 2 * 1024 * 1024 * 1024  # floats = 8 GiB
    big = type(codec)(0, max_batch_chunks=128)
    words = torch.empty(n, dtype=torch.int32, device="cuda")
    big.generate_kat_device(words, 0)
    assert np.array_equal(words[:100000].cpu().numpy().view(np.uint32), util.kat_words(100000))
    rec, planes = big.compress_device(words, 8, 0)
    assert sum(planes) == rec.numel()
    first = oracle.compress(util.kat_words(util.CHUNK).tobytes(), 8)[17:]
    assert rec[: len(first)].cpu().numpy().tobytes() == first          # chunk 0 of the volume == the oracle's container of that chunk
    out, consumed = big.uncompress_device(rec, n)
    assert consumed == rec.numel() and big.last_fallbacks() == 0
    big.erase_bits_device(words, 8, 0)                                     # in place: words is now erasebytes(input)
    assert torch.equal(out, words)
    big.close()


def _container_from_python_zlib(words, strategy, level=6):
    """A container in the reference's format whose plane streams were written by the system zlib with a
    DIFFERENT strategy (general distances / fixed codes / Huffman only): exercises the decoder's
    fallback paths (SURVEY 8(f)-4 decoder tolerance)."""
    import zlib
    n = len(words)
    b = words.view(np.uint8).reshape(-1, 4)
    out = bytearray(struct.pack("<QIb4b", 4 * n, util.CHUNK, 0, 0, 0, 0, 0))
    for c0 in range(0, n, util.CHUNK):
        c1 = min(n, c0 + util.CHUNK)
        hdr, pay = bytearray(), bytearray()
        for j in range(4):
            plane = np.ascontiguousarray(b[c0:c1, j]).tobytes()
            co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
            z = co.compress(plane) + co.flush(zlib.Z_FULL_FLUSH)
            if len(plane) > len(z) + 4:
                hdr += struct.pack("<I", len(z))
                pay += z
            else:
                hdr += struct.pack("<I", len(plane) | 0x80000000)
                pay += plane
        out += hdr + pay
    return bytes(out)


def test_lz4_byte_streams_decode(codec):
    """Decoder tolerance: LZ4 / LZ4HC byte streams (header ztypes 2 / 4, zip.c:69-86), fixtures written with the
    reference's vendored LZ4; afterwards the context decodes deflate containers again."""
    from golden.make_golden import lz4_cases
    for name, (data, hc) in lz4_cases().items():
        z = open(os.path.join(util.GOLDEN, name + ".zip"), "rb").read()
        assert codec.unzip_bytes(z) == data[: len(data) // 4 * 4], name
    w = util.gauss_words(50000, seed=4)
    assert codec.unzip_bytes(codec.zip_bytes(w.tobytes(), 8)) == util.erase_expected(w, 8).tobytes()
    bad = bytearray(open(os.path.join(util.GOLDEN, "lz4_runs.zip"), "rb").read())
    bad[14] = 1                                     # ZLIB_INF is not a type a file can carry
    from datacompressionfloat_amd import MrczError
    with pytest.raises(MrczError):
        codec.unzip_bytes(bytes(bad))


def test_foreign_streams_decode_through_fallbacks(codec):
    import zlib
    w = util.poisson_words(700000, seed=5)
    w[300000:300400] = w[1000:1400]            # a long-distance repeat: distance != 1 under the default strategy
    for strategy, expect_seq in ((zlib.Z_DEFAULT_STRATEGY, True), (zlib.Z_FIXED, False), (zlib.Z_HUFFMAN_ONLY, False), (zlib.Z_RLE, False)):
        z = _container_from_python_zlib(w, strategy)
        back = codec.unzip_bytes(z)
        assert back == w.tobytes(), strategy
        if expect_seq:
            assert codec.last_fallbacks() > 0     # general distances went to the sequential decoder
        if strategy == zlib.Z_RLE:
            assert codec.last_fallbacks() == 0 and z == codec.zip_bytes(w.tobytes(), 0)
