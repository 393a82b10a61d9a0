"""GPU: the headline and the large configurations of BASELINE.json at their real sizes, pinned against the oracle.

  config 2   1 GiB N(10, 3^2) volume of SURVEY 8(d) (numpy default_rng(1234)), b = 8: the WHOLE container equals the oracle's
  config 4   64 GiB SURVEY App. D volume streamed through ONE GPU in 128-chunk batches: every batch round-trips to
             erasebytes(input); the first, a middle and the last (partial) chunk record equal the oracle's
  config 5   mrc_tarx over several 2 GiB files (run_mt_large_full_test.sh shape, script/run_mt_large_full_test.sh:83-102):
             two containers equal the oracle's, every file round-trips to erasebytes(input)
The 8-GPU forms of configs 4 and 5 need hardware this suite does not have; what runs here is their one-GPU form.
"""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

BIN = os.path.join(util.ROOT, "datacompressionfloat_amd", "bin")
CORES = max(1, min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 64))


def _chunk_record_of_the_oracle(oracle, chunk_words: np.ndarray, bits: int, first: bool) -> bytes:
    """The 16-byte header + 4 payloads the oracle writes for one chunk.  A chunk that is not the file's first one has no
    header exemption (workers.c:777,804): it is coded behind an all-zero stand-in chunk and the stand-in's record skipped."""
    if first:
        return oracle.compress(chunk_words, bits)[17:]
    z = oracle.compress(np.concatenate([np.zeros(util.CHUNK, np.uint32), chunk_words]), bits)
    h = struct.unpack("<4I", z[17:33])
    return z[17 + 16 + sum(x & 0x7fffffff for x in h):]


def _record_bounds(rec, nchunks: int):
    """byte offsets of the chunk records inside a device buffer of records (walks the 16-byte headers, workers.c:52-69)"""
    offs, off = [], 0
    for _ in range(nchunks):
        h = struct.unpack("<4I", rec[off: off + 16].cpu().numpy().tobytes())
        offs.append(off)
        off += 16 + sum(x & 0x7fffffff for x in h)
    offs.append(off)
    return offs


def test_one_gib_container_equals_the_oracle(oracle):
    """BASELINE config 2 at its size: the container of the SURVEY 8(d) volume, byte for byte (43 chunks, 172 streams)."""
    import torch
    from datacompressionfloat_amd import MrcZipCodec
    n = 268435456
    w = util.gauss_words(n, seed=1234)
    ref = oracle.compress(w, 8, threads=CORES)
    big = MrcZipCodec(0, max_batch_chunks=43)
    dev = torch.from_numpy(w.view(np.int32)).cuda()
    rec, planes = big.compress_device(dev, 8, 0)
    assert rec.numel() == len(ref) - 17 and sum(planes) == rec.numel()
    got = rec.cpu().numpy()
    assert np.array_equal(got, np.frombuffer(ref, np.uint8)[17:]), "the 1 GiB container differs from the oracle's"
    import json
    pin = json.load(open(os.path.join(util.GOLDEN, "golden.json")))["config2_1GiB_b8"]
    assert (len(ref), util.sha256(ref)) == (pin["container_bytes"], pin["sha256"])          # the size and hash recorded when the test first ran
    out, consumed = big.uncompress_device(rec, n)
    assert consumed == rec.numel() and big.last_fallbacks() == 0
    assert np.array_equal(out.cpu().numpy().view(np.uint32), util.erase_expected(w, 8))   # numpy's erasebytes, not the library's
    big.close()


def test_one_gib_twelve_bits_equals_the_oracle(oracle):
    """The same volume at b = 12 (BASELINE's mask sweep): plane 1 becomes a plane of 4-bit codes, every block of which is one full
    16 KiB window and a stub of ~500 bytes -- the case the merge's three-segments-per-tile path and the hint-sized windows exist for."""
    import torch
    from datacompressionfloat_amd import MrcZipCodec
    n = 268435456
    w = util.gauss_words(n, seed=1234)
    ref = oracle.compress(w, 12, threads=CORES)
    big = MrcZipCodec(0, max_batch_chunks=43)
    dev = torch.from_numpy(w.view(np.int32)).cuda()
    rec, planes = big.compress_device(dev, 12, 0)
    assert rec.numel() == len(ref) - 17 and sum(planes) == rec.numel()
    assert np.array_equal(rec.cpu().numpy(), np.frombuffer(ref, np.uint8)[17:]), "the 1 GiB container (b = 12) differs from the oracle's"
    out, consumed = big.uncompress_device(rec, n)
    assert consumed == rec.numel() and big.last_fallbacks() == 0
    assert np.array_equal(out.cpu().numpy().view(np.uint32), util.erase_expected(w, 12))
    big.close()


def test_sixty_four_gib_streamed_through_one_gpu(oracle):
    """BASELINE config 4 (bnr_large.sh shape) on ONE GPU: 2731 chunks in 22 batches of 128, what bench.py --gib-per-gpu 64 does."""
    import torch
    from datacompressionfloat_amd import MrcZipCodec
    total = 16 * (1 << 30)                                        # floats = 64 GiB
    free, _ = torch.cuda.mem_get_info()
    if free < 80 * (1 << 30):
        pytest.skip(f"needs 80 GiB of free HBM, this device has {free >> 30}")
    B = 128
    big = MrcZipCodec(0, max_batch_chunks=B)
    words = torch.empty(total, dtype=torch.int32, device="cuda")
    big.generate_kat_device(words, 0)
    bfl = B * util.CHUNK
    nbatch = (total + bfl - 1) // bfl
    nchunks = (total + util.CHUNK - 1) // util.CHUNK
    assert (nbatch, nchunks) == (22, 2731)
    rec_buf = torch.empty(big.records_bound(bfl), dtype=torch.uint8, device="cuda")
    out_buf = torch.empty(bfl, dtype=torch.int32, device="cuda")
    picks = {0: 0, 10: 57, nbatch - 1: (nchunks - 1) % B}         # batch -> chunk inside it: first, a middle one, the last (partial)
    zbytes = 0
    for b in range(nbatch):
        sub = words[b * bfl: min(total, (b + 1) * bfl)]
        rec, planes = big.compress_device(sub, 8, b * B, out=rec_buf)
        assert sum(planes) == rec.numel()
        zbytes += rec.numel()
        out, consumed = big.uncompress_device(rec, sub.numel(), out=out_buf[: sub.numel()])
        assert consumed == rec.numel() and big.last_fallbacks() == 0, b
        if b in picks:
            k = picks[b]
            cw = sub[k * util.CHUNK: (k + 1) * util.CHUNK].cpu().numpy().view(np.uint32)
            if b == nbatch - 1:
                assert len(cw) == 4194304                         # 64 GiB = 2730 full chunks + 4 Mi floats
            offs = _record_bounds(rec, (sub.numel() + util.CHUNK - 1) // util.CHUNK)
            mine = rec[offs[k]: offs[k + 1]].cpu().numpy().tobytes()
            assert mine == _chunk_record_of_the_oracle(oracle, cw, 8, first=(b == 0 and k == 0)), (b, k)
            # the device's erasebytes against numpy's on this chunk (the batches below are compared with the device's)
            e = cw.copy()
            e[(256 if (b == 0 and k == 0) else 0):] &= np.uint32(0xffffff00)
            assert np.array_equal(out[k * util.CHUNK: (k + 1) * util.CHUNK].cpu().numpy().view(np.uint32), e)
        exp = sub.clone()
        big.erase_bits_device(exp, 8, b * bfl)
        assert torch.equal(out, exp), f"batch {b} does not round-trip to erasebytes(input)"
        del exp
    print("64 GiB App. D volume: container payload", zbytes, "bytes")
    big.close()


def test_mrc_tarx_over_two_gib_files(tmp_path, oracle):
    """BASELINE config 5 shape on one GPU: mrc_tarx -n k over a list of 2 GiB files in /dev/shm."""
    import torch
    from datacompressionfloat_amd import MrcZipCodec
    exe = os.path.join(BIN, "mrc_tarx")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    fl = 1 << 29                                                   # floats per file = 2 GiB
    per_file = 4 * fl * 2.7                                        # input + container + decoded file
    room = shutil.disk_usage(shm or str(tmp_path)).free
    nfiles = int(min(4, (room * 0.8) // per_file))
    if nfiles < 2:
        pytest.skip(f"needs room for two 2 GiB files and their outputs in {shm or tmp_path}, free {room >> 30} GiB")
    d = os.path.join(shm, f"mrcz_cfg5_{os.getpid()}") if shm else str(tmp_path / "cfg5")
    os.makedirs(d)
    try:
        gen = MrcZipCodec(0, max_batch_chunks=1)
        names = []
        for i in range(nfiles):
            w = torch.empty(fl, dtype=torch.int32, device="cuda")
            gen.generate_kat_device(w, i * fl + 977 * i)          # App. D generator with a per-file offset
            p = os.path.join(d, f"vol{i}.mrc")
            w.cpu().numpy().tofile(p)
            names.append(p)
            del w
        gen.close()
        torch.cuda.empty_cache()
        lst = os.path.join(d, "files.txt")
        open(lst, "w").write("\n".join(names) + "\n")
        zdir, udir = os.path.join(d, "z"), os.path.join(d, "u")
        os.mkdir(zdir)
        os.mkdir(udir)
        r = subprocess.run([exe, "-i", lst, "-t", "zip", "-o", zdir, "-b", "8", "-n", str(nfiles)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        assert r.returncode == 0, r.stderr
        assert "MB/s" in r.stdout                                  # mrc_tarx.c:231
        znames = [os.path.join(zdir, f"vol{i}.mrc.zip") for i in range(nfiles)]   # adapt.c:303-305
        for i in (0, nfiles - 1):                                  # two containers against the oracle, byte for byte
            w = np.fromfile(names[i], np.uint32)
            ref = oracle.compress(w, 8, threads=CORES)
            got = np.fromfile(znames[i], np.uint8)
            assert len(got) == len(ref) and np.array_equal(got, np.frombuffer(ref, np.uint8)), i
            del w, ref, got
        zl = os.path.join(d, "zips.txt")
        open(zl, "w").write("\n".join(znames) + "\n")
        r = subprocess.run([exe, "-i", zl, "-t", "unzip", "-o", udir, "-n", str(nfiles)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        assert r.returncode == 0, r.stderr
        for i in range(nfiles):                                    # every file by round trip (run_mt_large_full_test.sh:83-102)
            w = np.fromfile(names[i], np.uint32)
            back = np.fromfile(os.path.join(udir, f"vol{i}.mrc"), np.uint32)   # adapt.c:307-308
            assert np.array_equal(back, util.erase_expected(w, 8)), i
            del w, back
    finally:
        shutil.rmtree(d, ignore_errors=True)
