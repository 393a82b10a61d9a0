#!/usr/bin/env python3
"""SURVEY 8(d) side measurements (not the bench line): mask-level sweep on the 1 GiB volume, the mrc_small shape
(config 3), the command-line tools end to end (file -> pinned host -> HBM -> file, PCIe and page cache included) and
the reference's own single-thread numbers on the 64 MiB block (config 1).  Writes one JSON document to stdout."""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from datacompressionfloat_amd import MrcZipCodec

CHUNK = 6 * 1048576
dev = torch.device("cuda", 0)


def volume(n, seed=1234):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.empty(n, dtype=torch.float32, device=dev).normal_(10.0, 3.0, generator=g).view(torch.int32)
    x[:256] = 0
    x[0], x[1], x[2], x[3] = 4096, 4096, n // (4096 * 4096), 2
    return x


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, r


def measure(codec, words, bits):
    n = words.numel()
    rec_buf = torch.empty(codec.records_bound(n), dtype=torch.uint8, device=dev)
    out_buf = torch.empty(n, dtype=torch.int32, device=dev)
    tc, (rec, _) = timed(lambda: codec.compress_device(words, bits, 0, out=rec_buf))
    td, (out, _) = timed(lambda: codec.uncompress_device(rec, n, out=out_buf))
    exp = words.clone()
    codec.erase_bits_device(exp, bits, 0)
    assert torch.equal(out, exp), f"round trip differs at b={bits}"
    return {"bits": bits, "ratio": round(rec.numel() / (4.0 * n), 4), "compress_GBps": round(4.0 * n / tc / 1e9, 1),
            "decompress_GBps": round(4.0 * n / td / 1e9, 1), "decoder_fallback_streams": int(codec.last_fallbacks())}


res = {"device": torch.cuda.get_device_name(0), "note": "HBM-resident, 3 repetitions after one warm-up, bit-exact round trip asserted"}
n = 268435456
if "--cli-only" not in sys.argv:
    codec = MrcZipCodec(0, max_batch_chunks=43)
    w = volume(n)
    res["config2_1GiB_mask_sweep"] = [measure(codec, w, b) for b in (0, 8, 12, 16, 23)]
    del w
    # config 3: mrc_small shape -- 1024-byte header + 1024 x 1024 x 16 Poisson-like detector counts as float32
    g = torch.Generator(device=dev).manual_seed(7)
    cnt = torch.poisson(torch.full((16 * 1048576,), 8.0, device=dev), generator=g)
    w3 = torch.cat([torch.zeros(256, dtype=torch.float32, device=dev), cnt]).view(torch.int32).contiguous()
    res["config3_mrc_small_64MiB_poisson"] = [measure(codec, w3, b) for b in (0, 8)]
    del w3, cnt
    # the same detector-count statistics at the 1 GiB size of config 2 (what a large .mrc stack looks like to the codec)
    cnt = torch.poisson(torch.full((n - 256,), 8.0, device=dev), generator=g)
    w4 = torch.cat([torch.zeros(256, dtype=torch.float32, device=dev), cnt]).view(torch.int32).contiguous()
    del cnt
    res["poisson_counts_1GiB"] = [measure(codec, w4, b) for b in (0, 8)]
    del w4
    codec.close()

# command-line tools end to end on a 1 GiB file in /dev/shm (what a user of mrc_tar sees: file I/O + PCIe + codec)
shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
with tempfile.TemporaryDirectory(dir=shm) as d:
    src, z, back = os.path.join(d, "vol.mrc"), os.path.join(d, "vol.zip"), os.path.join(d, "vol.out")
    volume(n).cpu().numpy().tofile(src)
    tool = os.path.join(ROOT, "datacompressionfloat_amd", "bin", "mrc_tar")
    e2e = {}
    torch.cuda.empty_cache()  # (this process keeps its GPU context while the tools run: their start-up is slower than stand-alone,
    #  tools/jobs/cli_wall_times.sh measures them without it)
    for name, cmd in (("zip", [tool, "-i", src, "-o", z, "-t", "zip", "-b", "8"]), ("unzip", [tool, "-i", z, "-o", back, "-t", "unzip"])):
        subprocess.run(cmd, stdout=subprocess.DEVNULL, check=True)  # warm (page cache, context creation)
        if name == "unzip" and os.path.exists(back):
            os.remove(back)  # (freeing a 1 GiB tmpfs file is not part of the tool's work)
        t0 = time.perf_counter()
        subprocess.run(cmd, stdout=subprocess.DEVNULL, check=True)
        dt = time.perf_counter() - t0
        e2e[name] = {"wall_s": round(dt, 3), "GBps_of_floats": round(4.0 * n / dt / 1e9, 2)}
    res["cli_end_to_end_1GiB_devshm"] = e2e
    # config 5 shape, scaled to one GPU: a list of 8 files x 256 MiB through mrc_tarx (worker threads keep their codec
    # session between files), zip then unzip, wall time of the whole command
    tarx = os.path.join(ROOT, "datacompressionfloat_amd", "bin", "mrc_tarx")
    files = []
    for i in range(8):
        f = os.path.join(d, f"part{i}.mrc")
        volume(n // 4, seed=100 + i).cpu().numpy().tofile(f)
        files.append(f)
    lst, zl = os.path.join(d, "files.txt"), os.path.join(d, "zips.txt")
    zdir, udir = os.path.join(d, "z"), os.path.join(d, "u")
    os.mkdir(zdir); os.mkdir(udir)
    open(lst, "w").write("\n".join(files) + "\n")
    multi = {}
    for nthreads in (1, 2, 4):
        t0 = time.perf_counter()
        subprocess.run([tarx, "-i", lst, "-t", "zip", "-o", zdir, "-b", "8", "-n", str(nthreads)], stdout=subprocess.DEVNULL, check=True)
        t1 = time.perf_counter()
        zips = sorted(os.path.join(zdir, x) for x in os.listdir(zdir))
        open(zl, "w").write("\n".join(zips) + "\n")
        subprocess.run([tarx, "-i", zl, "-t", "unzip", "-o", udir, "-n", str(nthreads)], stdout=subprocess.DEVNULL, check=True)
        t2 = time.perf_counter()
        multi[f"threads_{nthreads}"] = {"zip_GBps": round(8 * n / (t1 - t0) / 1e9, 2), "unzip_GBps": round(8 * n / (t2 - t1) / 1e9, 2)}
        for x in zips:
            os.remove(x)
        for x in os.listdir(udir):
            os.remove(os.path.join(udir, x))
    res["cli_mrc_tarx_8x256MiB_devshm"] = multi
    # the reference itself, one thread, 64 MiB block (config 1): compress + decompress wall time
    ref = os.path.join(ROOT, "oracle", "_ref", "mrc_tar_c")
    if os.path.exists(ref):
        small, zs, bs = os.path.join(d, "s.mrc"), os.path.join(d, "s.zip"), os.path.join(d, "s.out")
        volume(16777216).cpu().numpy().tofile(small)
        t0 = time.perf_counter()
        subprocess.run([ref, "-i", small, "-o", zs, "-t", "zip", "-b", "8"], stdout=subprocess.DEVNULL, check=True)
        t1 = time.perf_counter()
        subprocess.run([ref, "-i", zs, "-o", bs, "-t", "unzip"], stdout=subprocess.DEVNULL, check=True)
        t2 = time.perf_counter()
        res["config1_reference_1thread_64MiB"] = {"compress_s": round(t1 - t0, 2), "decompress_s": round(t2 - t1, 2),
                                                  "compress_GBps": round(67108864 / (t1 - t0) / 1e9, 3),
                                                  "decompress_GBps": round(67108864 / (t2 - t1) / 1e9, 3)}
print(json.dumps(res, indent=1))
