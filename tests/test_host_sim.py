"""CPU-only: the C host (run_compress / run_uncompress pipeline of host/workers_gpu.c, the file adapters and both
front-ends) linked against the SIMT-emulator build of the codec instead of libmrcz_hip.so.  Exercises the reader /
caller / writer threads, the event protocol of the asynchronous C ABI, the per-plane summary table and the
container bytes on inputs small enough for the emulator -- before any GPU minute is spent on them."""
import os
import subprocess

import numpy as np
import pytest

import util

HOST = os.path.join(util.ROOT, "datacompressionfloat_amd", "host")


@pytest.fixture(scope="module")
def simbin(tmp_path_factory):
    util.load_sim()  # builds tests/sim/libmrcz_sim.so
    d = tmp_path_factory.mktemp("hostsim")
    out = {}
    for main in ("mrc_tar", "mrc_tarx"):
        exe = d / main
        subprocess.check_call(["gcc", "-O1", "-g", "-std=gnu99", "-Wall", "-o", str(exe), os.path.join(HOST, main + ".c"),
                               os.path.join(HOST, "workers_gpu.c"), os.path.join(HOST, "common_gpu.c"), os.path.join(HOST, "adapt_gpu.c"),
                               "-L" + util.SIM_DIR, "-lmrcz_sim", "-lpthread", "-lm", "-lstdc++", "-Wl,-rpath," + util.SIM_DIR])
        out[main] = str(exe)
    return out


def _run(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=e)


@pytest.mark.parametrize("n,bits,tail", [(100, 0, b""), (70001, 8, b"xyz"), (300000, 12, b"")])
def test_mrc_tar_on_the_emulator(simbin, oracle, tmp_path, n, bits, tail):
    w = util.gauss_words(n, seed=n & 255)
    src, z, back = tmp_path / "in.mrc", tmp_path / "o.zip", tmp_path / "b.mrc"
    src.write_bytes(w.tobytes() + tail)
    r = _run([simbin["mrc_tar"], "-i", str(src), "-o", str(z), "-b", str(bits), "-t", "zip"])
    assert r.returncode == 0, r.stderr
    assert z.read_bytes() == oracle.compress(w.tobytes() + tail, bits)
    assert "Whole File" in r.stdout and "Compression Summary Result Information" in r.stdout      # print_result, zip.c:401-466
    r = _run([simbin["mrc_tar"], "-i", str(z), "-o", str(back), "-t", "unzip"])
    assert r.returncode == 0, r.stderr
    assert back.read_bytes() == util.erase_expected(w, bits).tobytes()
    assert "Decompress Result Info Information" in r.stdout


def test_empty_and_tiny_inputs(simbin, tmp_path):
    src, z = tmp_path / "e.mrc", tmp_path / "e.zip"
    src.write_bytes(b"abc")                                       # fewer than 4 bytes: nothing is written (workers.c:757)
    r = _run([simbin["mrc_tar"], "-i", str(src), "-o", str(z), "-t", "zip"])
    assert r.returncode == 0 and z.read_bytes() == b""


def test_mrc_tarx_threads_share_one_engine(simbin, oracle, tmp_path):
    srcdir, zdir, udir = tmp_path / "s", tmp_path / "z", tmp_path / "u"
    for d in (srcdir, zdir, udir):
        d.mkdir()
    names, datas = [], {}
    for i, n in enumerate((5000, 40000, 257, 90000)):
        p = srcdir / f"f{i}.mrc"
        w = util.poisson_words(n, seed=i)
        p.write_bytes(w.tobytes())
        names.append(str(p))
        datas[f"f{i}"] = w
    lst = tmp_path / "l.txt"
    lst.write_text("\n".join(names) + "\n")
    r = _run([simbin["mrc_tarx"], "-i", str(lst), "-t", "zip", "-o", str(zdir), "-b", "4", "-n", "3"])
    assert r.returncode == 0, r.stderr
    zs = []
    for stem, w in datas.items():
        z = zdir / f"{stem}.mrc.zip"
        assert z.read_bytes() == oracle.compress(w.tobytes(), 4), stem
        zs.append(str(z))
    lst2 = tmp_path / "z.txt"
    lst2.write_text("\n".join(zs) + "\n")
    r = _run([simbin["mrc_tarx"], "-i", str(lst2), "-t", "unzip", "-o", str(udir), "-n", "2"])
    assert r.returncode == 0, r.stderr
    for stem, w in datas.items():
        assert (udir / f"{stem}.mrc").read_bytes() == util.erase_expected(w, 4).tobytes()
    # throughput mode: the codec runs, nothing is written (workers.c:39)
    tdir = tmp_path / "t"
    tdir.mkdir()
    r = _run([simbin["mrc_tarx"], "-i", str(lst), "-t", "zip", "-o", str(tdir), "-n", "2", "-d", "1"])
    assert r.returncode == 0 and all(os.path.getsize(tdir / f"{s}.mrc.zip") == 0 for s in datas)
