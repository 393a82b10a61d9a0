"""GPU: the C front-ends (same command line as the reference's mrc_tar_c / mrc_tarx_c) write the
reference's bytes and read the reference's files."""
import os
import subprocess

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

BIN = os.path.join(util.ROOT, "datacompressionfloat_amd", "bin")


def _run(args, **kw):
    return subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, **kw)


def test_mrc_tar_zip_unzip_matches_oracle(tmp_path, oracle):
    exe = os.path.join(BIN, "mrc_tar")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    for n, bits in ((100, 0), (70001, 8), (util.CHUNK + 12345, 8), (3 * util.CHUNK + 5, 12)):
        w = util.gauss_words(n, seed=n & 1023)
        src, dst, back = tmp_path / "in.mrc", tmp_path / "out.zip", tmp_path / "back.mrc"
        data = w.tobytes() + (b"xyz" if n == 70001 else b"")
        src.write_bytes(data)
        r = _run([exe, "-i", str(src), "-o", str(dst), "-b", str(bits), "-t", "zip"])
        assert r.returncode == 0, r.stderr
        assert dst.read_bytes() == oracle.compress(data, bits)
        r = _run([exe, "-i", str(dst), "-o", str(back), "-t", "unzip"])
        assert r.returncode == 0, r.stderr
        assert back.read_bytes() == util.erase_expected(w, bits).tobytes()


def test_pipeline_over_many_small_batches(tmp_path, oracle):
    """The reader / caller / writer pipeline of host/workers_gpu.c with one chunk per batch: 5 batches in flight over the two
    device buffers and the ring slots (buffer reuse, event chains), both directions, bytes as the oracle's."""
    exe = os.path.join(BIN, "mrc_tar")
    n = 4 * util.CHUNK + 4321
    w = util.gauss_words(n, seed=3)
    src, dst, back = tmp_path / "in.mrc", tmp_path / "out.zip", tmp_path / "back.mrc"
    src.write_bytes(w.tobytes())
    env = dict(os.environ, MRCZ_BATCH_CHUNKS="1")
    r = _run([exe, "-i", str(src), "-o", str(dst), "-b", "8", "-t", "zip"], env=env)
    assert r.returncode == 0, r.stderr
    ref = oracle.compress(w.tobytes(), 8, threads=8)
    assert dst.read_bytes() == ref
    r = _run([exe, "-i", str(dst), "-o", str(back), "-t", "unzip"], env=env)
    assert r.returncode == 0, r.stderr
    assert back.read_bytes() == util.erase_expected(w, 8).tobytes()
    env2 = dict(os.environ, MRCZ_BATCH_CHUNKS="2")
    assert _run([exe, "-i", str(src), "-o", str(dst), "-b", "8", "-t", "zip"], env=env2).returncode == 0
    assert dst.read_bytes() == ref


def test_mrc_tar_rejects_bad_arguments(tmp_path):
    exe = os.path.join(BIN, "mrc_tar")
    src = tmp_path / "in.mrc"
    src.write_bytes(util.gauss_words(1000).tobytes())
    assert _run([exe, "-i", str(src), "-o", str(tmp_path / "o"), "-b", "33"]).returncode != 0
    assert _run([exe, "-i", str(tmp_path / "missing"), "-o", str(tmp_path / "o")]).returncode != 0


def test_mrc_tarx_file_list_naming_and_threads(tmp_path, oracle):
    exe = os.path.join(BIN, "mrc_tarx")
    srcdir, zdir, udir = tmp_path / "src", tmp_path / "z", tmp_path / "u"
    for d in (srcdir, zdir, udir):
        d.mkdir()
    names, datas = [], {}
    for i, n in enumerate((5000, 300000, util.CHUNK + 77)):
        p = srcdir / f"stack{i}.mrc"
        w = util.poisson_words(n, seed=i)
        p.write_bytes(w.tobytes())
        names.append(str(p))
        datas[f"stack{i}"] = w
    lst = tmp_path / "files.txt"
    lst.write_text("\n".join(names) + "\n")
    r = _run([exe, "-i", str(lst), "-t", "zip", "-o", str(zdir), "-b", "8", "-n", "2"])
    assert r.returncode == 0, r.stderr
    assert "MB/s" in r.stdout                                   # mrc_tarx.c:231 summary line
    znames = []
    for stem, w in datas.items():
        z = zdir / f"{stem}.mrc.zip"                            # adapt.c:303-305 naming
        assert z.read_bytes() == oracle.compress(w.tobytes(), 8)
        znames.append(str(z))
    lst2 = tmp_path / "zips.txt"
    lst2.write_text("\n".join(znames) + "\n")
    r = _run([exe, "-i", str(lst2), "-t", "unzip", "-o", str(udir), "-n", "3"])
    assert r.returncode == 0, r.stderr
    for stem, w in datas.items():
        assert (udir / f"{stem}.mrc").read_bytes() == util.erase_expected(w, 8).tobytes()   # adapt.c:307-308
    # throughput mode writes nothing (workers.c:39, -d 1)
    tdir = tmp_path / "t"
    tdir.mkdir()
    r = _run([exe, "-i", str(lst), "-t", "zip", "-o", str(tdir), "-b", "8", "-n", "2", "-d", "1"])
    assert r.returncode == 0 and all(os.path.getsize(tdir / f"{s}.mrc.zip") == 0 for s in datas)
    bad = tmp_path / "bad.txt"
    bad.write_text(str(srcdir / "x.dat") + "\n")
    assert _run([exe, "-i", str(bad), "-t", "zip", "-o", str(zdir)]).returncode != 0     # adapt.c:309-311


@pytest.mark.skipif(util.ref_binary("mrc_tar_c") is None, reason="oracle/_ref not present on this box")
def test_cross_decode_with_the_reference_binary(tmp_path):
    exe, ref = os.path.join(BIN, "mrc_tar"), util.ref_binary("mrc_tar_c")
    w = util.gauss_words(400000, seed=77)
    src, z1, z2, b1, b2 = (tmp_path / n for n in ("in.mrc", "gpu.zip", "ref.zip", "b1", "b2"))
    src.write_bytes(w.tobytes())
    assert _run([exe, "-i", str(src), "-o", str(z1), "-b", "10", "-t", "zip"]).returncode == 0
    assert _run([ref, "-i", str(src), "-o", str(z2), "-b", "10", "-t", "zip"]).returncode == 0
    assert z1.read_bytes() == z2.read_bytes()
    assert _run([ref, "-i", str(z1), "-o", str(b1), "-t", "unzip"]).returncode == 0      # reference decodes GPU output
    assert _run([exe, "-i", str(z2), "-o", str(b2), "-t", "unzip"]).returncode == 0      # GPU decodes reference output
    assert b1.read_bytes() == b2.read_bytes() == util.erase_expected(w, 10).tobytes()


@pytest.mark.skipif(util.ref_binary("mrc_tar_c") is None, reason="oracle/_ref not present on this box")
def test_int_mode_cli_cross_decode_with_the_reference_binary(tmp_path):
    """mrc_tar -s int (workers.c:782-787, 604-609): same container as the reference binary, each decodes the other's file"""
    exe, ref = os.path.join(BIN, "mrc_tar"), util.ref_binary("mrc_tar_c")
    w = util.int_mode_words(300000, seed=41)
    src, z1, z2, b1, b2 = (tmp_path / n for n in ("in.mrc", "gpu.zip", "ref.zip", "b1", "b2"))
    src.write_bytes(w.tobytes())
    assert _run([exe, "-i", str(src), "-o", str(z1), "-t", "zip", "-s", "int"]).returncode == 0
    assert _run([ref, "-i", str(src), "-o", str(z2), "-t", "zip", "-s", "int"]).returncode == 0
    assert z1.read_bytes() == z2.read_bytes()
    assert _run([ref, "-i", str(z1), "-o", str(b1), "-t", "unzip", "-s", "int"]).returncode == 0
    assert _run([exe, "-i", str(z2), "-o", str(b2), "-t", "unzip", "-s", "int"]).returncode == 0
    assert b1.read_bytes() == b2.read_bytes() == util.int_mode_expected(w).tobytes()


def test_erasebytes_tool_matches_the_reference_tool(tmp_path):
    """the verification tool with the reference's command line, masking on the GPU; compared with the
    reference's own erasebytes (built as oracle/_ref/erasebytes_c) where present, and with numpy always."""
    exe = os.path.join(BIN, "erasebytes")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    ref = os.path.join(util.ROOT, "oracle", "_ref", "erasebytes_c")
    for n, bits, tail in ((100, 8, b""), (256, 8, b""), (70001, 12, b"xy"), (2 * util.CHUNK + 3, 23, b"")):
        w = util.gauss_words(n, seed=n & 511)
        src, dst = tmp_path / "in.mrc", tmp_path / "out.mrc"
        src.write_bytes(w.tobytes() + tail)
        r = _run([exe, "-i", str(src), "-o", str(dst), "-b", str(bits)])
        assert r.returncode == 0, r.stderr
        got = dst.read_bytes()
        assert got == util.erase_expected(w, bits).tobytes()
        if os.path.exists(ref) and 4 * n >= 1024:  # below 1024 bytes the reference writes 1024 bytes of its malloc'd buffer
            rdst = tmp_path / "ref.mrc"
            r = _run([ref, "-i", str(src), "-o", str(rdst), "-b", str(bits)])
            assert r.returncode == 0 and rdst.read_bytes() == got


def _accounting_row(stdout: str):
    """(allFileSize, allZipFileSize) from the row print_context_info writes under its four column titles (common.c:66-90)"""
    lines = stdout.splitlines()
    for i, l in enumerate(lines):
        if l.startswith("[Original File Size(Bytes)]"):
            f = lines[i + 1].split()
            return int(f[0]), int(f[1])
    raise AssertionError("no accounting row in:\n" + stdout)


def _plane_table(stdout: str):
    """print_result (zip.c:401-466): [(before, after)] of the four byte streams + the 'Whole File' row"""
    rows = []
    for l in stdout.splitlines():
        f = l.split()
        if len(f) == 4 and f[0] in ("0", "1", "2", "3") and f[1].isdigit() and f[2].isdigit():
            rows.append((int(f[1]), int(f[2])))
        elif l.startswith("Whole File"):
            rows.append((int(f[2]), int(f[3])))
    assert len(rows) == 5, stdout
    return rows


@pytest.mark.skipif(util.ref_binary("mrc_tar_c") is None, reason="oracle/_ref not present on this box")
def test_ctx_accounting_matches_the_reference(tmp_path):
    """zip -> sum of mzip_t.zfsz == container size - 17 (payload + 4-byte plane headers, workers.c:870-873);
    unzip -> allFileSize == 4 * floats, allZipFileSize == payload bytes (workers.c:679-685).  The per-plane table
    (print_result, zip.c:401-466) and the context row (common.c:66-90) carry the same numbers as the reference binary
    prints on the same input."""
    exe, ref = os.path.join(BIN, "mrc_tar"), util.ref_binary("mrc_tar_c")
    n = util.CHUNK + 54321
    w = util.gauss_words(n, seed=5)
    src, z, zr, b = (tmp_path / x for x in ("in.mrc", "gpu.zip", "ref.zip", "back"))
    src.write_bytes(w.tobytes() + b"\x01\x02")          # a file size that is not a multiple of 4
    r = _run([exe, "-i", str(src), "-o", str(z), "-b", "8", "-t", "zip"])
    rr = _run([ref, "-i", str(src), "-o", str(zr), "-b", "8", "-t", "zip"])
    assert r.returncode == 0 and rr.returncode == 0
    t = _plane_table(r.stdout)
    assert t == _plane_table(rr.stdout)
    assert t[4] == (4 * n, os.path.getsize(z) - 17)
    fs, zs = _accounting_row(r.stdout)                  # our mrc_tar also prints the context row after a zip
    assert (fs, zs) == (4 * n + 2, os.path.getsize(z) - 17)
    r = _run([exe, "-i", str(z), "-o", str(b), "-t", "unzip"])
    rr = _run([ref, "-i", str(zr), "-o", str(tmp_path / "rb"), "-t", "unzip"])
    assert r.returncode == 0 and rr.returncode == 0
    assert _plane_table(r.stdout) == _plane_table(rr.stdout)
    fs, zs = _accounting_row(r.stdout)
    assert (fs, zs) == (4 * n, os.path.getsize(z) - 17 - 16 * 2)
    assert (fs, zs) == _accounting_row(rr.stdout)


@pytest.mark.skipif(util.ref_binary("mrc_tar_refmain_gpu") is None, reason="oracle/_ref not present on this box")
def test_reference_front_ends_run_on_the_drop_in_library(tmp_path, oracle):
    """the reference's own src/main/mrc_tar.c and mrc_tarx.c, compiled unchanged and linked against
    libmrcz_workers.so (oracle/Makefile), produce the reference's bytes with the codec on the GPU."""
    w = util.gauss_words(200000, seed=9)
    src, z, b = tmp_path / "in.mrc", tmp_path / "o.zip", tmp_path / "b.mrc"
    src.write_bytes(w.tobytes())
    exe = util.ref_binary("mrc_tar_refmain_gpu")
    r = _run([exe, "-i", str(src), "-o", str(z), "-b", "12", "-t", "zip"])
    assert r.returncode == 0, r.stderr
    assert z.read_bytes() == oracle.compress(w.tobytes(), 12)
    assert _run([exe, "-i", str(z), "-o", str(b), "-t", "unzip"]).returncode == 0
    assert b.read_bytes() == util.erase_expected(w, 12).tobytes()
    # mrc_tarx main(): list file in, <out>/<name>.mrc.zip out (adapt.c:297-311)
    exx = util.ref_binary("mrc_tarx_refmain_gpu")
    lst, zdir = tmp_path / "l.txt", tmp_path / "zz"
    zdir.mkdir()
    lst.write_text(str(src) + "\n")
    r = _run([exx, "-i", str(lst), "-t", "zip", "-o", str(zdir), "-b", "12", "-n", "1"])
    assert r.returncode == 0, r.stderr
    assert (zdir / "in.mrc.zip").read_bytes() == oracle.compress(w.tobytes(), 12)


def test_one_file_over_two_engines_on_this_gpu(tmp_path, oracle):
    """SURVEY 8(e) in the C host, rehearsed on one GPU: MRCZ_DEVICES=2 deals the batches of one file to two logical devices,
    MRCZ_DEVICE_ALIAS=1 folds both onto the GPU that is present (two engines: contexts, streams, events, buffers of their
    own).  Same container bytes as with one device, and as the oracle."""
    exe = os.path.join(BIN, "mrc_tar")
    n = 5 * util.CHUNK + 999
    w = util.gauss_words(n, seed=12)
    src, z, back = tmp_path / "in.mrc", tmp_path / "o.zip", tmp_path / "b.mrc"
    src.write_bytes(w.tobytes())
    env = dict(os.environ, MRCZ_DEVICES="2", MRCZ_DEVICE_ALIAS="1", MRCZ_BATCH_CHUNKS="1", MRCZ_TRACE="1")
    r = _run([exe, "-i", str(src), "-o", str(z), "-b", "8", "-t", "zip"], env=env)
    assert r.returncode == 0, r.stderr
    assert r.stderr.count("mrcz_create") == 2, r.stderr
    assert z.read_bytes() == oracle.compress(w.tobytes(), 8, threads=8)
    r = _run([exe, "-i", str(z), "-o", str(back), "-t", "unzip"], env=env)
    assert r.returncode == 0, r.stderr
    assert back.read_bytes() == util.erase_expected(w, 8).tobytes()


@pytest.mark.skipif(util.ref_binary("erroranalysis_c") is None, reason="oracle/_ref not present on this box")
def test_erroranalysis_tool_matches_the_reference_tool(tmp_path):
    """bin/erroranalysis (selection of the worst points on the GPU: radix select + candidate hand-back) prints
    the same lines as the reference's erroranalysis on a 3-chunk volume against its own decoded file (script/run_full_test.sh:106)."""
    exe, ref, tar = os.path.join(BIN, "erroranalysis"), util.ref_binary("erroranalysis_c"), os.path.join(BIN, "mrc_tar")
    n = 2 * util.CHUNK + 4567
    w = util.gauss_words(n, seed=21)
    a, z, b = tmp_path / "a.mrc", tmp_path / "a.zip", tmp_path / "b.mrc"
    a.write_bytes(w.tobytes())
    assert _run([tar, "-i", str(a), "-o", str(z), "-b", "13", "-t", "zip"]).returncode == 0
    assert _run([tar, "-i", str(z), "-o", str(b), "-t", "unzip"]).returncode == 0
    for k in ("2", "10"):
        mine, theirs = _run([exe, "-a", str(a), "-b", str(b), "-k", k]), _run([ref, "-a", str(a), "-b", str(b), "-k", k])
        assert mine.returncode == 0 and theirs.returncode == 0, (mine.stderr, theirs.stderr)
        assert mine.stdout == theirs.stdout and len(mine.stdout.splitlines()) == int(k)


def test_mrc_tar_decodes_lz4_containers(tmp_path):
    """the C host passes the header's ztypes on (run_uncompress, workers.c:584): an LZ4HC container through bin/mrc_tar"""
    from golden.make_golden import lz4_cases
    exe = os.path.join(BIN, "mrc_tar")
    data, hc = lz4_cases()["lz4hc_poisson"]
    out = tmp_path / "o.mrc"
    r = _run([exe, "-i", os.path.join(util.GOLDEN, "lz4hc_poisson.zip"), "-o", str(out), "-t", "unzip"])
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == data


def test_damaged_containers_end_with_an_error_status_not_a_signal(tmp_path):
    """exit(-1) semantics of the reference (workers.c:708-712) when the error is raised by a pipeline thread while the others still
    use the session's pinned rings, device buffers and events: a plain non-zero status, never a signal out of an exit handler.
    Then one orderly run with MRCZ_FULL_TEARDOWN=1 (the sessions are released through atexit) must give the same bytes as the
    fast-exit path."""
    from test_host_sim import _damaged_containers
    exe, exex = os.path.join(BIN, "mrc_tar"), os.path.join(BIN, "mrc_tarx")
    w = util.gauss_words(util.CHUNK + 60000, seed=5)
    src, z, z2 = tmp_path / "in.mrc", tmp_path / "o.zip", tmp_path / "o2.zip"
    src.write_bytes(w.tobytes())
    assert _run([exe, "-i", str(src), "-o", str(z), "-b", "8", "-t", "zip"]).returncode == 0
    good = z.read_bytes()
    for what, bad in _damaged_containers(good).items():
        b = tmp_path / "bad.zip"
        b.write_bytes(bad)
        r = _run([exe, "-i", str(b), "-o", str(tmp_path / "x.mrc"), "-t", "unzip"])
        assert r.returncode > 0 and "ERROR" in r.stderr, (what, r.returncode, r.stderr)
        lst = tmp_path / "l.txt"
        lst.write_text(str(b) + "\n")
        r = _run([exex, "-i", str(lst), "-t", "unzip", "-o", str(tmp_path), "-n", "2"])
        assert r.returncode > 0, (what, r.returncode, r.stderr)
    two = tmp_path / "two.txt"                      # two worker threads that both fail to open their output (adapt.c:34-44)
    two.write_text(str(src) + "\n" + str(src) + "\n")
    r = _run([exex, "-i", str(two), "-t", "zip", "-o", str(tmp_path / "no" / "such" / "dir"), "-n", "2"])
    assert r.returncode > 0 and "fail open" in r.stderr, (r.returncode, r.stderr)
    env = dict(os.environ, MRCZ_FULL_TEARDOWN="1")
    assert _run([exe, "-i", str(src), "-o", str(z2), "-b", "8", "-t", "zip"], env=env).returncode == 0
    assert z2.read_bytes() == good
    r = _run([exe, "-i", str(z2), "-o", str(tmp_path / "ok.mrc"), "-t", "unzip"], env=env)
    assert r.returncode == 0 and (tmp_path / "ok.mrc").read_bytes() == util.erase_expected(w, 8).tobytes()
