#!/usr/bin/env python3
"""bench.py -- headline benchmark: compress + decompress GB/s of input floats on MI355X.

One "step" = one pass of the hot path over synthetic input that is already resident in HBM: compress the
rank's volume (mask + byte planes + DEFLATE Z_RLE -> chunk records) and decompress it again.

  N = 1 (default)   BASELINE.json configs[1]: 1 GiB synthetic float32 volume of SURVEY 8(d) (256 header words + N(10, 3^2),
                    numpy default_rng(1234)), single mask level b = 8, 43 chunks = 172 plane streams, one codec call each way.
  --gib-per-gpu G   G > 4 (or --stream): streaming mode for the large configurations (configs[3]: 64 GiB).  The
                    volume is generated ON the device by the SURVEY App. D integer generator and stays resident
                    (only the input); it is coded in batches of 128 chunks (3 GiB), each batch compressed,
                    decompressed and -- in one extra, untimed pass -- compared with erasebytes(input).
  N > 1             weak scaling: every rank owns a contiguous chunk range of the N x G GiB volume
                    (datacompressionfloat_amd/shard.py float_range: chunks are independent); the only
                    exchange is the concatenation gather of the compressed records to rank 0 over RCCL, inside the
                    timed region; rank 0 decodes the gathered container once, outside it.

Prints ONE JSON line on rank 0 (contract in the task statement), with these extra objects:
  "roofline"      dominant kernel vs the HBM roof (HIP events on the codec's own stream); "whole_path" = SURVEY 8(d)'s
                  (4N + Z) / t against the HBM peak for the compress and the decompress direction
  "cpu_baseline"  the reference's own pthread path (oracle/_ref) timed on this box's host cores, both directions, all cores
                  and one thread; "parity_check" = chunk records 0, middle, last of THIS run's container against the oracle
  "cli"           the C front-ends of this repo (mrc_tarx, file -> HBM -> file) in the reference's -d 1 throughput mode on the
                  same files as cpu_baseline (outside the timed region; PCIe and file reads included)

The step loop (class Pipeline) is importable: tests/test_bench_gloo.py runs the N > 1 sequence over gloo on the emulator codec.
"""
import argparse
import json
import os
import struct
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)
CHUNK = 6 * 1048576
STREAM_BATCH_CHUNKS = 128


def make_volume(nfloats, seed, first):
    """N(10, 3^2) float32 from numpy default_rng(seed) (the generator tests/util.gauss_words and the
    GPU test of the 1 GiB container use), first 256 words of the FILE a header.  Host array of uint32, made in slabs."""
    import numpy as np
    rng = np.random.default_rng(seed)
    w = np.empty(nfloats, np.uint32)
    slab = 1 << 24
    for a in range(0, nfloats, slab):
        b = min(nfloats, a + slab)
        w[a:b] = rng.normal(10.0, 3.0, b - a).astype(np.float32).view(np.uint32)
    if first and nfloats >= 256:
        w[:256] = 0
        w[0], w[1], w[2], w[3] = 4096, 4096, 1, 2
    return w


class Pipeline:
    """One rank's step: per batch, compress the chunk range, start the concatenation gather of its records on rank 0 (sizes via
    all_gather, records via grouped send/recv: datacompressionfloat_amd/shard.py), decompress.  Two record buffers: the gather of
    call i (its own stream) runs under the decompress of call i and the compress of call i + 1, which writes the other buffer; a
    gather is always finished before the next one starts."""

    def __init__(self, codec, words, bits, first_chunk, f_lo, bfl, batch_chunks, world, rank, dist, torch, shard, sync):
        self.codec, self.words, self.bits, self.first_chunk, self.f_lo = codec, words, bits, first_chunk, f_lo
        self.bfl, self.batch_chunks, self.world, self.rank, self.dist, self.torch, self.shard, self.sync = bfl, batch_chunks, world, rank, dist, torch, shard, sync
        nfloats = words.numel()
        self.nfloats = nfloats
        self.nbatch = (nfloats + bfl - 1) // bfl
        cap = codec.records_bound(min(bfl, nfloats))
        dev = words.device
        self.rec_bufs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(2 if world > 1 else 1)]
        self.out_buf = torch.empty(min(bfl, nfloats), dtype=torch.int32, device=dev)
        self.gather_buf = torch.empty(int(cap * world * 0.75) + 1024, dtype=torch.uint8, device=dev) if (world > 1 and rank == 0) else None
        self.pending = None
        self.ncall = 0

    def finish_gather(self):
        """the previous call's concatenation must be complete (on the device) before its buffers are reused"""
        if self.pending is not None:
            r = self.pending.wait()
            self.sync()
            self.pending = None
            return r
        return None

    def step(self, verify=False):
        tc = td = 0.0
        zbytes = 0
        ok = True
        last = None
        for b in range(self.nbatch):
            sub = self.words[b * self.bfl: min(self.nfloats, (b + 1) * self.bfl)]
            rec_buf = self.rec_bufs[self.ncall % len(self.rec_bufs)]
            self.ncall += 1
            t0 = time.perf_counter()
            rec, planes = self.codec.compress_device(sub, self.bits, self.first_chunk + b * self.batch_chunks, out=rec_buf)
            t1 = time.perf_counter()
            if self.world > 1:
                self.finish_gather()
                self.pending = self.shard.gather_records_start(rec, self.dist, dst=0, out=self.gather_buf)
            out, _ = self.codec.uncompress_device(rec, sub.numel(), out=self.out_buf[: sub.numel()])
            t2 = time.perf_counter()
            tc += t1 - t0
            td += t2 - t1
            zbytes += rec.numel()
            last = (rec, planes)
            if verify:  # bit-exact round trip == erasebytes(input), batch by batch (outside the timed region)
                exp = sub.clone()
                self.codec.erase_bits_device(exp, self.bits, self.f_lo + b * self.bfl)
                ok = ok and bool(self.torch.equal(out, exp))
                del exp
        return tc, td, zbytes, ok, last

    def check_gathered(self, gathered, total_floats):
        """rank 0: the concatenation of all ranks' records (rank order = file order) is the chunk-record part of ONE container of
        the whole volume: decode all of it; returns (decoded tensor, per-rank sizes)"""
        cat, sizes = gathered
        dec = self.torch.empty(total_floats, dtype=self.torch.int32, device=self.words.device)
        out_all, used = self.codec.uncompress_device(cat, total_floats, out=dec)
        assert used == sum(sizes), "the gathered records are not one decodable container"
        return out_all, sizes


def run_timed(pipe, steps, warmup, barrier):
    """W untimed steps, then exactly K steps between two barriers; returns (elapsed, compress seconds, decompress seconds)"""
    for _ in range(warmup):
        pipe.step()
    pipe.finish_gather()
    barrier()
    t0 = time.perf_counter()
    tc = td = 0.0
    for _ in range(steps):
        c, d, _, _, _ = pipe.step()
        tc += c
        td += d
    pipe.finish_gather()  # the last records have arrived on rank 0 before the clock stops
    barrier()
    return time.perf_counter() - t0, tc, td


def _ref_tools():
    ref = os.path.join(ROOT, "oracle", "_ref", "mrc_tarx_c")
    ref1 = os.path.join(ROOT, "oracle", "_ref", "mrc_tar_c")
    return (ref, ref1) if os.path.exists(ref) and os.path.exists(ref1) else (None, None)


def _tarx(exe, lst, op, outd, bits, n, extra_env=None):
    cmd = [exe, "-i", lst, "-t", op, "-o", outd, "-n", str(n), "-d", "1"] + (["-b", str(bits)] if op == "zip" else [])
    t0 = time.time()
    subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True, env=dict(os.environ, **(extra_env or {})))
    return time.time() - t0


def cpu_baseline_and_cli(sample_words, bits, cores):
    """cpu_baseline: the reference's own file-level pthread pool (src/main/mrc_tarx.c:134-176, built as oracle/_ref/mrc_tarx_c) in
    throughput mode (-d 1: no output writes, mrc_tarx.c:226-231) on a bounded sample: zip of N copies of the sample, then unzip of N
    copies of its container; value = bytes of floats through both directions / total wall time (the metric's definition); also
    with ONE thread.  cli: this repo's mrc_tarx on the SAME files with the same flag (file -> pinned ring -> HBM -> codec; no
    output writes), so that the two front-ends are compared like with like."""
    ref, ref1 = _ref_tools()
    if ref is None:
        # fall back to the in-repo restatement (still a CPU baseline, never the measured product)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import util
        o = util.load_oracle()
        t0 = time.time()
        z = o.compress(sample_words, bits, threads=cores)
        tz = time.time() - t0
        t0 = time.time()
        o.uncompress(z)
        tu = time.time() - t0
        return {"value": round(sample_words.nbytes / (tz + tu) / 1e9, 4), "unit": "GB/s", "cores": cores, "kind": "port",
                "sample": f"{sample_words.nbytes >> 20} MiB, chunk-parallel pthread oracle compress (wall {tz:.2f} s) + 1-thread oracle uncompress (wall {tu:.2f} s)"}, None
    nfiles = max(cores, 8)
    mine = os.path.join(ROOT, "datacompressionfloat_amd", "bin", "mrc_tarx")
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
        src, zsrc = os.path.join(d, "sample0.mrc"), os.path.join(d, "sample0.zip")
        sample_words.tofile(src)
        subprocess.run([ref1, "-i", src, "-o", zsrc, "-b", str(bits), "-t", "zip"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        names, znames = [src], [zsrc]
        for i in range(1, nfiles):
            p, z = os.path.join(d, f"sample{i}.mrc"), os.path.join(d, f"sample{i}.zip")
            os.link(src, p)
            os.link(zsrc, z)
            names.append(p)
            znames.append(z)
        lst, zlst, lst1, zlst1 = (os.path.join(d, n) for n in ("files.txt", "zips.txt", "one.txt", "onez.txt"))
        open(lst, "w").write("\n".join(names) + "\n")
        open(zlst, "w").write("\n".join(znames) + "\n")
        open(lst1, "w").write(names[0] + "\n")
        open(zlst1, "w").write(znames[0] + "\n")
        outd = os.path.join(d, "out")
        os.mkdir(outd)
        tz, tu = _tarx(ref, lst, "zip", outd, bits, cores), _tarx(ref, zlst, "unzip", outd, bits, cores)
        tz1, tu1 = _tarx(ref, lst1, "zip", outd, bits, 1), _tarx(ref, zlst1, "unzip", outd, bits, 1)
        total, one = nfiles * sample_words.nbytes, sample_words.nbytes
        cpu = {"value": round(total / (tz + tu) / 1e9, 4), "unit": "GB/s", "cores": cores, "kind": "reference",
               "compress_GBps": round(total / tz / 1e9, 4), "decompress_GBps": round(total / tu / 1e9, 4),
               "one_thread": {"value": round(one / (tz1 + tu1) / 1e9, 4), "compress_GBps": round(one / tz1 / 1e9, 4), "decompress_GBps": round(one / tu1 / 1e9, 4),
                              "sample": f"1 file, mrc_tarx_c -n 1 -d 1 (zip {tz1:.2f} s, unzip {tu1:.2f} s)"},
               "sample": f"{nfiles} files x {sample_words.nbytes >> 20} MiB of the same N(10,3) volume, b={bits}: mrc_tarx_c -t zip -n {cores} -d 1 "
                         f"(wall {tz:.2f} s) then mrc_tarx_c -t unzip -n {cores} -d 1 on their containers (wall {tu:.2f} s); throughput mode, "
                         f"value = floats through both directions / total wall"}
        cli = None
        if os.path.exists(mine):
            try:
                nthr = 8
                _tarx(mine, lst1, "zip", outd, bits, 1)  # (the first start on a box pages the HIP runtime in)
                gz, gu = _tarx(mine, lst, "zip", outd, bits, nthr), _tarx(mine, zlst, "unzip", outd, bits, nthr)
                cli = {"tool": "datacompressionfloat_amd/bin/mrc_tarx", "mode": "-d 1 (throughput mode of the reference, mrc_tarx.c:226-231): files read, coded on the GPU, nothing written",
                       "threads": nthr, "files": nfiles, "file_MiB": sample_words.nbytes >> 20,
                       "compress_GBps": round(total / gz / 1e9, 3), "decompress_GBps": round(total / gu / 1e9, 3),
                       "value": round(total / (gz + gu) / 1e9, 3), "unit": "GB/s (whole command wall time, process start and HIP runtime start included)",
                       "vs_cpu_reference_same_files": round((tz + tu) / (gz + gu), 2)}
            except Exception as e:  # the bench line must not die on the front-end
                cli = {"error": str(e)[:200]}
        return cpu, cli


def parity_check(words_dev, rec, bits, nchunks):
    """Chunk records 0, the middle one and the last one of this run's container against the CPU oracle (oracle/mrcz_oracle.c),
    byte for byte.  The oracle is the checker here, outside every timed region."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import numpy as np
        import util
        o = util.load_oracle()
        offs, off = [], 0
        for _ in range(nchunks):
            h = struct.unpack("<4I", rec[off: off + 16].cpu().numpy().tobytes())
            offs.append(off)
            off += 16 + sum(x & 0x7fffffff for x in h)
        offs.append(off)
        picks = sorted({0, nchunks // 2, nchunks - 1})
        for k in picks:
            cw = words_dev[k * CHUNK: (k + 1) * CHUNK].cpu().numpy().view(np.uint32)
            if k == 0:
                ref = o.compress(cw, bits)[17:]
            else:  # not the file's first chunk: no header exemption (workers.c:777,804); coded behind an all-zero stand-in chunk
                z = o.compress(np.concatenate([np.zeros(CHUNK, np.uint32), cw]), bits)
                hh = struct.unpack("<4I", z[17:33])
                ref = z[17 + 16 + sum(x & 0x7fffffff for x in hh):]
            got = rec[offs[k]: offs[k + 1]].cpu().numpy().tobytes()
            if got != ref:
                return f"MISMATCH in chunk {k}"
        return f"chunk records {picks} of {nchunks} == oracle, byte for byte"
    except Exception as e:
        return f"not run: {str(e)[:120]}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bits", type=int, default=8)
    ap.add_argument("--gib-per-gpu", type=float, default=1.0)
    ap.add_argument("--stream", action="store_true", help="streaming mode (device-generated App. D volume, batches of 128 chunks); implied above 4 GiB")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    from datacompressionfloat_amd import MrcZipCodec, shard

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    streaming = args.stream or args.gib_per_gpu > 4.0

    # the whole job is ONE volume of world x G GiB; rank r owns the contiguous chunk range shard.float_range gives it
    total_floats = int(args.gib_per_gpu * world * (1 << 30)) // 4
    f_lo, f_hi, first_chunk = shard.float_range(rank, world, total_floats)
    nfloats = f_hi - f_lo
    nchunks = (nfloats + CHUNK - 1) // CHUNK
    batch_chunks = STREAM_BATCH_CHUNKS if streaming else min(128, nchunks)
    codec = MrcZipCodec(local, max_batch_chunks=batch_chunks)
    if streaming:
        words = torch.empty(nfloats, dtype=torch.int32, device=device)
        codec.generate_kat_device(words, f_lo)
    else:
        words = torch.from_numpy(make_volume(nfloats, 1234 + rank, first=(rank == 0)).view("int32")).to(device)
    bfl = batch_chunks * CHUNK if streaming else nfloats          # floats per codec call
    pipe = Pipeline(codec, words, args.bits, first_chunk, f_lo, bfl, batch_chunks, world, rank, dist, torch, shard, torch.cuda.synchronize)
    nbatch = pipe.nbatch

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    elapsed, tc, td = run_timed(pipe, args.steps, args.warmup, barrier)
    tt = torch.tensor([elapsed, tc, td], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed, tc, td = tt.tolist()

    # ---- bit-exact round trip check: one more, untimed, pass ----
    _, _, zbytes, ok, last = pipe.step(verify=True)
    gathered = pipe.finish_gather()
    assert ok, "round trip is not bit-exact"
    gather_note = None
    if world > 1 and rank == 0 and not streaming and gathered is not None and gathered[0] is not None:
        out_all, sizes = pipe.check_gathered(gathered, total_floats)
        exp = words.clone()
        codec.erase_bits_device(exp, args.bits, 0)
        assert torch.equal(out_all[:nfloats], exp), "rank 0's range of the gathered container differs from erasebytes(input)"
        gather_note = f"{sum(sizes)} record bytes gathered from {world} ranks decode as one container of {total_floats} floats"
        del out_all, exp

    if rank == 0:
        in_bytes_all = 4.0 * total_floats
        ms_step = elapsed / args.steps * 1e3
        value = in_bytes_all / (elapsed / args.steps) / 1e9
        # ---- roofline of the dominant kernel (HIP events on the codec's stream, per launch) ----
        sub = words[: min(nfloats, bfl)]
        codec.set_timing(True)
        reps = 3
        acc = {}
        for _ in range(reps):
            r2, pb2 = codec.compress_device(sub, args.bits, first_chunk, out=pipe.rec_bufs[0])
            for k, v in codec.last_timings().items():
                acc[k] = acc.get(k, 0.0) + v / reps
            codec.uncompress_device(r2, sub.numel(), out=pipe.out_buf[: sub.numel()])
            for k, v in codec.last_timings().items():
                acc[k] = acc.get(k, 0.0) + v / reps
        codec.set_timing(False)
        dom = max(acc, key=acc.get)
        # algorithmic bytes of one launch: what the kernel must read once + write once
        # planes stored verbatim (RAW, zip.c:184-190) never pass through the entropy kernels: leave them out of those kernels' bytes
        nsub, csub, zsub = sub.numel(), (sub.numel() + CHUNK - 1) // CHUNK, r2.numel()
        raw_planes = sum(1 for pb in pb2 if pb == nsub + 4 * csub)
        z_coded, n_coded = zsub - raw_planes * nsub, (4 - raw_planes) * nsub
        alg = {"k_tile_summary": 8.0 * nsub, "k_histogram": 4.0 * nsub, "k_emit": 4.0 * nsub + zsub,
               "k_blk_count": z_coded + n_coded, "k_merge_segments": 4.0 * nsub + n_coded + raw_planes * nsub,
               "k_inflate_par": z_coded + n_coded}.get(dom, 4.0 * nsub + zsub)
        achieved = alg / (acc[dom] * 1e-3) / 1e9
        # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
        # separate runs, summed over the launches of one pass and corrected per kernel as the file says); null if the
        # dominant kernel has no committed measurement
        traffic = None
        try:
            import glob
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[::-1]:
                k = json.load(open(f)).get("kernels", {}).get(dom)
                if k:
                    traffic = k.get("bytes_per_launch", k.get("corrected_bytes"))
                    break
        except Exception:
            traffic = None
        # the whole path against the HBM roof, (4N + Z) / t per direction (host-side wall time of the timed steps)
        z_all = float(zbytes) * world
        t_c, t_d = tc / args.steps, td / args.steps
        whole = {"definition": "(4N + Z) / t / HBM peak, N floats in, Z container payload bytes (SURVEY 8(d)); t = wall time of the timed steps per direction",
                 "compress": {"GBps": round((in_bytes_all + z_all) / t_c / 1e9, 1), "frac": round((in_bytes_all + z_all) / t_c / 1e9 / (HBM_PEAK_GBS * world), 4)},
                 "decompress": {"GBps": round((in_bytes_all + z_all) / t_d / 1e9, 1), "frac": round((in_bytes_all + z_all) / t_d / 1e9 / (HBM_PEAK_GBS * world), 4)},
                 "step": {"GBps": round(2.0 * (in_bytes_all + z_all) / (t_c + t_d) / 1e9, 1), "frac": round(2.0 * (in_bytes_all + z_all) / (t_c + t_d) / 1e9 / (HBM_PEAK_GBS * world), 4)},
                 "read_side_of_compress": {"GBps": round(in_bytes_all / t_c / 1e9, 1), "frac": round(in_bytes_all / t_c / 1e9 / (HBM_PEAK_GBS * world), 4),
                                           "north_star_target_frac": 0.70}}
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "algorithmic_bytes": int(alg),
                    "avg_launch_ms": round(acc[dom], 4), "whole_path": whole,
                    "kernel_ms": {k: round(v, 4) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])}}
        cpu = cli = None
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is timed at N = 1 only
            cores = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)))
            sample = words[: 16 * 1048576].cpu().numpy()  # first 64 MiB of the volume (3 chunks)
            cpu, cli = cpu_baseline_and_cli(sample, args.bits, cores)
            if not streaming:
                cpu["parity_check"] = parity_check(words, last[0], args.bits, nchunks)
        vol = (f"{args.gib_per_gpu:g} GiB of the SURVEY App. D integer-generator volume per GPU, generated on the device, coded in {nbatch} batches of "
               f"{batch_chunks} chunks" if streaming else f"{args.gib_per_gpu:g} GiB synthetic float32 volume per GPU (256-word header + N(10,3^2), numpy default_rng(1234 + rank))")
        line = {
            "metric": "compress + decompress GB/s (input floats), bit-exact round trip",
            "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{vol}, single mask level b={args.bits}, {nchunks} chunks/GPU, compress then decompress, HBM-resident",
                       "bits": args.bits, "chunks_per_gpu": nchunks, "compressed_bytes_per_gpu": int(zbytes),
                       "ratio": round(zbytes / (4.0 * nfloats), 4), "streaming": streaming,
                       "sharding": "contiguous chunk ranges of one volume per rank (shard.float_range); grouped RCCL send/recv of the records "
                                   "to rank 0, overlapped with the decompress of the same call and the compress of the next",
                       "gather_check": gather_note},
            "compress_GBps": round(in_bytes_all / (tc / args.steps) / 1e9, 3),
            "decompress_GBps": round(in_bytes_all / (td / args.steps) / 1e9, 3),
            "frac_of_hbm_peak": round(value / (HBM_PEAK_GBS * world), 5),
            "roofline": roofline, "cpu_baseline": cpu, "cli": cli,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
