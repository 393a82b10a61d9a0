#!/usr/bin/env python3
"""bench.py -- headline benchmark: compress + decompress GB/s of input floats on MI355X.

One "step" = one pass of the hot path over synthetic input that is already resident in HBM: compress the
rank's volume (mask + byte planes + DEFLATE Z_RLE -> chunk records) and decompress it again.

  N = 1 (default)   BASELINE.json configs[1]: 1 GiB synthetic float32 volume (256 header words + N(10, 3^2)),
                    single mask level b = 8, 43 chunks = 172 plane streams, one codec call each way.
  --gib-per-gpu G   G > 4 (or --stream): streaming mode for the large configurations (configs[3]: 64 GiB).  The
                    volume is generated ON the device by the SURVEY App. D integer generator and stays resident
                    (only the input); it is coded in batches of 128 chunks (3 GiB), each batch compressed,
                    decompressed and -- in one extra, untimed pass -- compared with erasebytes(input).
  N > 1             weak scaling: every rank owns a contiguous chunk range of the N x G GiB volume
                    (datacompressionfloat_amd/shard.py float_range: chunks are independent); the only
                    exchange is the concatenation gather of the compressed records to rank 0 over RCCL, inside the
                    timed region; rank 0 decodes the gathered container once, outside it.

Prints ONE JSON line on rank 0 (contract in the task statement), with two extra objects:
  "roofline"      dominant kernel vs the HBM roof (HIP events on the codec's own stream)
  "cpu_baseline"  the reference's own pthread path (oracle/_ref) timed on this box's host cores, both directions
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)
CHUNK = 6 * 1048576
STREAM_BATCH_CHUNKS = 128


def make_volume(torch, nfloats, seed, device, first):
    """N(10, 3^2) float32, first 256 words of the FILE are a header."""
    g = torch.Generator(device=device).manual_seed(seed)
    x = torch.empty(nfloats, dtype=torch.float32, device=device).normal_(10.0, 3.0, generator=g)
    w = x.view(torch.int32)
    if first:
        w[:256] = 0
        w[0], w[1], w[2], w[3] = 4096, 4096, max(1, nfloats // (4096 * 4096)), 2
    return w


def cpu_baseline(sample_words, bits, cores):
    """Time the reference's own file-level pthread pool (src/main/mrc_tarx.c:134-176, built as oracle/_ref/mrc_tarx_c) in
    throughput mode (-d 1: no output writes) on a bounded sample: zip of N copies of the sample, then unzip of N copies of
    its container.  value = bytes of floats through both directions / total wall time (the metric's definition)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "mrc_tarx_c")
    ref1 = os.path.join(ROOT, "oracle", "_ref", "mrc_tar_c")
    if os.path.exists(ref) and os.path.exists(ref1):
        nfiles = max(cores, 8)
        with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
            src = os.path.join(d, "sample0.mrc")
            sample_words.tofile(src)
            zsrc = os.path.join(d, "sample0.zip")
            subprocess.run([ref1, "-i", src, "-o", zsrc, "-b", str(bits), "-t", "zip"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            names, znames = [src], [zsrc]
            for i in range(1, nfiles):
                p, z = os.path.join(d, f"sample{i}.mrc"), os.path.join(d, f"sample{i}.zip")
                os.link(src, p)
                os.link(zsrc, z)
                names.append(p)
                znames.append(z)
            lst, zlst = os.path.join(d, "files.txt"), os.path.join(d, "zips.txt")
            open(lst, "w").write("\n".join(names) + "\n")
            open(zlst, "w").write("\n".join(znames) + "\n")
            outd = os.path.join(d, "out")
            os.mkdir(outd)
            t0 = time.time()
            subprocess.run([ref, "-i", lst, "-t", "zip", "-o", outd, "-b", str(bits), "-n", str(cores), "-d", "1"],
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            tz = time.time() - t0
            t0 = time.time()
            subprocess.run([ref, "-i", zlst, "-t", "unzip", "-o", outd, "-n", str(cores), "-d", "1"],
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            tu = time.time() - t0
            total = nfiles * sample_words.nbytes
            return {"value": round(total / (tz + tu) / 1e9, 4), "unit": "GB/s", "cores": cores, "kind": "reference",
                    "compress_GBps": round(total / tz / 1e9, 4), "decompress_GBps": round(total / tu / 1e9, 4),
                    "sample": f"{nfiles} files x {sample_words.nbytes >> 20} MiB of the same N(10,3) volume, b={bits}: mrc_tarx_c -t zip -n {cores} -d 1 "
                              f"(wall {tz:.2f} s) then mrc_tarx_c -t unzip -n {cores} -d 1 on their containers (wall {tu:.2f} s); throughput mode, "
                              f"value = floats through both directions / total wall"}
    # fall back to the in-repo restatement (still a CPU baseline, never the measured product)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import util
    o = util.load_oracle()
    t0 = time.time()
    z = o.compress(sample_words.tobytes(), bits, threads=cores)
    tz = time.time() - t0
    t0 = time.time()
    o.uncompress(z)
    tu = time.time() - t0
    return {"value": round(sample_words.nbytes / (tz + tu) / 1e9, 4), "unit": "GB/s", "cores": cores, "kind": "port",
            "sample": f"{sample_words.nbytes >> 20} MiB, chunk-parallel pthread oracle compress (wall {tz:.2f} s) + 1-thread oracle uncompress (wall {tu:.2f} s)"}


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
        words = make_volume(torch, nfloats, 1234 + rank, device, first=(rank == 0))
    bfl = batch_chunks * CHUNK if streaming else nfloats          # floats per codec call
    nbatch = (nfloats + bfl - 1) // bfl
    cap = codec.records_bound(min(bfl, nfloats))
    # two record buffers: the gather of call i (RCCL, its own stream) runs under the decompress of call i and the compress
    # of call i + 1, which writes the other buffer; a gather is always finished before the next one starts
    rec_bufs = [torch.empty(cap, dtype=torch.uint8, device=device) for _ in range(2 if world > 1 else 1)]
    out_buf = torch.empty(min(bfl, nfloats), dtype=torch.int32, device=device)
    gather_buf = torch.empty(int(cap * world * 0.75) + 1024, dtype=torch.uint8, device=device) if (world > 1 and rank == 0) else None
    pending = [None]
    ncall = [0]

    def finish_gather():
        """the previous call's concatenation must be complete (on the device) before its buffers are reused"""
        if pending[0] is not None:
            r = pending[0].wait()
            torch.cuda.synchronize()
            pending[0] = None
            return r
        return None

    def step(verify=False):
        """one pass over the rank's volume: per batch, compress the chunk range, start the concatenation gather of its records
        on rank 0 (sizes via all_gather, records via grouped RCCL send/recv: datacompressionfloat_amd/shard.py), decompress"""
        tc = td = 0.0
        zbytes = 0
        ok = True
        last = None
        for b in range(nbatch):
            sub = words[b * bfl: min(nfloats, (b + 1) * bfl)]
            rec_buf = rec_bufs[ncall[0] % len(rec_bufs)]
            ncall[0] += 1
            t0 = time.perf_counter()
            rec, planes = codec.compress_device(sub, args.bits, first_chunk + b * batch_chunks, out=rec_buf)
            t1 = time.perf_counter()
            if world > 1:
                finish_gather()
                pending[0] = shard.gather_records_start(rec, dist, dst=0, out=gather_buf)
            out, _ = codec.uncompress_device(rec, sub.numel(), out=out_buf[: sub.numel()])
            t2 = time.perf_counter()
            tc += t1 - t0
            td += t2 - t1
            zbytes += rec.numel()
            last = (rec, planes)
            if verify:  # bit-exact round trip == erasebytes(input), batch by batch (outside the timed region)
                exp = sub.clone()
                codec.erase_bits_device(exp, args.bits, f_lo + b * bfl)
                ok = ok and bool(torch.equal(out, exp))
                del exp
        return tc, td, zbytes, ok, last

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    finish_gather()
    barrier()
    t0 = time.perf_counter()
    tc = td = 0.0
    for _ in range(args.steps):
        c, d, zbytes, _, last = step()
        tc += c
        td += d
    finish_gather()  # the last records have arrived on rank 0 before the clock stops
    barrier()
    elapsed = time.perf_counter() - t0
    tt = torch.tensor([elapsed, tc, td], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed, tc, td = tt.tolist()

    # ---- bit-exact round trip check: one more, untimed, pass ----
    _, _, zbytes, ok, last = step(verify=True)
    gathered = finish_gather()
    assert ok, "round trip is not bit-exact"
    gather_note = None
    if world > 1 and rank == 0 and not streaming and gathered is not None and gathered[0] is not None:
        # the concatenation of all ranks' records (rank order = file order) is the chunk-record part of ONE container of the
        # whole volume: decode all of it here; rank 0's own range must come back as erasebytes(its input)
        cat, sizes = gathered
        dec = torch.empty(total_floats, dtype=torch.int32, device=device)
        out_all, used = codec.uncompress_device(cat, total_floats, out=dec)
        assert used == sum(sizes), "the gathered records are not one decodable container"
        exp = words.clone()
        codec.erase_bits_device(exp, args.bits, 0)
        assert torch.equal(out_all[:nfloats], exp), "rank 0's range of the gathered container differs from erasebytes(input)"
        gather_note = f"{sum(sizes)} record bytes gathered from {world} ranks decode as one container of {total_floats} floats"
        del dec, exp

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
            r2, pb2 = codec.compress_device(sub, args.bits, first_chunk, out=rec_bufs[0])
            for k, v in codec.last_timings().items():
                acc[k] = acc.get(k, 0.0) + v / reps
            codec.uncompress_device(r2, sub.numel(), out=out_buf[: sub.numel()])
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
        # separate runs of this same command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); null if the
        # dominant kernel has no committed measurement
        traffic = None
        try:
            import glob
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[::-1]:
                k = json.load(open(f)).get("kernels", {}).get(dom)
                if k:
                    traffic = k["corrected_bytes"]
                    break
        except Exception:
            traffic = None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "algorithmic_bytes": int(alg),
                    "avg_launch_ms": round(acc[dom], 4),
                    "kernel_ms": {k: round(v, 4) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])}}
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is timed at N = 1 only
            cores = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)))
            sample = words[: 16 * 1048576].cpu().numpy()  # first 64 MiB of the volume (3 chunks)
            cpu = cpu_baseline(sample, args.bits, cores)
        vol = (f"{args.gib_per_gpu:g} GiB of the SURVEY App. D integer-generator volume per GPU, generated on the device, coded in {nbatch} batches of "
               f"{batch_chunks} chunks" if streaming else f"{args.gib_per_gpu:g} GiB synthetic float32 volume per GPU (256-word header + N(10,3^2))")
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
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
