#!/usr/bin/env python3
"""bench.py -- headline benchmark: compress + decompress GB/s of input floats on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input that is already resident in
HBM: compress the rank's volume (mask + byte planes + DEFLATE Z_RLE -> chunk records) and decompress
it again.  N = 1 workload: BASELINE.json configs[1] -- 1 GiB synthetic float32 volume
(256 header words + N(10, 3^2)), single mask level b = 8, 43 chunks = 172 plane streams.
N > 1: weak scaling -- every rank owns a 1 GiB range of chunks of an N GiB volume (chunks are
independent); the only exchange is the final concatenation gather of the compressed
records to rank 0 over RCCL, which is inside the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement), with two extra objects:
  "roofline"      dominant kernel vs the HBM roof (HIP events on the codec's own stream)
  "cpu_baseline"  the reference's own pthread path (oracle/_ref) timed on this box's host cores
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


def make_volume(torch, nfloats, seed, device, first):
    """N(10, 3^2) float32, first 256 words of the FILE are a header."""
    g = torch.Generator(device=device).manual_seed(seed)
    x = torch.empty(nfloats, dtype=torch.float32, device=device).normal_(10.0, 3.0, generator=g)
    w = x.view(torch.int32)
    if first:
        w[:256] = 0
        w[0], w[1], w[2], w[3] = 4096, 4096, nfloats // (4096 * 4096), 2
    return w


def cpu_baseline(sample_words, bits, cores):
    """Time the reference's own file-level pthread pool (src/main/mrc_tarx.c:134-176, built as
    oracle/_ref/mrc_tarx_c) in throughput mode (-d 1: no output writes) on a bounded sample."""
    ref = os.path.join(ROOT, "oracle", "_ref", "mrc_tarx_c")
    if os.path.exists(ref):
        nfiles = max(cores, 8)
        with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
            src = os.path.join(d, "sample0.mrc")
            sample_words.tofile(src)
            names = [src]
            for i in range(1, nfiles):
                p = os.path.join(d, f"sample{i}.mrc")
                os.link(src, p)
                names.append(p)
            lst = os.path.join(d, "files.txt")
            open(lst, "w").write("\n".join(names) + "\n")
            outd = os.path.join(d, "out")
            os.mkdir(outd)
            t0 = time.time()
            subprocess.run([ref, "-i", lst, "-t", "zip", "-o", outd, "-b", str(bits), "-n", str(cores), "-d", "1"],
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            dt = time.time() - t0
            total = nfiles * sample_words.nbytes
            return {"value": round(total / dt / 1e9, 4), "unit": "GB/s", "cores": cores, "kind": "reference",
                    "sample": f"compress only: {nfiles} files x {sample_words.nbytes >> 20} MiB of the same N(10,3) volume, b={bits}, "
                              f"mrc_tarx_c -n {cores} -d 1 (throughput mode), wall {dt:.2f} s"}
    # fall back to the in-repo restatement (still a CPU baseline, never the measured product)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import util
    o = util.load_oracle()
    t0 = time.time()
    o.compress(sample_words.tobytes(), bits, threads=cores)
    dt = time.time() - t0
    return {"value": round(sample_words.nbytes / dt / 1e9, 4), "unit": "GB/s", "cores": cores, "kind": "port",
            "sample": f"compress only: {sample_words.nbytes >> 20} MiB, chunk-parallel pthread oracle, wall {dt:.2f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bits", type=int, default=8)
    ap.add_argument("--gib-per-gpu", type=float, default=1.0)
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

    nchunks = max(1, int(round(args.gib_per_gpu * (1 << 30) / (4 * CHUNK) + 0.49)))  # 1 GiB -> 43 chunks
    nfloats = int(args.gib_per_gpu * (1 << 30)) // 4
    first_chunk = rank * nchunks                      # this rank's chunk range inside the N GiB volume
    words = make_volume(torch, nfloats, 1234 + rank, device, first=(rank == 0))
    codec = MrcZipCodec(local, max_batch_chunks=min(128, nchunks))
    cap = codec.records_bound(nfloats)
    # two record buffers: the gather of step i (RCCL, its own stream) runs under the decompress of step i and the compress
    # of step i + 1, which writes the other buffer; a gather is always finished before the next one starts
    rec_bufs = [torch.empty(cap, dtype=torch.uint8, device=device) for _ in range(2 if world > 1 else 1)]
    out_buf = torch.empty(nfloats, dtype=torch.int32, device=device)
    gather_buf = torch.empty(int(cap * world * 0.75) + 1024, dtype=torch.uint8, device=device) if (world > 1 and rank == 0) else None
    pending = [None]
    nstep = [0]

    def finish_gather():
        """the previous step's concatenation must be complete (on the device) before its buffers are reused"""
        if pending[0] is not None:
            pending[0].wait()
            torch.cuda.synchronize()
            pending[0] = None

    def step():
        """compress this rank's chunk range, start the concatenation gather of the records on rank 0 (sizes via
        all_gather, records via grouped RCCL send/recv: datacompressionfloat_amd/shard.py), decompress"""
        rec_buf = rec_bufs[nstep[0] % len(rec_bufs)]
        nstep[0] += 1
        a = time.perf_counter()
        rec, _ = codec.compress_device(words, args.bits, first_chunk, out=rec_buf)
        b = time.perf_counter()
        if world > 1:
            finish_gather()
            pending[0] = shard.gather_records_start(rec, dist, dst=0, out=gather_buf)
        out, _ = codec.uncompress_device(rec, nfloats, out=out_buf)
        c = time.perf_counter()
        return rec, out, b - a, c - b

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
        rec, out, dc, dd = step()
        tc += dc
        td += dd
    finish_gather()  # the last step's records have arrived on rank 0 before the clock stops
    barrier()
    elapsed = time.perf_counter() - t0
    tt = torch.tensor([elapsed, tc, td], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed, tc, td = tt.tolist()

    # ---- bit-exact round trip check of the last step (outside the timed region) ----
    exp = words.clone()
    codec.erase_bits_device(exp, args.bits, first_chunk * CHUNK)
    assert torch.equal(out, exp), "round trip is not bit-exact"
    zbytes = rec.numel()

    if rank == 0:
        in_bytes_all = 4.0 * nfloats * world
        ms_step = elapsed / args.steps * 1e3
        value = in_bytes_all / (elapsed / args.steps) / 1e9
        # ---- roofline of the dominant kernel (HIP events on the codec's stream, per launch) ----
        codec.set_timing(True)
        reps = 3
        acc = {}
        for _ in range(reps):
            codec.compress_device(words, args.bits, first_chunk, out=rec_bufs[0])
            for k, v in codec.last_timings().items():
                acc[k] = acc.get(k, 0.0) + v / reps
            codec.uncompress_device(rec, nfloats, out=out_buf)
            for k, v in codec.last_timings().items():
                acc[k] = acc.get(k, 0.0) + v / reps
        codec.set_timing(False)
        dom = max(acc, key=acc.get)
        # algorithmic bytes of one launch: what the kernel must read once + write once
        # planes stored verbatim (RAW, zip.c:184-190) never pass through the entropy kernels: leave them out of those kernels' bytes
        _, plane_bytes = codec.compress_device(words, args.bits, first_chunk, out=rec_bufs[0])
        raw_planes = sum(1 for pb in plane_bytes if pb == nfloats + 4 * nchunks)
        z_coded, n_coded = zbytes - raw_planes * nfloats, (4 - raw_planes) * nfloats
        alg = {"k_tile_summary": 8.0 * nfloats, "k_histogram": 4.0 * nfloats, "k_emit": 4.0 * nfloats + zbytes,
               "k_blk_count": z_coded + n_coded, "k_blk_gather": 2.0 * n_coded, "k_inflate_par": z_coded + n_coded,
               "k_merge_planes": 8.0 * nfloats}.get(dom, 4.0 * nfloats + zbytes)
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
        line = {
            "metric": "compress + decompress GB/s (input floats), bit-exact round trip",
            "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.gib_per_gpu:g} GiB synthetic float32 volume per GPU (256-word header + N(10,3^2)), "
                                   f"single mask level b={args.bits}, {nchunks} chunks/GPU, compress then decompress, HBM-resident",
                       "bits": args.bits, "chunks_per_gpu": nchunks, "compressed_bytes_per_gpu": int(zbytes),
                       "ratio": round(zbytes / (4.0 * nfloats), 4),
                       "sharding": "contiguous chunk ranges per rank; grouped RCCL send/recv of the records to rank 0, "
                                   "overlapped with the decompress of the same step and the compress of the next"},
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
