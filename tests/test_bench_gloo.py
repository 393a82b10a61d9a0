"""CPU-only, world_size 2 over gloo: the REAL N > 1 sequence of bench.py (class Pipeline: two record buffers, the gather of call i
started before the decompress of call i and finished before call i + 1 reuses its buffers, finish_gather before the clock
stops, rank 0's decode of the gathered container) with the SIMT-emulator build of the product kernels as the per-rank codec.
What the 8-GPU run of the driver executes is this loop with the HIP codec and RCCL; the hardware scaling curve itself is
still unmeasured."""
import os
import subprocess
import sys
import tempfile

import numpy as np

import util

WORKER = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["REPO"]); sys.path.insert(0, os.path.join(os.environ["REPO"], "tests"))
import util
import bench
from datacompressionfloat_amd import shard

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["PORT"], rank=rank, world_size=world)


class SimTorchCodec:
    """the emulator behind the method names bench.Pipeline calls on MrcZipCodec (tensors live in host memory)"""
    def __init__(self):
        self.sim = util.load_sim()
    def records_bound(self, n):
        return int(self.sim.lib.mrcz_records_bound(n))
    def compress_device(self, words, bits, first_chunk=0, out=None):
        rec = self.sim.compress_records(words.numpy().view(np.uint32), bits, first_chunk)
        out[: len(rec)] = torch.frombuffer(bytearray(rec), dtype=torch.uint8)
        return out[: len(rec)], [0, 0, 0, 0]
    def uncompress_device(self, rec, nfloats, out=None):
        w = self.sim.uncompress_records(rec.numpy().tobytes(), nfloats)
        out[:nfloats] = torch.from_numpy(w.view(np.int32))
        return out[:nfloats], rec.numel()
    def erase_bits_device(self, t, bits, first_word_index=0):
        a = t.numpy().view(np.uint32)
        m = np.uint32(0) if bits >= 32 else np.uint32((0xFFFFFFFF << bits) & 0xFFFFFFFF)
        lo = max(0, 256 - first_word_index)            # the first 256 words of the FILE keep their bits
        a[lo:] &= m
        return t


allw = np.fromfile(os.environ["INPUT"], dtype=np.uint32)
total = len(allw)
f_lo, f_hi, first_chunk = shard.float_range(rank, world, total)
words = torch.from_numpy(allw[f_lo:f_hi].view(np.int32).copy())
nchunks = (f_hi - f_lo + util.CHUNK - 1) // util.CHUNK
codec = SimTorchCodec()
pipe = bench.Pipeline(codec, words, 8, first_chunk, f_lo, f_hi - f_lo, nchunks, world, rank, dist, torch, shard, lambda: None)
assert len(pipe.rec_bufs) == 2 and (pipe.gather_buf is not None) == (rank == 0)
elapsed, tc, td = bench.run_timed(pipe, 2, 1, dist.barrier)      # 1 warm-up + 2 timed steps: three gathers in flight one after the other
assert pipe.pending is None and pipe.ncall == 3
_, _, zbytes, ok, last = pipe.step(verify=True)
gathered = pipe.finish_gather()
assert ok
if rank == 0:
    out_all, sizes = pipe.check_gathered(gathered, total)
    assert len(sizes) == world and sizes[0] == zbytes
    open(os.environ["OUTPUT"], "wb").write(gathered[0].numpy().tobytes())
    open(os.environ["OUTPUT"] + ".dec", "wb").write(out_all.numpy().tobytes())
else:
    assert gathered[0] is None
dist.barrier()
dist.destroy_process_group()
'''


def test_bench_step_loop_over_two_ranks(oracle):
    util.load_sim()  # build once, before the ranks race for it
    n = 3 * util.CHUNK + 50001                      # four chunks: rank 0 codes chunks 0-1, rank 1 chunks 2-3 (the last one partial)
    words = np.zeros(n, np.uint32)                  # mostly zero planes keep the emulator fast; every chunk has a noisy stretch
    words[:256] = util.kat_words(256)
    for c in range(4):
        a = c * util.CHUNK + 700 * (c + 1)
        words[a: a + 4000] = util.gauss_words(4000, seed=40 + c, header=False)
    with tempfile.TemporaryDirectory() as d:
        inp, out, wk = os.path.join(d, "in.bin"), os.path.join(d, "out.bin"), os.path.join(d, "worker.py")
        words.tofile(inp)
        open(wk, "w").write(WORKER)
        env = dict(os.environ, REPO=util.ROOT, INPUT=inp, OUTPUT=out, WORLD_SIZE="2", PORT=str(31500 + os.getpid() % 2000))
        procs = [subprocess.Popen([sys.executable, wk], env=dict(env, RANK=str(r))) for r in range(2)]
        for p in procs:
            assert p.wait(timeout=1500) == 0
        got = open(out, "rb").read()
        dec = np.fromfile(out + ".dec", np.uint32)
    assert got == oracle.compress(words.tobytes(), 8, threads=4)[17:]     # the gathered records ARE the single-process container
    assert np.array_equal(dec, util.erase_expected(words, 8))             # and rank 0's decode of them is erasebytes(whole volume)
