"""GPU, two processes on the ONE device of the box: the N > 1 path of bench.py (chunk ranges per rank, first_chunk offsets, records
gathered on rank 0, rank 0 decodes the concatenation) with the real HIP codec in every rank.  The exchange runs over gloo on host
copies of the records -- RCCL refuses two ranks on one device --, so what this adds to tests/test_bench_gloo.py (the same loop on the
emulator) is the product library under two concurrent processes and its handling of a shard that does not begin at chunk 0.  The
RCCL transport itself and the scaling curve still need a multi-GPU node."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["REPO"]); sys.path.insert(0, os.path.join(os.environ["REPO"], "tests"))
import util
from datacompressionfloat_amd import MrcZipCodec, shard

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["PORT"], rank=rank, world_size=world)
allw = np.fromfile(os.environ["INPUT"], dtype=np.uint32)
total = len(allw)
f_lo, f_hi, first_chunk = shard.float_range(rank, world, total)
assert rank == 0 or first_chunk > 0
dev = torch.device("cuda", 0)
words = torch.from_numpy(allw[f_lo:f_hi].view(np.int32).copy()).to(dev)
nchunks = (f_hi - f_lo + util.CHUNK - 1) // util.CHUNK
codec = MrcZipCodec(0, max_batch_chunks=nchunks)
for it in range(3):                                   # a few calls back to back, both ranks on the device at once
    rec, planes = codec.compress_device(words, 8, first_chunk)
    out, used = codec.uncompress_device(rec, words.numel())
    assert used == rec.numel() and codec.last_fallbacks() == 0
    exp = words.clone()
    codec.erase_bits_device(exp, 8, f_lo)
    assert torch.equal(out, exp), (rank, it)
host = rec.cpu()
h = shard.gather_records_start(host, dist, dst=0)     # the asynchronous form bench.Pipeline uses
cat, sizes = h.wait()
if rank == 0:
    assert len(sizes) == world and sizes[0] == rec.numel()
    open(os.environ["OUTPUT"], "wb").write(cat.numpy().tobytes())
    big = MrcZipCodec(0, max_batch_chunks=(total + util.CHUNK - 1) // util.CHUNK)
    dec, used = big.uncompress_device(cat.to(dev), total)   # the gathered records are ONE container of the whole volume
    assert used == cat.numel()
    open(os.environ["OUTPUT"] + ".dec", "wb").write(dec.cpu().numpy().tobytes())
    big.close()
codec.close()
dist.barrier()
dist.destroy_process_group()
'''


def test_two_ranks_share_the_device(oracle):
    n = 5 * util.CHUNK + 123457                      # six chunks: three per rank, the last one partial
    words = util.gauss_words(n, seed=77)
    with tempfile.TemporaryDirectory() as d:
        inp, out, wk = os.path.join(d, "in.bin"), os.path.join(d, "out.bin"), os.path.join(d, "worker.py")
        words.tofile(inp)
        open(wk, "w").write(WORKER)
        env = dict(os.environ, REPO=util.ROOT, INPUT=inp, OUTPUT=out, WORLD_SIZE="2", PORT=str(30500 + os.getpid() % 2000))
        procs = [subprocess.Popen([sys.executable, wk], env=dict(env, RANK=str(r))) for r in range(2)]
        for p in procs:
            assert p.wait(timeout=900) == 0
        got = open(out, "rb").read()
        dec = np.fromfile(out + ".dec", np.uint32)
    assert got == oracle.compress(words.tobytes(), 8, threads=8)[17:]     # byte for byte the single-process container
    assert np.array_equal(dec, util.erase_expected(words, 8))
