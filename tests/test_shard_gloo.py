"""CPU-only, world_size 2 over gloo: the N>1 path (chunk ranges per rank + gather of the records on
rank 0) reproduces the single-process container byte for byte.  The per-rank codec here is the
SIMT-emulator build of the product kernels (tests/sim), the exchange code is the product's."""
import os
import subprocess
import sys
import tempfile

import numpy as np

import util

WORKER = r'''
import os, sys, pickle
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["REPO"]); sys.path.insert(0, os.path.join(os.environ["REPO"], "tests"))
import util
from datacompressionfloat_amd import shard
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["PORT"], rank=rank, world_size=world)
words = np.fromfile(os.environ["INPUT"], dtype=np.uint32)
lo, hi, first_chunk = shard.float_range(rank, world, len(words))
sim = util.load_sim()
rec = sim.compress_records(words[lo:hi], 8, first_chunk) if hi > lo else b""
t = torch.frombuffer(bytearray(rec), dtype=torch.uint8) if rec else torch.empty(0, dtype=torch.uint8)
full, sizes = shard.gather_records(t, dist, dst=0)
# the asynchronous form (what bench.py pipelines) must deliver the same bytes
h = shard.gather_records_start(t, dist, dst=0)
full2, sizes2 = h.wait()
assert sizes2 == sizes and (rank != 0 or torch.equal(full2, full))
if rank == 0:
    open(os.environ["OUTPUT"], "wb").write(full.numpy().tobytes())
    assert sizes[0] == len(rec) and len(sizes) == world
dist.barrier()
dist.destroy_process_group()
'''


def test_two_ranks_reproduce_the_single_process_container(oracle):
    util.load_sim()  # build once, before the ranks race for it
    n = util.CHUNK + 70000  # 2 chunks: rank 0 gets chunk 0 (with the header words), rank 1 the tail chunk
    words = util.gauss_words(n, seed=21)
    with tempfile.TemporaryDirectory() as d:
        inp, out, wk = os.path.join(d, "in.bin"), os.path.join(d, "out.bin"), os.path.join(d, "worker.py")
        words.tofile(inp)
        open(wk, "w").write(WORKER)
        env = dict(os.environ, REPO=util.ROOT, INPUT=inp, OUTPUT=out, WORLD_SIZE="2", PORT=str(29500 + os.getpid() % 2000))
        procs = [subprocess.Popen([sys.executable, wk], env=dict(env, RANK=str(r))) for r in range(2)]
        for p in procs:
            assert p.wait(timeout=900) == 0
        got = open(out, "rb").read()
    assert got == oracle.compress(words.tobytes(), 8)[17:]


def test_chunk_ranges_partition_the_file():
    from datacompressionfloat_amd import shard
    for nchunks in (1, 2, 7, 43, 2731):
        for world in (1, 2, 4, 8):
            spans = [shard.chunk_range(r, world, nchunks) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == nchunks
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
