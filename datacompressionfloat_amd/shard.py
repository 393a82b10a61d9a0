"""Chunk sharding across ranks.

Every (chunk, plane) stream is independent, so a file is cut into contiguous chunk ranges, one per
rank (one process per GPU).  The only exchange is the final concatenation of the compressed chunk
records on the writer rank: an all-gather of the per-rank byte counts, then point-to-point sends of the
variable-size records to rank 0 (on MI355X every peer has its own xGMI link to the writer, so direct
send/recv beats a ring).  Works with any torch.distributed backend ("nccl" = RCCL on ROCm; "gloo" in the
CPU tests).  This module never imports the codec: it only moves bytes.
"""
import torch

CHUNK_FLOATS = 6 * 1048576


def chunk_range(rank: int, world: int, nchunks: int):
    """Contiguous, balanced chunk range [lo, hi) of `rank`; the first (nchunks % world) ranks get one more."""
    base, extra = divmod(nchunks, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def float_range(rank: int, world: int, nfloats: int):
    """[lo, hi) in floats of the rank's chunk range of a file with `nfloats` floats."""
    nchunks = (nfloats + CHUNK_FLOATS - 1) // CHUNK_FLOATS
    clo, chi = chunk_range(rank, world, nchunks)
    return min(clo * CHUNK_FLOATS, nfloats), min(chi * CHUNK_FLOATS, nfloats), clo


class PendingGather:
    """Handle of a gather started with gather_records_start(): wait() returns (concatenation on dst else None, sizes)."""

    def __init__(self, works, result, sizes):
        self._works, self._result, self.sizes = works, result, sizes

    def wait(self):
        for w in self._works:
            w.wait()
        self._works = []
        return self._result, self.sizes


def gather_records_start(records: torch.Tensor, dist=None, dst: int = 0, out: torch.Tensor = None) -> PendingGather:
    """Start concatenating every rank's chunk records on `dst` in rank (= file) order and return at once.

    records: 1-D uint8 tensor (cuda for nccl, cpu for gloo); it must stay untouched until wait().  The sizes travel
    in one small all-gather; the records themselves go point to point to the writer, all peers in ONE grouped
    operation (batch_isend_irecv) so that on MI355X the seven xGMI links into the writer run concurrently instead
    of one receive after the other.  `out` may be a preallocated buffer on dst."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return PendingGather([], records, [int(records.numel())])
    rank, world = dist.get_rank(), dist.get_world_size()
    sizes = torch.zeros(world, dtype=torch.int64, device=records.device)
    mine = torch.tensor([records.numel()], dtype=torch.int64, device=records.device)
    dist.all_gather_into_tensor(sizes, mine)
    sz = [int(x) for x in sizes.tolist()]
    if rank != dst:
        works = dist.batch_isend_irecv([dist.P2POp(dist.isend, records, dst)]) if records.numel() else []
        return PendingGather(works, None, sz)
    total = sum(sz)
    if out is None or out.numel() < total:
        out = torch.empty(total, dtype=torch.uint8, device=records.device)
    off, ops = 0, []
    for r in range(world):
        if r == dst:
            out[off: off + sz[r]].copy_(records)
        elif sz[r]:
            ops.append(dist.P2POp(dist.irecv, out[off: off + sz[r]], r))
        off += sz[r]
    works = dist.batch_isend_irecv(ops) if ops else []
    return PendingGather(works, out[:total], sz)


def gather_records(records: torch.Tensor, dist=None, dst: int = 0, out: torch.Tensor = None):
    """Blocking form of gather_records_start(): returns (tensor view of the concatenation on dst, else None; list of
    per-rank sizes)."""
    return gather_records_start(records, dist, dst, out).wait()
