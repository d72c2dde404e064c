"""Multi-GPU layout of the hot path: streams are independent units (per-stream LSTM state, read-only weights), so
they are partitioned into contiguous blocks, one block per rank / GPU, with NO collective in the forward pass.  The
only exchange is the final gather of the per-chunk speech probabilities to rank 0 (RCCL `gather` on GPUs; the same
code runs over gloo on CPU in tests and in `bench.py --dry-run`): 4 bytes per chunk -- the speech probability alone."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def stream_block(rank: int, world: int, total_streams: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of global stream ids owned by `rank`; sizes differ by at most one."""
    if not (0 <= rank < world) or total_streams < 0:
        raise ValueError("bad rank/world/total_streams")
    base, extra = divmod(total_streams, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def owner_of(stream: int, world: int, total_streams: int) -> int:
    for r in range(world):
        lo, hi = stream_block(r, world, total_streams)
        if lo <= stream < hi:
            return r
    raise ValueError("stream out of range")


class ProbabilityGather:
    """The path's one collective: per step, every rank's probabilities -> rank `dst`.

    slot = 1 (default): the speech probability alone -- element 1 of the engine's [streams_of_rank, chunks, 2] output (vadc.c:704-713: the host
    reads `output[i * stride + 1]`), 4 B per chunk on the wire, `result()` = [total_streams, chunks].  slot = None: both elements as the engine
    wrote them, 8 B per chunk, `result()` = [total_streams, chunks, 2].
    Block sizes follow from `stream_block`, so no size exchange is needed; equal blocks are ONE `dist.gather` into
    preallocated buffers (what bench.py issues inside its timed region), ragged blocks are padded to the largest.
    `gather()` only enqueues (on the current stream for RCCL); `result()` assembles the whole job's tensor on `dst`."""

    def __init__(self, total_streams: int, chunks: int, device, dtype=torch.float32, dst: int = 0, slot: Optional[int] = 1):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.dst = dst
        self.slot = slot
        self.sizes = [hi - lo for lo, hi in (stream_block(r, self.world, total_streams) for r in range(self.world))]
        self.lo, self.hi = stream_block(self.rank, self.world, total_streams)
        self.mx = max(self.sizes) if self.sizes else 0
        self.ragged = any(s != self.mx for s in self.sizes)
        shape = (self.mx, chunks) if slot is not None else (self.mx, chunks, 2)
        # the send buffer: the slot's strided view is made contiguous into it (one small copy kernel on the gather's stream); ragged blocks are padded in it
        self.send = torch.zeros(shape, dtype=dtype, device=device) if (self.ragged or slot is not None) else None
        self.bufs: Optional[List[torch.Tensor]] = None
        if self.world > 1 and self.rank == dst:
            self.bufs = [torch.empty(shape, dtype=dtype, device=device) for _ in range(self.world)]
        self.bytes_per_chunk = 4 if slot is not None else 8

    def gather(self, local: torch.Tensor) -> None:
        if local.shape[0] != self.hi - self.lo:
            raise ValueError("local block does not match this rank's stream block")
        if self.slot is not None:
            local = local[:, :, self.slot]
        if self.world == 1:
            self._single = local
            return
        send = local
        if self.send is not None:
            self.send[: local.shape[0]].copy_(local)
            send = self.send
        dist.gather(send.contiguous(), self.bufs if self.rank == self.dst else None, dst=self.dst)

    def result(self) -> Optional[torch.Tensor]:
        if self.world == 1:
            return self._single
        if self.rank != self.dst:
            return None
        return torch.cat([b[:n] for b, n in zip(self.bufs, self.sizes)], dim=0)


def gather_probabilities(local: torch.Tensor, total_streams: int, dst: int = 0, slot: Optional[int] = 1):
    """One-shot form: returns the [total_streams, chunks] speech probabilities (slot = None: [total_streams, chunks, 2]) on rank `dst` (None elsewhere)."""
    g = ProbabilityGather(total_streams, local.shape[1], local.device, local.dtype, dst, slot)
    g.gather(local)
    return g.result()
