"""Multi-GPU layout of the hot path: streams are independent units (per-stream LSTM state, read-only weights), so
they are partitioned into contiguous blocks, one block per rank / GPU, with NO collective in the forward pass.  The
only exchange is the final gather of the per-chunk speech probabilities to rank 0 (RCCL `gather` on GPUs; the same
code runs over gloo on CPU in tests)."""
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


def gather_probabilities(local: torch.Tensor, dst: int = 0, bufs: Optional[List[torch.Tensor]] = None):
    """local: [streams_of_this_rank, chunks, 2].  Returns the [total_streams, chunks, 2] tensor on rank `dst`
    (None elsewhere).  Equal block sizes use one `gather`; ragged blocks are padded to the largest block."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return local
    rank = dist.get_rank()
    n_local = torch.tensor([local.shape[0]], device=local.device, dtype=torch.int64)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    send = local
    if local.shape[0] != mx:
        send = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    if rank == dst:
        if bufs is None:
            bufs = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send.contiguous(), bufs, dst=dst)
        return torch.cat([b[:n] for b, n in zip(bufs, sizes)], dim=0)
    dist.gather(send.contiguous(), None, dst=dst)
    return None
