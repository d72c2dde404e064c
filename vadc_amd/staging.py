"""numpy <-> device tensors through PAGE-LOCKED torch buffers (test and bench plumbing).

Handed a pageable array, the HIP runtime page-locks the array's heap pages for the transfer (a userptr mapping made and torn down per copy).  On this pool that path
produced a rare `Memory access fault by GPU ... on address <host heap address>` -- a dozen clean runs of the GPU suite, then one abort inside a
`torch.from_numpy(...).cuda()` -- so nothing here hands it a pageable pointer: the bytes go through torch's page-locked host allocator
(hipHostMalloc'ed blocks, cached), one extra host copy."""
import numpy as np


def pinned(x):
    """a page-locked CPU tensor with the contents of numpy array `x`"""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(x))
    p = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    p.copy_(t)
    return p


def to_device(x, device="cuda:0"):
    """numpy array -> device tensor"""
    return pinned(x).to(device)


def to_host_tensor(t):
    """device tensor -> page-locked CPU tensor (synchronous, like Tensor.cpu())"""
    import torch
    p = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    p.copy_(t.detach())
    return p


def to_host(t):
    """device tensor -> numpy array (a copy of its own)"""
    return to_host_tensor(t).numpy().copy()
