"""The N > 1 path of bench.py (stream sharding + final probability gather) on CPU: world_size 2 and 8 over gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vadc_amd import shard


def test_stream_blocks_partition_exactly():
    for world in (1, 2, 3, 8):
        for total in (0, 1, 7, 8, 255, 256, 32768):
            blocks = [shard.stream_block(r, world, total) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1
            if total:
                r = shard.owner_of(total - 1, world, total)
                assert shard.stream_block(r, world, total)[1] == total
    with pytest.raises(ValueError):
        shard.stream_block(2, 2, 10)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_streams, chunks, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.stream_block(rank, world, total_streams)
    # stand-in for the per-rank engine output: a value that encodes (global stream, chunk, column)
    s = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1)
    c = torch.arange(chunks, dtype=torch.float32).view(1, -1, 1)
    k = torch.arange(2, dtype=torch.float32).view(1, 1, 2)
    local = s * 1000 + c * 2 + k
    out = shard.gather_probabilities(local, total_streams, dst=0, slot=None)       # both elements, as the engine wrote them: 8 B per chunk
    # the persistent form bench.py uses: preallocated buffers, one dist.gather per step, several steps -- and the speech probability alone (element 1:
    # vadc.c:704-713), 4 B per chunk on the wire
    g = shard.ProbabilityGather(total_streams, chunks, "cpu")
    assert g.bytes_per_chunk == 4 and g.slot == 1
    for i in range(3):
        g.gather(local + i)
        if rank == 0:
            assert g.result().shape == (total_streams, chunks) and torch.equal(g.result(), out[:, :, 1] + i)
    both = shard.ProbabilityGather(total_streams, chunks, "cpu", slot=None)
    both.gather(local + 7)
    if rank == 0:
        assert torch.equal(both.result(), out + 7)
        q.put(out.numpy())
    else:
        assert out is None and both.result() is None and g.result() is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total_streams", [(2, 8), (2, 7), (8, 61), (8, 5)])      # world 8 = the node of the scaling run; 61: ragged blocks; 5: three ranks own nothing
def test_gather_over_gloo(world, total_streams):
    chunks = 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total_streams, chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    s = np.arange(total_streams, dtype=np.float32).reshape(-1, 1, 1)
    c = np.arange(chunks, dtype=np.float32).reshape(1, -1, 1)
    k = np.arange(2, dtype=np.float32).reshape(1, 1, 2)
    assert got.shape == (total_streams, chunks, 2)
    assert np.array_equal(got, s * 1000 + c * 2 + k)
