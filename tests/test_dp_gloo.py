"""Multi-process data-parallel host logic on CPU (gloo, world_size 2): the N > 1 path of
bench.py / Trainer minus the HIP kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from objectdetection.pytorch_amd import dp
    try:
        g = torch.Generator().manual_seed(100 + rank)
        flat = torch.randn(10007, generator=g)
        keep = flat.clone()
        dp.allreduce_mean_(flat)
        flat_b = keep.clone()
        dp.allreduce_mean_(flat_b, bucket_bytes=4096)                  # bucketed path, same result
        losses = torch.tensor([0.0, 0.0]) if rank == 0 else torch.tensor([1.5, 0.5])
        red = dp.reduce_losses_(losses.clone())
        # SyncBN statistics of a batch sharded over ranks
        x = torch.randn(4, 6, 5, 5, generator=torch.Generator().manual_seed(7))   # same on all ranks
        lo, hi = dp.shard_range(4, rank, world)
        xs = x[lo:hi]
        mean, inv_std, unbiased, n = dp.syncbn_stats(xs.sum((0, 2, 3)), (xs * xs).sum((0, 2, 3)),
                                                     xs.numel() // 6, 1e-5)
        out[rank] = dict(flat=flat.numpy(), flat_b=flat_b.numpy(), keep=keep.numpy(), red=red.numpy(),
                         mean=mean.numpy(), inv_std=inv_std.numpy(), unbiased=unbiased.numpy(), n=float(n),
                         x=x.numpy())
    finally:
        dist.destroy_process_group()


def test_dp_host_logic_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    expect = 0.5 * (r0['keep'] + r1['keep'])
    for r in (r0, r1):
        np.testing.assert_allclose(r['flat'], expect, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(r['flat_b'], expect, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(r['red'], [0.75, 0.25], rtol=1e-6)   # identical skip decision
    assert np.array_equal(r0['flat'], r1['flat'])
    x = torch.from_numpy(r0['x'])
    np.testing.assert_allclose(r0['mean'], x.mean((0, 2, 3)).numpy(), rtol=1e-5, atol=1e-6)
    var_b = x.var((0, 2, 3), unbiased=False)
    np.testing.assert_allclose(r0['inv_std'], var_b.clamp(min=1e-5).pow(-0.5).numpy(), rtol=1e-4)
    np.testing.assert_allclose(r0['unbiased'], x.var((0, 2, 3), unbiased=True).numpy(), rtol=1e-4)
    assert r0['n'] == 100.0
    assert np.array_equal(r0['mean'], r1['mean'])


def test_shard_range_covers_everything():
    from objectdetection.pytorch_amd import dp
    for n in (0, 1, 7, 16, 129):
        for size in (1, 2, 3, 8):
            parts = [dp.shard_range(n, r, size) for r in range(size)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(size - 1))
            assert max(hi - lo for lo, hi in parts) - min(hi - lo for lo, hi in parts) <= 1
