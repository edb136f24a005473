"""GPU: SynchronizedBatchNorm2d drop-in (utils/sync_batchnorm/batchnorm.py:63-145) — HIP reductions /
normalisation, one all-reduce of the statistics per direction."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize('shape', [(4, 64, 16, 16), (2, 7, 5, 3), (3, 256, 8, 8)])
def test_single_replica_is_plain_batchnorm(shape):
    """One replica: the reference falls back to F.batch_norm (batchnorm.py:65-69)."""
    from objectdetection.pytorch_amd.utils.sync_batchnorm import SynchronizedBatchNorm2d
    dev = torch.device('cuda', 0)
    C = shape[1]
    x, g = _mk(shape, 1), _mk(shape, 2)
    ref = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.01)
    ref.weight.data = _mk((C,), 3).abs() + 0.5
    ref.bias.data = _mk((C,), 4)
    mine = SynchronizedBatchNorm2d(C, eps=1e-3, momentum=0.01)
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(dev)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    (yr * g).sum().backward()
    xm = x.to(dev).requires_grad_(True)
    ym = mine(xm)
    (ym * g.to(dev)).sum().backward()
    tol = dict(rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(ym.detach().cpu().numpy(), yr.detach().numpy(), **tol)
    np.testing.assert_allclose(xm.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(mine.weight.grad.cpu().numpy(), ref.weight.grad.numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(mine.bias.grad.cpu().numpy(), ref.bias.grad.numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(mine.running_mean.cpu().numpy(), ref.running_mean.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mine.running_var.cpu().numpy(), ref.running_var.numpy(), rtol=1e-5, atol=1e-6)
    mine.eval(), ref.eval()
    np.testing.assert_allclose(mine(x.to(dev)).cpu().detach().numpy(), ref(x).detach().numpy(), **tol)


def _worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from objectdetection.pytorch_amd.utils.sync_batchnorm import SynchronizedBatchNorm2d
        dev = torch.device('cuda', 0)
        N, C, H, W = 6, 12, 5, 7
        x, g = _mk((N, C, H, W), 1), _mk((N, C, H, W), 2)
        x[:, 0] *= 1e-3                                   # channel 0: variance below eps -> the clamp is active
        lo, hi = rank * N // world, (rank + 1) * N // world
        bn = SynchronizedBatchNorm2d(C, eps=1e-4, momentum=0.1).to(dev)
        bn.weight.data = (_mk((C,), 3).abs() + 0.5).to(dev)
        bn.bias.data = _mk((C,), 4).to(dev)
        xs = x[lo:hi].to(dev).requires_grad_(True)
        y = bn(xs)
        (y * g[lo:hi].to(dev)).sum().backward()
        out[rank] = dict(y=y.detach().cpu().numpy(), dx=xs.grad.cpu().numpy(), dw=bn.weight.grad.cpu().numpy(),
                         db=bn.bias.grad.cpu().numpy(), rm=bn.running_mean.cpu().numpy(),
                         rv=bn.running_var.cpu().numpy())
    finally:
        dist.destroy_process_group()


def test_two_replicas_match_full_batch_reference_arithmetic():
    """Two ranks, half the batch each == the reference's statistics over the whole batch: mean = sum/n,
    inv_std = clamp(biased var, eps)^-1/2 (clamp, not add), running_var from the unbiased estimate."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    N, C, H, W = 6, 12, 5, 7
    x, g = _mk((N, C, H, W), 1).double(), _mk((N, C, H, W), 2).double()
    x[:, 0] *= 1e-3
    w, b = (_mk((C,), 3).abs() + 0.5).double().requires_grad_(True), _mk((C,), 4).double().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    n = N * H * W
    s1, s2 = xr.sum((0, 2, 3)), (xr * xr).sum((0, 2, 3))
    mean = s1 / n
    sumvar = s2 - s1 * mean
    inv_std = (sumvar / n).clamp(min=1e-4) ** -0.5
    y = (xr - mean[None, :, None, None]) * (inv_std * w)[None, :, None, None] + b[None, :, None, None]
    (y * g).sum().backward()
    got_y = np.concatenate([out[0]['y'], out[1]['y']])
    got_dx = np.concatenate([out[0]['dx'], out[1]['dx']])
    np.testing.assert_allclose(got_y, y.detach().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(got_dx, xr.grad.numpy(), rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(out[0]['dw'] + out[1]['dw'], w.grad.numpy(), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(out[0]['db'] + out[1]['db'], b.grad.numpy(), rtol=1e-4, atol=1e-3)
    rm = 0.1 * mean.detach().numpy()
    rv = 0.9 + 0.1 * (sumvar / (n - 1)).detach().numpy()
    for r in (0, 1):
        np.testing.assert_allclose(out[r]['rm'], rm, rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(out[r]['rv'], rv, rtol=1e-4, atol=1e-6)
