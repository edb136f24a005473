"""Drop-in for utils/sync_batchnorm/batchnorm.py:`SynchronizedBatchNorm2d` (:63-93, :128-145) for the
one-process-per-GPU world: the per-channel reductions and the normalisation run as HIP kernels
(include/mi355det_syncbn.h), the replicas exchange [sum | sumsq | count] (forward) and
[sum dy | sum dy*xhat] (backward) with one all-reduce each over RCCL (`dp.syncbn_stats`).

Semantics kept from the reference: with ONE replica or in eval mode it is `F.batch_norm` (:65-69, variance
+ eps under the root); with several replicas in training the normaliser is `clamp(biased var, eps)^-1/2`
(a clamp, not an add, :145) and the running variance takes the UNBIASED estimate (:136-141).

This module normalises NCHW fp32 tensors on its own; the fused RetinaNet / EfficientDet launch plans fold
FROZEN BatchNorm into their convs (train.py:273-277) and refuse train-mode BN."""
import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import _lib, dp
from ..._lib import c_int, c_vp, ptr


def _stream(t):
    return c_vp(torch.cuda.current_stream(t.device).cuda_stream)


def _reduce(p, q, mean, invstd, mode):
    """-> (sum_a [C], sum_b [C]) per channel of an NCHW fp32 tensor (see mi355_bn_reduce_nchw)."""
    N, C = p.shape[0], p.shape[1]
    HW = p.numel() // (N * C)
    s = ctypes.c_int()
    _lib.call('mi355_bn_reduce_splits', c_int(N), c_int(C), c_int(HW), ctypes.byref(s))
    part = torch.empty((C, s.value, 2), dtype=torch.float32, device=p.device)
    _lib.call('mi355_bn_reduce_nchw', ptr(p), ptr(q), ptr(mean), ptr(invstd), c_int(N), c_int(C), c_int(HW),
              c_int(mode), c_int(s.value), ptr(part), _stream(p))
    tot = part.sum(1)                                   # [C, 2], fixed order
    return tot[:, 0].contiguous(), tot[:, 1].contiguous()


def _affine(p, q, a, b, c0):
    N, C = p.shape[0], p.shape[1]
    HW = p.numel() // (N * C)
    out = torch.empty_like(p)
    _lib.call('mi355_bn_affine_nchw', ptr(p), ptr(q), ptr(a), ptr(b), ptr(c0), ptr(out), c_int(N), c_int(C),
              c_int(HW), _stream(p))
    return out


class _SyncBNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, group):
        _lib.require_gpu(x)
        x = x.contiguous().float()
        C = x.shape[1]
        n_local = x.numel() // C
        s, ss = _reduce(x, None, None, None, 0)
        size = dp.world(group)[1]
        if size > 1:                                    # batchnorm.py:128-145
            mean, inv_std, unbiased, n = dp.syncbn_stats(s, ss, n_local, eps, group)
            n = float(n)
            biased = unbiased * ((n - 1.0) / n)
            grad_through_var = (biased >= eps)          # where the clamp is inactive
        else:                                           # one replica: F.batch_norm(training=True)
            n = float(n_local)
            mean = s / n
            biased = (ss / n - mean * mean).clamp(min=0.0)
            inv_std = (biased + eps).rsqrt()
            unbiased = biased * (n / max(n - 1.0, 1.0))
            grad_through_var = torch.ones_like(mean, dtype=torch.bool)
        with torch.no_grad():
            if running_mean is not None:
                running_mean.mul_(1 - momentum).add_(momentum * mean)
                running_var.mul_(1 - momentum).add_(momentum * unbiased)
        w = weight if weight is not None else torch.ones_like(mean)
        b0 = bias if bias is not None else torch.zeros_like(mean)
        a = (inv_std * w).contiguous()
        y = _affine(x, None, a, None, (b0 - mean * a).contiguous())
        ctx.save_for_backward(x, w, mean.contiguous(), inv_std.contiguous(), grad_through_var)
        ctx.group, ctx.n, ctx.has_affine = group, n, weight is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, inv_std, through = ctx.saved_tensors
        dy = dy.contiguous().float()
        sdy, sdyx = _reduce(dy, x, mean, inv_std, 1)
        dweight = sdyx.clone() if ctx.has_affine else None       # local sums (DP averages parameter grads later)
        dbias = sdy.clone() if ctx.has_affine else None
        if dp.world(ctx.group)[1] > 1:
            buf = torch.cat([sdy, sdyx])
            torch.distributed.all_reduce(buf, group=ctx.group)
            sdy, sdyx = buf[:sdy.numel()], buf[sdy.numel():]
        mdy, mdyx = sdy / ctx.n, (sdyx / ctx.n) * through.float()
        a = (w * inv_std).contiguous()
        b = (-a * inv_std * mdyx).contiguous()
        c0 = (a * (mean * inv_std * mdyx - mdy)).contiguous()
        dx = _affine(dy, x, a, b, c0)
        return dx, dweight, dbias, None, None, None, None, None


class SynchronizedBatchNorm2d(nn.BatchNorm2d):
    """Same constructor / parameters / buffers as the reference class (an nn.BatchNorm2d subclass there too)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, group=None):
        super(SynchronizedBatchNorm2d, self).__init__(num_features, eps=eps, momentum=momentum, affine=affine)
        self.group = group

    def forward(self, input):
        if not self.training:                           # batchnorm.py:65-69
            return F.batch_norm(input, self.running_mean, self.running_var, self.weight, self.bias, False,
                                self.momentum, self.eps)
        if input.dim() != 4:
            raise ValueError('expected 4D input (got %dD input)' % input.dim())
        return _SyncBNFn.apply(input, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                               self.momentum, self.group)
