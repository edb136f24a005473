"""Training-step driver = the loop body of the reference's train.py:292-321 on the HIP launch plan.

    trainer = Trainer(model)                 # AdamW(lr 1e-4, wd 5e-4) + clip_grad_norm_(0.1)
    losses = trainer.step(imgs, annots)      # device tensor [cls_loss, reg_loss]; no host sync

Parameters, gradients and Adam moments live in flat fp32 arenas: the DP exchange is one RCCL
all-reduce, the optimizer is three HBM-bound launches (include/mi355det_optim.h).
"""
import ctypes

import torch

from . import _lib, dp
from ._lib import c_f32, c_i64, c_int, c_vp, ptr


class Trainer(object):
    def __init__(self, model, lr=1e-4, weight_decay=5e-4, betas=(0.9, 0.999), eps=1e-8,
                 clip_norm=0.1, group=None, bucket_bytes=0):
        self.model, self.group, self.bucket_bytes = model, group, bucket_bytes
        self.lr, self.wd, self.betas, self.eps, self.clip_norm = lr, weight_decay, betas, eps, clip_norm
        self.lib = _lib.lib()
        self._flatten()

    def _flatten(self):
        """Re-home every parameter into ONE flat fp32 arena (same order as model.parameters(), same
        shapes/strides, so state_dict / load_state_dict / checkpoints are unaffected)."""
        params = [p for p in self.model.parameters()]
        dev = params[0].device
        _lib.require_gpu(params[0])
        n = sum(p.numel() for p in params)
        npad = (n + 3) // 4 * 4
        flat = torch.zeros(npad, dtype=torch.float32, device=dev)
        off = 0
        for p in params:
            view = flat[off:off + p.numel()]
            view = view.as_strided(p.shape, p.stride()) if p.dim() == 4 else view.view(p.shape)
            view.copy_(p.data)
            p.data = view
            off += p.numel()
        self.params, self.n, self.npad, self.flat = params, n, npad, flat
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)
        self.state = torch.zeros(4, dtype=torch.int32, device=dev)      # OptState (16 bytes)
        self.partials = torch.zeros(1024, dtype=torch.float32, device=dev)
        self.ones = torch.ones(2, dtype=torch.float32, device=dev)

    @property
    def step_count(self):
        return int(self.state[0].item())

    def step(self, img, annot):
        model = self.model
        plan = model._plan(img)
        anchors = model.anchors(img)
        st = torch.cuda.current_stream(img.device).cuda_stream
        plan.run_heads(img)
        losses = plan.run_loss(anchors, annot)
        # loss = cls.mean() + reg.mean() (train.py:312-314): upstream gradient 1 for both
        plan.run_backward(self.ones[0:1], self.ones[1:2])
        grad = plan.grad_arenas[plan.last_arena]
        dp.allreduce_mean_(grad, self.group, self.bucket_bytes)
        loss_flag = losses
        if dp.world(self.group)[1] > 1:
            loss_flag = dp.reduce_losses_(losses.clone(), self.group)
        s = c_vp(st)
        self._call('mi355_grad_sumsq', ptr(grad), c_i64(plan.grad_numel_padded), ptr(self.partials), s)
        self._call('mi355_adamw_prepare', ptr(self.partials), ptr(loss_flag), c_f32(self.clip_norm),
                   ptr(self.state), s)
        self._call('mi355_adamw_flat', ptr(self.flat), ptr(grad), ptr(self.m), ptr(self.v),
                   c_i64(self.npad), c_f32(self.lr), c_f32(self.betas[0]), c_f32(self.betas[1]),
                   c_f32(self.eps), c_f32(self.wd), ptr(self.state), c_int(1), s)
        return losses

    def _call(self, name, *args):
        rc = getattr(self.lib, name)(*args)
        if rc:
            raise _lib.Mi355Error('%s failed (%d): %s' % (name, rc, self.lib.mi355_last_error().decode()))
