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


class _FlatAdamWHandle(torch.optim.Optimizer):
    """The `optimizer` object of the reference loop (train.py:259-268) for the flat-arena trainer: a real
    torch.optim.Optimizer whose single param_group carries lr / betas / eps / weight_decay, so that
    `optim.lr_scheduler.ReduceLROnPlateau(optimizer, patience=3, verbose=True)` (train.py:268, stepped at :346)
    and any other scheduler drive the HIP AdamW launches.  The update itself is `Trainer.step`; `step()` here is a
    no-op and `zero_grad()` has nothing to do (the plan owns and re-zeroes the gradient arena)."""

    def __init__(self, trainer, params, lr, betas, eps, weight_decay):
        self._trainer = trainer
        super(_FlatAdamWHandle, self).__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    def step(self, closure=None):
        return None

    def zero_grad(self, set_to_none=True):
        return None

    def state_dict(self):
        """param_groups as usual; the moments as flat arenas (keys 'exp_avg', 'exp_avg_sq', 'step')."""
        t = self._trainer
        d = super(_FlatAdamWHandle, self).state_dict()
        d['flat_state'] = {'exp_avg': t.m.detach().clone(), 'exp_avg_sq': t.v.detach().clone(),
                           'opt_state': t.state.detach().clone()}
        return d

    def load_state_dict(self, d):
        d = dict(d)
        flat = d.pop('flat_state', None)
        super(_FlatAdamWHandle, self).load_state_dict(d)
        if flat is not None:
            t = self._trainer
            t.m.copy_(flat['exp_avg'])
            t.v.copy_(flat['exp_avg_sq'])
            t.state.copy_(flat['opt_state'])


class Trainer(object):
    def __init__(self, model, lr=1e-4, weight_decay=5e-4, betas=(0.9, 0.999), eps=1e-8,
                 clip_norm=0.1, group=None, bucket_bytes=0, graph=False, max_annots=128, overlap=True,
                 segment_bytes=32 << 20):
        """graph=True replays the step from two captured hipGraphs (forward+loss+backward | clip+AdamW;
        the RCCL all-reduce stays between them on the stream): ~400-2000 launches cost one graph
        launch each.  Inputs are copied into static buffers; annotations are padded with -1 rows
        to `max_annots` (pad rows are ignored exactly, losses.py:50)."""
        self.model, self.group, self.bucket_bytes = model, group, bucket_bytes
        self.use_graph, self.max_annots = graph, max_annots
        self.overlap, self.segment_bytes = overlap, segment_bytes
        self._graphs = None
        self.clip_norm = clip_norm
        self.lib = _lib.lib()
        self._flatten()
        self.optimizer = _FlatAdamWHandle(self, self.params, lr, betas, eps, weight_decay)

    # hyper-parameters live in the optimizer handle's param_group (what LR schedulers edit)
    @property
    def lr(self):
        return self.optimizer.param_groups[0]['lr']

    @lr.setter
    def lr(self, v):
        self.optimizer.param_groups[0]['lr'] = v

    @property
    def wd(self):
        return self.optimizer.param_groups[0]['weight_decay']

    @property
    def betas(self):
        return self.optimizer.param_groups[0]['betas']

    @property
    def eps(self):
        return self.optimizer.param_groups[0]['eps']

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
        if self.use_graph:
            # hipGraph replay (forward + loss + backward | clip + AdamW); with more than one rank the gradient arena is
            # exchanged in ONE all-reduce between the two replays (the overlapped per-range exchange is an eager path)
            return self._step_graph(img, annot)
        if self._overlap():
            # (launched eagerly: the step is GPU-bound, ~260 launches cost ~2 ms of host time per ~14 ms
            # step, and collectives issued between hipGraph replays serialised badly in rehearsal)
            plan, losses = self._fwd_bwd_overlapped(img, annot)
            self._update(plan, dp.reduce_losses_(losses.clone(), self.group))
            return losses
        plan, losses = self._fwd_bwd(img, annot)
        self._exchange_and_update(plan, losses)
        return losses

    # ------------------------------------------------------------------ overlapped DP exchange
    def _overlap(self):
        """world > 1: the gradient arena is exchanged in a few large tail ranges, each all-reduced on the
        RCCL stream as soon as the backward pass has finished it (heads, pyramid, layer4, ... in that
        order), while the remaining backward launches keep the compute stream busy."""
        from . import engine
        pc = getattr(self.model, '_plan_class', None)            # (EfficientDet's / BiFPN's tape-built backward: one exchange)
        import os
        # (MI355_FORCE_OVERLAP=1: take this path on a world-size-1 process group too, to rehearse the exchange
        #  against the real RCCL stream machinery on one GPU: tools/nccl_rehearse.py)
        multi = dp.world(self.group)[1] > 1 or (os.environ.get('MI355_FORCE_OVERLAP') == '1' and
                                                torch.distributed.is_available() and torch.distributed.is_initialized())
        return (multi and self.overlap and pc is not None and
                pc._build_backward is engine.RetinaPlan._build_backward)

    def _allreduce_async(self, t):
        import torch.distributed as dist
        size = dp.world(self.group)[1]
        if dist.get_backend(self.group) == 'nccl':                       # RCCL: average inside the collective
            return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=True), None
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True), 1.0 / size

    def _begin(self, img, annot):
        model = self.model
        plan = model._plan(img)
        # RCCL: the two compute streams stay (weight gradients beside data gradients) and each range is handed to
        # the collective from the stream it is final on.  Needs GPU_MAX_HW_QUEUES >= 8 (set on package import):
        # with HIP's default of 4 hardware queues RCCL's streams push both compute streams onto one queue
        # (world-size-1 RCCL rehearsal, tools/nccl_rehearse.py: 11.0 ms per step against 13.1 ms, 12.4 ms on
        # one compute stream).  gloo (CPU tests) moves tensors through the host and synchronises streams: one
        # compute stream there (two gloo ranks time-slicing one GPU degenerated to 1.2 s per step on two).
        import os
        import torch.distributed as dist
        two = os.environ.get('MI355_DP_TWO_STREAMS', '1' if dist.get_backend(self.group) == 'nccl' else '0')
        plan.use_side = two == '1'
        anchors = model.anchors(img)
        plan.run_heads(img)
        losses = plan.run_loss(anchors, annot)
        plan.begin_backward(self.ones[0:1], self.ones[1:2])
        return plan, losses

    def _finish_exchange(self, plan, works):
        grad = plan.grad_arenas[plan.last_arena]
        for (w, scale), seg in works:
            w.wait()
            if scale is not None:
                grad[seg[1]:seg[2]].mul_(scale)

    def _fwd_bwd_overlapped(self, img, annot):
        plan, losses = self._begin(img, annot)
        grad = plan.grad_arenas[plan.last_arena]
        works = []
        for seg in plan.backward_segments(self.segment_bytes):
            plan.run_backward_segment(seg)
            with torch.cuda.stream(plan.exchange_stream()):        # (= the current stream: one compute stream)
                works.append((self._allreduce_async(grad[seg[1]:seg[2]]), seg))
        plan.join_side()
        self._finish_exchange(plan, works)
        return plan, losses

    def _fwd_bwd(self, img, annot):
        model = self.model
        plan = model._plan(img)
        anchors = model.anchors(img)
        plan.run_heads(img)
        losses = plan.run_loss(anchors, annot)
        # loss = cls.mean() + reg.mean() (train.py:312-314): upstream gradient 1 for both
        plan.run_backward(self.ones[0:1], self.ones[1:2], want_views=False)
        return plan, losses

    def _exchange(self, plan, losses):
        grad = plan.grad_arenas[plan.last_arena]
        dp.allreduce_mean_(grad, self.group, self.bucket_bytes)
        if dp.world(self.group)[1] > 1:
            return dp.reduce_losses_(losses.clone(), self.group)
        return losses

    def _exchange_and_update(self, plan, losses):
        self._update(plan, self._exchange(plan, losses))

    def _step_graph(self, img, annot):
        dev = img.device
        if self._graphs is None:
            B, M = img.shape[0], annot.shape[1]
            if M > self.max_annots:
                raise ValueError('annotations per image %d > max_annots %d' % (M, self.max_annots))
            self._s_img = torch.empty_like(img, dtype=torch.float32).contiguous()
            self._s_ann = -torch.ones((B, self.max_annots, 5), dtype=torch.float32, device=dev)
            self._s_flag = torch.zeros(2, dtype=torch.float32, device=dev)
        if annot.shape[1] > self.max_annots or tuple(img.shape) != tuple(self._s_img.shape):
            raise ValueError('graph mode needs a fixed image shape and <= max_annots annotations')
        self._s_img.copy_(img)
        self._s_ann.fill_(-1.0)
        self._s_ann[:, :annot.shape[1]].copy_(annot)
        if self._graphs is None:
            # warm-up on a side stream (plan build, hipFuncSetAttribute, allocator), then capture
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                plan, losses = self._fwd_bwd(self._s_img, self._s_ann)
                self._update(plan, self._exchange(plan, losses))
                first = losses.clone()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                plan, losses = self._fwd_bwd(self._s_img, self._s_ann)
            self._g_plan, self._g_losses = plan, losses
            with torch.cuda.graph(g2, pool=g1.pool()):
                self._update(plan, self._s_flag)
            self._graphs = (g1, g2)
            return first                           # this step already ran eagerly (the warm-up)
        g1, g2 = self._graphs
        if self._skip_sig != (id(self._g_plan), tuple(p.requires_grad for p in self.params)):
            # a requires_grad flag flipped: the skip table (pointer + count are by-value arguments of the captured
            # launches) changed: capture the update again
            g2 = torch.cuda.CUDAGraph()
            self._skip_table(self._g_plan)                     # (the table upload must not fall into the capture)
            torch.cuda.synchronize(dev)
            with torch.cuda.graph(g2, pool=g1.pool()):
                self._update(self._g_plan, self._s_flag)
            self._graphs = (g1, g2)
        self._refresh_hyper()
        draw = getattr(self._g_plan, 'draw_drop_connect', None)
        if draw is not None:
            draw()                                 # stochastic-depth uniforms are drawn outside the captured region
        g1.replay()
        self._s_flag.copy_(self._exchange(self._g_plan, self._g_losses))
        g2.replay()
        return self._g_losses

    def _skip_table(self, plan):
        """Device table of the arena ranges the reference's clip + AdamW never touch (`p.grad is None`): frozen
        (requires_grad == False, the commented head-only fine-tune of train.py:206-216) and unused parameters
        (BiFPN fusion weights of D6/D7).  Rebuilt when a requires_grad flag flips."""
        sig = (id(plan), tuple(p.requires_grad for p in self.params))
        if getattr(self, '_skip_sig', None) != sig:
            rs = plan.skip_ranges()
            self._skip = (torch.tensor(rs, dtype=torch.int64, device=self.flat.device).reshape(-1) if rs else None,
                          len(rs))
            self._skip_sig = sig
        return self._skip

    def _hyper_values(self):
        return (float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.wd))

    def _refresh_hyper(self):
        """Graph mode: lr / betas / eps / weight_decay live in a device vector the captured AdamW launch reads, so
        that `ReduceLROnPlateau(trainer.optimizer)` (train.py:268, :346) and any other scheduler edit keep driving
        the replays; refreshed (one 20-byte copy on the launch stream) only when a value changed."""
        vals = self._hyper_values()
        if getattr(self, '_hyper_sig', None) != vals:
            if getattr(self, '_hyper', None) is None:
                self._hyper = torch.zeros(8, dtype=torch.float32, device=self.flat.device)
            self._hyper[:5].copy_(torch.tensor(vals, dtype=torch.float32), non_blocking=False)
            self._hyper_sig = vals
        return self._hyper

    def _update(self, plan, loss_flag):
        grad = plan.grad_arenas[plan.last_arena]
        s = c_vp(torch.cuda.current_stream(grad.device).cuda_stream)
        skip, nskip = self._skip_table(plan)
        if self.use_graph:
            capturing = torch.cuda.is_current_stream_capturing()
            hyper = self._hyper if capturing else self._refresh_hyper()       # (no copy inside a capture)
            self._call('mi355_grad_sumsq_masked', ptr(grad), c_i64(plan.grad_numel_padded), ptr(self.partials),
                       ptr(skip), c_int(nskip), s)
            self._call('mi355_adamw_prepare', ptr(self.partials), ptr(loss_flag), c_f32(self.clip_norm),
                       ptr(self.state), s)
            self._call('mi355_adamw_flat_hyper', ptr(self.flat), ptr(grad), ptr(self.m), ptr(self.v),
                       c_i64(self.npad), ptr(hyper), ptr(self.state), c_int(1), ptr(skip), c_int(nskip), s)
            return
        self._call('mi355_grad_sumsq_masked', ptr(grad), c_i64(plan.grad_numel_padded), ptr(self.partials),
                   ptr(skip), c_int(nskip), s)
        self._call('mi355_adamw_prepare', ptr(self.partials), ptr(loss_flag), c_f32(self.clip_norm),
                   ptr(self.state), s)
        self._call('mi355_adamw_flat_masked', ptr(self.flat), ptr(grad), ptr(self.m), ptr(self.v),
                   c_i64(self.npad), c_f32(self.lr), c_f32(self.betas[0]), c_f32(self.betas[1]),
                   c_f32(self.eps), c_f32(self.wd), ptr(self.state), c_int(1), ptr(skip), c_int(nskip), s)

    def _call(self, name, *args):
        if _lib.TRACE_PATH:
            return _lib.call(name, *args)
        rc = getattr(self.lib, name)(*args)
        if rc:
            raise _lib.Mi355Error('%s failed (%d): %s' % (name, rc, self.lib.mi355_last_error().decode()))
