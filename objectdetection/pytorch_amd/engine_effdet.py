"""Static launch plan for EfficientDet (D0-D5) on the MI355X kernels.

Mirrors architectures/efficientdet.py:528-547 (forward + loss) and its autograd backward:
EfficientNet MBConv backbone (efficientnet.py:80-115) -> BiFPN fast-attention cells
(efficientdet.py:195-267) -> separable-conv heads with per-level BN (:335-393) -> fused loss.

Design (the path is HBM-bound): every activation is written ONCE as the
pre-activation z (bf16); consumers apply swish while loading (depthwise conv, SE pool/scale, BiFPN
node -> depthwise), so BN, swish, the fusion weights and the SE gate never cost a pass of their own.
All 1x1 convs (expand / project / pointwise / down-channel) run on the MFMA gather conv with the
frozen-BN scale folded into the bf16 weights (per pyramid level for the heads' per-level BN).

The backward is emitted from a build-time tape: each forward op registers the launches of its
backward, replayed in reverse at plan construction; gradient buffers are written by their first
producer and accumulated (+=) by later ones.
"""
import ctypes
import os

import torch

from . import _lib, ops_conv, ops_detect, ops_effdet
from ._lib import c_f32, c_i64, c_int, c_vp, ptr
from .engine import LayerSpec, RetinaPlan
from .ops_conv import ACT_NONE, ACT_SIGMOID, ConvProblem, dgrad_problems, fwd_problem
from .ops_effdet import same_pad


class DwSpec(object):
    """Depthwise conv (+ optional frozen BN): weights fp32 [C][1][k][k]."""

    def __init__(self, name, conv, bn=None):
        self.name, self.conv, self.bn = name, conv, bn
        self.C = conv.weight.shape[0]
        self.k = conv.weight.shape[2]
        self.stride = conv.stride[0]
        self.T = self.k * self.k
        # a BatchNorm left in train mode (or made SynchronizedBatchNorm2d) is not folded: the conv stores the raw z and
        # the plan runs the statistics passes of engine.RetinaPlan.bn_fwd / bn_bwd behind it
        self.bn_train = bn is not None and bn.training
        self.folded_bn = bn if (bn is not None and not self.bn_train) else None
        self.w_t = self.scale = self.shift = self.G_t = self.colsum = None
        self.nslabs, self.slabs = 0, None       # partial-sum rows of the weight gradient (mi355_dwconv_wgrad_slabs)


class DwGradDesc(ctypes.Structure):
    """Mi355DwGradDesc (include/mi355det_effdet.h)."""
    _fields_ = [(n, c_vp) for n in ('G_t', 'w', 'scale', 'colsum', 'mean', 'var', 'dW', 'dgamma', 'dbeta')] + \
               [('eps', c_f32), ('C', c_int), ('T', c_int), ('accumulate', c_int), ('nslabs', c_int), ('slab_stride', c_int)]


class SeGradDesc(ctypes.Structure):
    """Mi355SeGradDesc (include/mi355det_effdet.h)."""
    _fields_ = [(n, c_vp) for n in ('pool', 'rpre', 'de', 'dr', 'dw1', 'db1', 'dw2', 'db2')] + \
               [('N', c_int), ('C', c_int), ('Cse', c_int), ('pad_', c_int)]


class DwPrepDesc(ctypes.Structure):
    """Mi355DwPrepDesc (include/mi355det_effdet.h)."""
    _fields_ = [(n, c_vp) for n in ('gamma', 'beta', 'mean', 'var', 'w', 'scale', 'shift', 'w_t')] + \
               [('eps', c_f32), ('C', c_int), ('T', c_int), ('pad_', c_int)]


class FusionDesc(ctypes.Structure):
    """Mi355FusionDesc (include/mi355det_effdet.h)."""
    _fields_ = [('p', c_vp), ('w', c_vp), ('n', c_int), ('pad_', c_int)]


class FusionBwdDesc(ctypes.Structure):
    """Mi355FusionBwdDesc (include/mi355det_effdet.h)."""
    _fields_ = [('p', c_vp), ('gw', c_vp), ('gp', c_vp), ('n', c_int), ('nparts', c_int)]


class SeSpec(object):
    def __init__(self, name, reduce_conv, expand_conv):
        self.name, self.reduce, self.expand = name, reduce_conv, expand_conv
        self.Cse, self.C = reduce_conv.weight.shape[0], reduce_conv.weight.shape[1]


class EffDetPlan(RetinaPlan):
    # ------------------------------------------------------------------ meta / layers
    def _meta(self):
        self.num_classes = self.model.num_classes
        self.A_cell = self.model.anchors.num_anchors_per_cell
        self.input_multiple = 128          # p7 (stride 128) must upsample exactly (efficientdet.py:216)

    def _collect_layers(self):
        m = self.model
        self.specs, self.dw_specs, self.se_specs = [], [], []

        def pw(name, wrapped, bn=None, tf_same=False):
            s = LayerSpec(name, wrapped.conv, bn, tf_same=tf_same)
            self.specs.append(s)
            return s

        def dw(name, wrapped, bn=None):
            s = DwSpec(name, wrapped.conv, bn)
            self.dw_specs.append(s)
            return s

        eff = m.backbone_net.model
        self.stem = pw('stem', eff._conv_stem, eff._bn0, tf_same=True)
        self.stem.need_dgrad = False
        self.blocks = []
        for i, blk in enumerate(eff._blocks):
            a = blk._block_args
            rec = {'k': a.kernel_size, 's': a.stride[0] if isinstance(a.stride, (list, tuple)) else a.stride,
                   'skip': bool(blk.id_skip and (a.stride[0] if isinstance(a.stride, (list, tuple)) else a.stride) == 1
                                and a.input_filters == a.output_filters)}
            pre = 'blocks.%d' % i
            rec['expand'] = pw(pre + '.expand', blk._expand_conv, blk._bn0) if a.expand_ratio != 1 else None
            rec['dw'] = dw(pre + '.dw', blk._depthwise_conv, blk._bn1)
            rec['se'] = SeSpec(pre + '.se', blk._se_reduce.conv, blk._se_expand.conv)
            self.se_specs.append(rec['se'])
            rec['project'] = pw(pre + '.project', blk._project_conv, blk._bn2)
            self.blocks.append(rec)
        self.cells = []
        for ci, cell in enumerate(m.bifpn):
            rec = {'mod': cell, 'first': cell.first_time}
            for nm in ('conv6_up', 'conv5_up', 'conv4_up', 'conv3_up', 'conv4_down', 'conv5_down', 'conv6_down',
                       'conv7_down'):
                blk = getattr(cell, nm)
                rec[nm] = (dw('bifpn.%d.%s.dw' % (ci, nm), blk.depthwise_conv),
                           pw('bifpn.%d.%s.pw' % (ci, nm), blk.pointwise_conv, blk.bn))
            if cell.first_time:
                for nm in ('p5_down_channel', 'p4_down_channel', 'p3_down_channel', 'p5_to_p6', 'p4_down_channel_2',
                           'p5_down_channel_2'):
                    seq = getattr(cell, nm)
                    rec[nm] = pw('bifpn.%d.%s' % (ci, nm), seq[0], seq[1])
            self.cells.append(rec)

        def head(mod, name):
            rec = {'dw': [dw('%s.conv_list.%d.dw' % (name, i), mod.conv_list[i].depthwise_conv)
                          for i in range(mod.num_layers)],
                   'pw': [[pw('%s.conv_list.%d.pw.l%d' % (name, i, lvl), mod.conv_list[i].pointwise_conv,
                              mod.bn_list[lvl][i]) for i in range(mod.num_layers)] for lvl in range(5)],
                   'hdw': dw(name + '.header.dw', mod.header.depthwise_conv),
                   'hpw': [pw('%s.header.pw.l%d' % (name, lvl), mod.header.pointwise_conv) for lvl in range(5)]}
            return rec

        self.reg_head = head(m.regressor, 'regressor')
        self.cls_head = head(m.classifier, 'classifier')

    def _alloc_params(self):
        super(EffDetPlan, self)._alloc_params()
        n_w = sum(s.T * s.C for s in self.dw_specs)
        n_bn = sum(2 * s.C for s in self.dw_specs if s.folded_bn is not None)
        n_g = sum(s.T * s.C + s.C for s in self.dw_specs)
        self.dw_wbuf = self.new((n_w + n_bn + 8,), torch.float32)
        self.dw_scratch = self.new((n_g + 8,), torch.float32, zero=True)
        wo, go = 0, 0
        for s in self.dw_specs:
            s.w_t = self.dw_wbuf[wo:wo + s.T * s.C]
            wo += s.T * s.C
            if s.folded_bn is not None:
                s.scale = self.dw_wbuf[wo:wo + s.C]
                s.shift = self.dw_wbuf[wo + s.C:wo + 2 * s.C]
                wo += 2 * s.C
            s.G_t = self.dw_scratch[go:go + s.T * s.C]
            s.colsum = self.dw_scratch[go + s.T * s.C:go + s.T * s.C + s.C]
            go += s.T * s.C + s.C

    # ------------------------------------------------------------------ prep
    def _build_prep(self):
        super(EffDetPlan, self)._build_prep()
        # every depthwise layer's frozen-BN fold + tap transpose in ONE table-driven launch
        rows = []
        for s in self.dw_specs:
            assert s.conv.weight.is_contiguous()
            bn = s.folded_bn

            def dp(t):
                return 0 if t is None else t.data_ptr()
            rows.append(DwPrepDesc(dp(bn.weight if bn is not None else None), dp(bn.bias if bn is not None else None),
                                   dp(bn.running_mean if bn is not None else None),
                                   dp(bn.running_var if bn is not None else None), dp(s.conv.weight),
                                   dp(s.scale), dp(s.shift), dp(s.w_t),
                                   float(bn.eps) if bn is not None else 0.0, s.C, s.k * s.k, 0))
        if rows:
            arr = (DwPrepDesc * len(rows))(*rows)
            self._dw_prep_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone().to(self.dev)
            self._step(self.prep_steps, 'mi355_dw_prep_table', ptr(self._dw_prep_table), c_int(len(rows)),
                       c_int(max(s.C * s.k * s.k for s in self.dw_specs)))
        self._fusion_rows, self._fusion_table = [], None

    # ------------------------------------------------------------------ gradient bookkeeping (build time)
    def grad(self, t):
        g = self._gbuf.get(id(t))
        if g is None:
            g = self.new(tuple(t.shape))
            self._gbuf[id(t)] = g
            self.keep.append(t)
        return g

    def first_write(self, t):
        """True if nobody wrote grad(t) yet (the caller then overwrites instead of accumulating)."""
        if id(t) in self._gwritten:
            return False
        self._gwritten.add(id(t))
        return True

    def has_grad(self, t):
        return id(t) in self._gwritten

    # ------------------------------------------------------------------ ops (forward launch + taped backward)
    def pw_op(self, spec, x, addend=None):
        out = self.conv(spec, x, ACT_NONE, addend=addend)

        def bwd():
            if not self.has_grad(out):
                return
            dy = self.grad(out)
            if addend is not None:                      # residual skip (efficientnet.py:111-114)
                self._axpy(dy, None, 0, self.grad(addend), not self.first_write(addend))
            if spec.need_dgrad:
                dx = self.grad(x)
                acc = not self.first_write(x)
                self.conv_bwd(spec, x, dy, dx=dx, addend=dx if acc else None)
            else:
                self.conv_bwd(spec, x, dy, dgrad=False)
        self.tape.append(bwd)
        return out

    def dw_op(self, spec, x, x_swish, emit_fwd=True, se_pool=False):
        """(emit_fwd=False: the caller launches the forward itself, fused with the producer of x; everything else —
        output buffer, slab rows, taped backward — is registered here.  se_pool=True: when the geometry is served by
        the LDS-tiled forward, the squeeze-excite pooling of the output rides in the same launch and
        self._last_se_pool = (partials [N][slabs][C], slabs) is set for se_op; otherwise it is None.)"""
        N, H, W, C = x.shape
        pt, P = same_pad(H, spec.k, spec.stride)
        pl, Q = same_pad(W, spec.k, spec.stride)
        out = self.new((N, P, Q, C))
        geo = (c_int(N), c_int(H), c_int(W), c_int(C), c_int(spec.k), c_int(spec.stride), c_int(pt), c_int(pl),
               c_int(P), c_int(Q))
        # the depthwise kernels apply swish as they load a stored pre-activation (once per loaded element in the
        # LDS-tiled forward, once per kernel row in the sliding-window weight gradient; v_exp + v_rcp, rounded to
        # bf16 exactly like a materialised swish(z)): no elementwise pass, no second copy of the activation
        xs, insw = x, int(bool(x_swish))
        self._last_se_pool = None
        raw = out
        if spec.bn_train:
            # raw depthwise output -> batch statistics -> normalised pre-activation (swish stays on the consumer's load);
            # the fused squeeze-excite forms pool / rebuild the BN OUTPUT, which this launch does not have yet
            assert emit_fwd, spec.name
            se_pool = False
        if se_pool and emit_fwd and self.prec != 'fp32' and os.environ.get('MI355_EFF_FUSE_SEPOOL', '1') != '0':
            ns = ctypes.c_int()
            _lib.call('mi355_dwconv_fwd_sepool_slabs', c_int(H), c_int(W), c_int(spec.k), c_int(spec.stride), c_int(P),
                      c_int(Q), ctypes.byref(ns))
            if ns.value > 0:
                part = self.new((N, ns.value, C), torch.float32)
                self._last_se_pool = (part, ns.value)
                self._step(self.fwd_steps, 'mi355_dwconv_fwd_sepool', ptr(xs), c_int(insw), ptr(spec.w_t),
                           ptr(spec.shift), ptr(out), ptr(part), *geo)
                emit_fwd = False
        if emit_fwd:
            self._step(self.fwd_steps, 'mi355_dwconv_fwd', ptr(xs), c_int(insw), ptr(spec.w_t), ptr(spec.shift),
                       ptr(out), *geo)
        if spec.bn_train:
            out = self.bn_fwd(spec, spec.bn, raw, ACT_NONE, None)
        has_beta = int(spec.folded_bn is not None)       # (train-mode BN: d beta / d gamma come from bn_bwd's sums)

        # bf16 plans: bit-reproducible weight gradient, every launch of the layer (the pyramid levels of a shared head
        # conv) stores its own partial rows of the layer's slab workspace; fp32 plans: the generic atomic kernel
        slab_base = None
        if self.prec != 'fp32' and spec.k in (3, 5) and spec.stride in (1, 2):
            slab_base = spec.nslabs
            spec.nslabs += ops_effdet.dwconv_wgrad_nslabs(N, C, spec.k, spec.stride, P, Q)

        # squeeze-excite backward inside this conv's data-gradient launch (the MBConv depthwise convs whose stride-1
        # "same" data gradient takes the LDS-tiled kernel): se_op's backward leaves (gu, gate, gpool) here instead of
        # launching mi355_se_bwd_apply
        fuse_apply = (se_pool and self.prec != 'fp32' and spec.stride == 1 and spec.k in (3, 5) and W >= 16 and H >= 8 and
                      os.environ.get('MI355_EFF_FUSE_SEAPPLY', '1') != '0')
        if fuse_apply:
            self._se_apply_in_dgrad[id(out)] = None

        def bwd():
            if not self.has_grad(out):
                return
            dy = self.grad(out)
            if spec.bn_train:
                dy = self.bn_bwd(spec, dy)
            pend = self._se_apply_in_dgrad.get(id(out)) if fuse_apply else None
            if pend is not None:                            # data gradient first: it also stores dy for the weight gradient
                gu, gate, gpool = pend
                dx = self.grad(x)
                acc = not self.first_write(x)
                self._step(self.bwd_steps, 'mi355_dwconv_dgrad_seapply', ptr(gu), ptr(out), ptr(gate), ptr(gpool), ptr(dy),
                           ptr(spec.w_t), ptr(dx), ptr(x if x_swish else None), c_int(N), c_int(H), c_int(W), c_int(C),
                           c_int(spec.k), c_int(int(acc)))
            if slab_base is not None:
                row = (spec.T + 1) * C
                if spec.slabs is None:                      # all forward launches of the layer are known by now
                    spec.slabs = self.new((spec.nslabs, row), torch.float32, zero=True)
                self._step(self.bwd_steps, 'mi355_dwconv_wgrad_slabs', ptr(xs), c_int(insw), ptr(dy),
                           c_vp(spec.slabs.data_ptr() + slab_base * row * 4),
                           c_int(has_beta), *geo)
            else:
                self._dw_atomic = True                      # dw_scratch is accumulated into: cleared every step
                self._step(self.bwd_steps, 'mi355_dwconv_wgrad', ptr(xs), c_int(insw), ptr(dy), ptr(spec.G_t),
                           ptr(spec.colsum if has_beta else None), *geo)
            if pend is None:
                dx = self.grad(x)
                acc = not self.first_write(x)
                self._step(self.bwd_steps, 'mi355_dwconv_dgrad', ptr(dy), ptr(spec.w_t), ptr(dx),
                           ptr(x if x_swish else None), *(geo + (c_int(int(acc)),)))
        self.tape.append(bwd)
        return out

    def pw_multi_op(self, specs, xs):
        """The pointwise conv of a head layer on every pyramid level: the levels share the conv but fold their own
        BatchNorm (efficientdet.py:335-353, bn_list[level][i]), so each problem of the grouped launch brings its own
        folded weights / bias (Mi355ConvProblem.weight).  One forward, one weight-gradient (per-level workspaces,
        mi355_conv_wgrad_multi) and one data-gradient launch instead of one of each per level."""
        if self.prec == 'fp32' or not self.deterministic or len(xs) < 2 or len(xs) > ops_conv.MAX_PROBLEMS or \
                any(sp.bn_train or sp.stride != 1 for sp in specs) or os.environ.get('MI355_EFF_GROUP_PW', '1') == '0':
            return [self.pw_op(sp, x) for sp, x in zip(specs, xs)]
        s0 = specs[0]
        outs, probs, geo = [], [], []
        for sp, x in zip(specs, xs):
            N, H, W, C = x.shape
            assert C == sp.Cpad and (sp.K, sp.Cpad, sp.T) == (s0.K, s0.Cpad, s0.T), sp.name
            pt, pl, P, Q = sp.geometry(H, W)
            out = self.new((N, P, Q, sp.Kpad), zero=sp.Kpad != sp.K)
            pr = fwd_problem(x, out.data_ptr(), sp.R, sp.S, sp.stride, (pt, pl), K=sp.K, out_size=(P, Q),
                             out_strides=(P * Q * sp.Kpad, Q * sp.Kpad, sp.Kpad))
            pr.weight = sp.w_fwd.data_ptr()
            pr.bias = 0 if sp.bias_tensor is None else sp.bias_tensor.data_ptr()
            probs.append(pr)
            outs.append(out)
            geo.append((N, H, W, pt, pl, P, Q))
            self.fwd_launches.append((sp, [(N, P, Q)]))
        self._gather(self.fwd_steps, probs, s0.w_fwd, s0.K, s0.T, bias=s0.bias_tensor)

        def bwd():
            live = [self.has_grad(o) for o in outs]
            if not any(live):
                return
            assert all(live), 'a head conv with gradient on some of its levels only'
            dys = [self.grad(o) for o in outs]
            group = []
            for sp, x, dy, (N, H, W, pt, pl, P, Q) in zip(specs, xs, dys, geo):
                wp = fwd_problem(x, dy.data_ptr(), sp.R, sp.S, sp.stride, (pt, pl), K=sp.Kpad, out_size=(P, Q))
                group.append((wp, sp, N * P * Q))
            self._wgrad_multi(group)
            if not all(sp.need_dgrad for sp in specs):
                return
            dprobs = []
            for sp, x, dy, (N, H, W, pt, pl, P, Q) in zip(specs, xs, dys, geo):
                dx = self.grad(x)
                acc = not self.first_write(x)
                ps = dgrad_problems(dy, dx.data_ptr(), H, W, sp.Cpad, sp.R, sp.S, sp.stride, pt,
                                    addend=dx if acc else None)
                assert len(ps) == 1
                ps[0].weight, ps[0].bias = sp.w_dgr.data_ptr(), 0
                dprobs.append(ps[0])
            self._gather(self.bwd_steps, dprobs, s0.w_dgr, s0.Cpad, s0.T)
        self.tape.append(bwd)
        return outs

    def dw_multi_op(self, spec, xs, x_swish):
        """The same depthwise conv applied to several feature maps (the pyramid levels of a head conv,
        efficientdet.py:335-353): ONE forward launch, ONE weight-gradient launch (partial rows summed over all the
        levels) and ONE data-gradient launch instead of one of each per level."""
        if self.prec == 'fp32' or spec.stride != 1 or spec.k not in (3, 5) or not (2 <= len(xs) <= 5):
            return [self.dw_op(spec, x, x_swish) for x in xs]
        N, C = xs[0].shape[0], xs[0].shape[3]
        outs = [self.new(tuple(x.shape)) for x in xs]
        insw, nl = int(bool(x_swish)), len(xs)
        arr = ops_effdet.dw_levels(list(zip(xs, outs)), spec.k, 1)
        self._step(self.fwd_steps, 'mi355_dwconv_fwd_multi', arr, c_int(nl), c_int(insw), ptr(spec.w_t),
                   ptr(spec.shift), c_int(N), c_int(C), c_int(spec.k))
        ns = ctypes.c_int()
        _lib.call('mi355_dwconv_wgrad_nslabs_multi', arr, c_int(nl), c_int(N), c_int(C), c_int(spec.k), c_int(1),
                  ctypes.byref(ns))
        slab_base = spec.nslabs
        spec.nslabs += ns.value

        def bwd():
            live = [self.has_grad(o) for o in outs]
            if not any(live):
                return
            assert all(live), 'a shared depthwise conv with gradient on some of its maps only'
            dys = [self.grad(o) for o in outs]
            row = (spec.T + 1) * C
            if spec.slabs is None:
                spec.slabs = self.new((spec.nslabs, row), torch.float32, zero=True)
            warr = ops_effdet.dw_levels(list(zip(xs, dys)), spec.k, 1)
            self._step(self.bwd_steps, 'mi355_dwconv_wgrad_slabs_multi', warr, c_int(nl), c_int(insw),
                       c_vp(spec.slabs.data_ptr() + slab_base * row * 4), c_int(int(spec.bn is not None)),
                       c_int(N), c_int(C), c_int(spec.k), c_int(1))
            dxs = [self.grad(x) for x in xs]
            accs = [not self.first_write(x) for x in xs]
            assert len(set(accs)) == 1, 'levels of one launch must all overwrite or all accumulate'
            garr = ops_effdet.dw_levels([(dy, dx, x if x_swish else None) for dy, dx, x in zip(dys, dxs, xs)], spec.k, 1)
            self._step(self.bwd_steps, 'mi355_dwconv_dgrad_multi', garr, c_int(nl), ptr(spec.w_t), c_int(N), c_int(C),
                       c_int(spec.k), c_int(int(accs[0])))
        self.tape.append(bwd)
        return outs

    def se_op(self, spec, z, pooled=None):
        """pooled = (partials, slabs) when the producing depthwise conv already stored the pooling partials."""
        N, H, W, C = z.shape
        HW = H * W
        ns = ctypes.c_int()
        _lib.call('mi355_se_slabs', c_int(N), c_int(HW), ctypes.byref(ns))
        slabs = ns.value                                    # (of the backward's reduction; the forward's may differ)
        part, fslabs = pooled if pooled is not None else (self.new((N, slabs, C), torch.float32), slabs)
        pool = self.new((N, C), torch.float32)
        rpre = self.new((N, spec.Cse), torch.float32)
        gate = self.new((N, C), torch.float32)
        u = self.new(tuple(z.shape))
        w1, b1, w2, b2 = spec.reduce.weight, spec.reduce.bias, spec.expand.weight, spec.expand.bias
        dims = (c_int(N), c_int(HW), c_int(C))
        if pooled is None:
            self._step(self.fwd_steps, 'mi355_se_pool', ptr(z), c_int(1), ptr(part), *dims)
        self._step(self.fwd_steps, 'mi355_se_fc', ptr(part), c_int(fslabs), ptr(pool), ptr(w1), ptr(b1), ptr(w2),
                   ptr(b2), ptr(rpre), ptr(gate), c_int(N), c_int(C), c_int(spec.Cse))
        self._step(self.fwd_steps, 'mi355_se_scale', ptr(z), ptr(gate), ptr(u), *dims)

        def bwd():
            if not self.has_grad(u):
                return
            gu = self.grad(u)
            ggate = self.new((N, slabs, C), torch.float32)
            gpool = self.new((N, C), torch.float32)
            self._step(self.bwd_steps, 'mi355_se_bwd_reduce', ptr(gu), ptr(z), ptr(ggate), *dims)
            # data path per sample now; the FC parameter gradients of all squeeze-excite layers in one table launch at
            # the end of the backward (_extra_param_grads): no atomics, bit-reproducible
            de = self.new((N, C), torch.float32)
            dr = self.new((N, spec.Cse), torch.float32)
            self._se_bwd.append((spec, pool, rpre, de, dr, N))
            self._step(self.bwd_steps, 'mi355_se_fc_bwd_data', ptr(w1), ptr(w2), ptr(rpre), ptr(gate), ptr(ggate),
                       c_int(slabs), ptr(de), ptr(dr), ptr(gpool), c_int(N), c_int(C), c_int(spec.Cse))
            gz = self.grad(z)
            assert self.first_write(z)
            if id(z) in self._se_apply_in_dgrad:
                # the depthwise conv that produced z builds gz inside its data-gradient launch (dw_op's backward runs next)
                self._se_apply_in_dgrad[id(z)] = (gu, gate, gpool)
            else:
                self._step(self.bwd_steps, 'mi355_se_bwd_apply', ptr(gu), ptr(z), ptr(gate), ptr(gpool), ptr(gz), *dims)
        self.tape.append(bwd)
        return u

    def pool_op(self, x):
        N, H, W, C = x.shape
        _, P = same_pad(H, 3, 2)
        _, Q = same_pad(W, 3, 2)
        out = self.new((N, P, Q, C))
        idx = self.new((N, P, Q, C), torch.int8)
        dims = (c_int(N), c_int(H), c_int(W), c_int(C))
        self._step(self.fwd_steps, 'mi355_maxpool_same_fwd', ptr(x), ptr(out), ptr(idx), *dims)

        def bwd():
            if not self.has_grad(out):
                return
            gx = self.grad(x)
            acc = not self.first_write(x)
            self._step(self.bwd_steps, 'mi355_maxpool_same_bwd', ptr(self.grad(out)), ptr(idx), ptr(gx),
                       *(dims + (c_int(int(acc)),)))
        self.tape.append(bwd)
        return out

    def node_sep_op(self, pair, param, in0, in1, up1, in2=None):
        """SeparableConvBlock(swish(node)): the node's weighted sum is built inside the depthwise conv's forward launch
        (mi355_bifpn_node_dwconv_fwd: one launch and one pass over z fewer per node); backward as the two ops."""
        dws, pws = pair
        if self.prec == 'fp32' or dws.k not in (3, 5) or dws.stride != 1 or os.environ.get('MI355_EFF_FUSE_NODE', '1') == '0':
            return self.sep_op(pair, self.node_op(param, in0, in1, up1, in2))
        z = self.node_op(param, in0, in1, up1, in2, emit_fwd=False)
        w = self._last_node_w
        t = self.dw_op(dws, z, True, emit_fwd=False)
        N, H, W, C = in0.shape
        self._step(self.fwd_steps, 'mi355_bifpn_node_dwconv_fwd', ptr(in0), ptr(in1), c_int(int(up1)), ptr(in2), ptr(w),
                   ptr(z), ptr(dws.w_t), ptr(dws.shift), ptr(t), c_int(N), c_int(H), c_int(W), c_int(C), c_int(dws.k))
        return self.pw_op(pws, t)

    def node_op(self, param, in0, in1, up1, in2=None, emit_fwd=True):
        """z = w0*in0 + w1*in1' (+ w2*in2), w = relu(p)/(sum relu(p) + 1e-4) (efficientdet.py:213-265)."""
        if param is None:
            # plain sum of the inputs (`_forward`, efficientdet.py:269-317: D6 / D7): unit weights, no parameter
            if getattr(self, '_unit_w', None) is None:
                self._unit_w = torch.ones(4, dtype=torch.float32, device=self.dev)
            w = self._unit_w
        else:
            n = param.numel()
            w = self.new((4,), torch.float32, zero=True)
            self._fusion_rows.append(FusionDesc(param.data_ptr(), w.data_ptr(), n, 0))   # one launch for all nodes
        N, H, W, C = in0.shape
        z = self.new(tuple(in0.shape))
        dims = (c_int(N), c_int(H), c_int(W), c_int(C))
        self._last_node_w = w
        if emit_fwd:
            self._step(self.fwd_steps, 'mi355_bifpn_node_fwd', ptr(in0), ptr(in1), c_int(int(up1)), ptr(in2), ptr(w),
                       ptr(z), *dims)

        def bwd():
            if not self.has_grad(z):
                return
            gz = self.grad(z)
            gw = None
            if param is not None:
                npt = ctypes.c_int()
                _lib.call('mi355_bifpn_node_bwd_parts', c_int(N), c_int(H), c_int(W), c_int(C), c_int(int(up1)),
                          ctypes.byref(npt))
                gw = self.new((npt.value, 4), torch.float32, zero=True)   # per-block partial rows, summed in order
                self._fusion_bwd.append((param, n, gw, npt.value))      # all nodes: one table launch behind the backward list
            # the three input gradients and the fusion-weight sums in ONE pass over gz
            g0, a0 = self.grad(in0), not self.first_write(in0)
            g1, a1 = self.grad(in1), not self.first_write(in1)
            g2, a2 = (self.grad(in2), not self.first_write(in2)) if in2 is not None else (None, False)
            self._step(self.bwd_steps, 'mi355_bifpn_node_bwd', ptr(gz), ptr(in0), ptr(in1), c_int(int(up1)), ptr(in2),
                       ptr(w), ptr(gw), ptr(g0), c_int(int(a0)), ptr(g1), c_int(int(a1)), ptr(g2), c_int(int(a2)), *dims)
        self.tape.append(bwd)
        return z

    def drop_connect_op(self, branch, skip, keep):
        """out = branch / keep * floor(keep + u[n]) + skip (efficientnet.py:109-113); u: this block's row of the
        per-step uniform table."""
        row = self._dc_rows
        self._dc_rows += 1
        N = branch.shape[0]
        per = branch.numel() // N
        u = self.dc_u[row]
        out = self.new(tuple(branch.shape))
        self._step(self.fwd_steps, 'mi355_drop_connect', ptr(branch), ptr(u), c_f32(keep), ptr(skip), ptr(out),
                   c_int(N), c_i64(per))

        def bwd():
            if not self.has_grad(out):
                return
            dy = self.grad(out)
            gs, acc = self.grad(skip), not self.first_write(skip)
            gb = self.grad(branch)
            assert self.first_write(branch)
            self._step(self.bwd_steps, 'mi355_drop_connect_bwd', ptr(dy), ptr(u), c_f32(keep), ptr(gb), ptr(gs),
                       c_int(int(acc)), c_int(N), c_i64(per))
        self.tape.append(bwd)
        return out

    def _swish_tmp(self, n):
        """One shared scratch tensor (launches are stream-ordered) for materialised swish(z)."""
        if getattr(self, '_swish_buf', None) is None or self._swish_buf.numel() < n:
            self._swish_buf = self.new((n,))
        return self._swish_buf[:n]

    def _axpy(self, src, w, wi, dst, accumulate):
        self._step(self.bwd_steps, 'mi355_axpy_bf16', ptr(src), ptr(w), c_int(wi), ptr(dst), c_i64(src.numel()),
                   c_int(int(accumulate)))

    def sep_op(self, pair, z):
        """SeparableConvBlock on a swish'ed node: dw3x3 (no bias) -> pw 1x1 (+bias) -> BN."""
        dws, pws = pair
        return self.pw_op(pws, self.dw_op(dws, z, True))

    # zero-initialised fp32 scratch, carved at build time, cleared once per forward / backward
    def _zero_fwd_alloc(self, n):
        t = self._zf[self._zf_off:self._zf_off + n]
        self._zf_off += (n + 3) // 4 * 4
        assert self._zf_off <= self._zf.numel()
        return t

    def _zero_bwd_alloc(self, n):
        t = self._zb[self._zb_off:self._zb_off + n]
        self._zb_off += (n + 3) // 4 * 4
        assert self._zb_off <= self._zb.numel()
        return t

    # ------------------------------------------------------------------ forward
    def _build_forward(self):
        B, H, W = self.B, self.H, self.W
        m = self.model
        self._gbuf, self._gwritten, self.tape = {}, set(), []
        self._se_bwd, self._fusion_bwd = [], []
        self._se_apply_in_dgrad = {}
        nse = sum(B * s.C for s in self.se_specs)
        self._zf = self.new((nse + 64,), torch.float32, zero=True)
        self._zb = self.new((nse + 64 * 8 * len(self.cells) + 64,), torch.float32, zero=True)
        self._zf_off = self._zb_off = 0
        self.img_nhwc = self.new((B, H, W, 8))
        self.dbg = {}
        cur = self.pw_op(self.stem, self.img_nhwc)          # pre-activation z of the stem (swish on load)
        self.dbg['stem_z'] = cur
        cur_sw = True
        feats, last = [], None
        nblk = len(self.blocks)
        eff_gp = self.model.backbone_net.model._global_params
        # drop-connect (utils_efficientnet.py:85-94) is part of the step while the MBConv blocks are in train mode
        self.drop_connect = bool(getattr(self.model, '_plan_drop_connect', False))
        self._dc_rows = 0
        ndc = sum(1 for r in self.blocks if r['skip'])
        self.dc_u = torch.zeros((max(ndc, 1), B), dtype=torch.float32, device=self.dev)
        self.dc_inject = None                                  # tests: a fixed [rows, B] table of uniforms
        for i, rec in enumerate(self.blocks):
            inp = cur
            if rec['expand'] is not None:
                assert not cur_sw
                dw_in, dw_sw = self.pw_op(rec['expand'], inp), True
            else:
                dw_in, dw_sw = inp, cur_sw
            z1 = self.dw_op(rec['dw'], dw_in, dw_sw, se_pool=True)
            u = self.se_op(rec['se'], z1, self._last_se_pool)
            rate = getattr(eff_gp, 'drop_connect_rate', None)
            rate = rate * float(i) / nblk if rate else 0.0                 # efficientdet.py:421-423
            if rec['skip'] and self.drop_connect and rate:
                cur = self.drop_connect_op(self.pw_op(rec['project'], u), inp, 1.0 - rate)
            else:
                cur = self.pw_op(rec['project'], u, addend=inp if rec['skip'] else None)
            pre = 'backbone_net.model._blocks.%d' % i
            self.dbg[pre + '.z1'], self.dbg[pre + '.u'], self.dbg['block%d' % i] = z1, u, cur
            if rec['expand'] is not None:
                self.dbg[pre + '.z0'] = dw_in
            cur_sw = False
            if rec['s'] == 2:                                # efficientdet.py:427-431
                feats.append(last)
            elif i == nblk - 1:
                feats.append(cur)
            last = cur
        p3, p4, p5 = feats[2], feats[3], feats[4]
        fe = (p3, p4, p5)
        for ci, rec in enumerate(self.cells):
            fe = self._cell_fwd(rec, fe)
            for l, f in enumerate(fe):
                self.dbg['cell%d_p%d' % (ci, l + 3)] = f
        self.levels = list(fe)
        self.cells_per_level = [t.shape[1] * t.shape[2] for t in self.levels]
        self.cell_off = [sum(self.cells_per_level[:i]) for i in range(5)]
        self.ncells = sum(self.cells_per_level)
        self.A = self.ncells * self.A_cell
        C = self.num_classes
        self.reg = self.new((B, self.A, 4), torch.float32)
        self.cls = self.new((B, self.A, C), torch.float32)
        self.cls_pitch = (C * self.A_cell + 7) // 8 * 8
        self.reg_pitch = (4 * self.A_cell + 7) // 8 * 8
        self.gcls = self.new((B, self.ncells, self.cls_pitch), zero=True)
        self.greg = self.new((B, self.ncells, self.reg_pitch), zero=True)
        # the two heads only share the pyramid: the box head's forward launches go to the side stream (every
        # buffer they write is their own) and run beside the class head; run_heads joins before the loss
        self._mark_side = True
        self._head_fwd(self.reg_head, self.reg, 4 * self.A_cell, ACT_NONE, self.greg, self.reg_pitch)
        self._mark_side = False
        self._head_fwd(self.cls_head, self.cls, C * self.A_cell, ACT_SIGMOID, self.gcls, self.cls_pitch)
        # loss state (same fused kernels as RetinaNet)
        self.losses = self.new((2,), torch.float32)
        self.argmax = self.new((B, self.A), torch.int32)
        self.state = self.new((B, self.A), torch.int8)
        self.num_pos = self.new((B,), torch.int32)
        nb = ctypes.c_int64()
        _lib.call('mi355_focal_loss_workspace_bytes', c_int(B), c_int(self.A), c_int(C), c_int(1), ctypes.byref(nb))
        self.loss_ws = self.new((nb.value,), torch.uint8)
        self.loss_ws_bytes = nb.value
        self.glosses = self.new((2,), torch.float32)

    def _cell_fwd(self, rec, inputs):
        cell = rec['mod']
        if rec['first']:
            p3, p4, p5 = inputs
            p6_in = self.pool_op(self.pw_op(rec['p5_to_p6'], p5))
            p7_in = self.pool_op(p6_in)
            p3_in = self.pw_op(rec['p3_down_channel'], p3)
            p4_in = self.pw_op(rec['p4_down_channel'], p4)
            p5_in = self.pw_op(rec['p5_down_channel'], p5)
        else:
            p3_in, p4_in, p5_in, p6_in, p7_in = inputs
        att = getattr(cell, 'attention', True)                   # False: plain sum, fusion parameters unused

        def fw(name):
            return getattr(cell, name) if att else None
        if not att:                 # efficientdet.py:269-317: the plain-sum path never touches the fusion parameters
            for n in ('p6_w1', 'p5_w1', 'p4_w1', 'p3_w1', 'p4_w2', 'p5_w2', 'p6_w2', 'p7_w2'):
                if hasattr(cell, n):
                    self.unused_param_ids.add(id(getattr(cell, n)))
        p6_up = self.node_sep_op(rec['conv6_up'], fw('p6_w1'), p6_in, p7_in, True)
        p5_up = self.node_sep_op(rec['conv5_up'], fw('p5_w1'), p5_in, p6_up, True)
        p4_up = self.node_sep_op(rec['conv4_up'], fw('p4_w1'), p4_in, p5_up, True)
        p3_out = self.node_sep_op(rec['conv3_up'], fw('p3_w1'), p3_in, p4_up, True)
        if rec['first']:
            p4_in = self.pw_op(rec['p4_down_channel_2'], inputs[1])
            p5_in = self.pw_op(rec['p5_down_channel_2'], inputs[2])
        p4_out = self.node_sep_op(rec['conv4_down'], fw('p4_w2'), p4_in, p4_up, False, self.pool_op(p3_out))
        p5_out = self.node_sep_op(rec['conv5_down'], fw('p5_w2'), p5_in, p5_up, False, self.pool_op(p4_out))
        p6_out = self.node_sep_op(rec['conv6_down'], fw('p6_w2'), p6_in, p6_up, False, self.pool_op(p5_out))
        p7_out = self.node_sep_op(rec['conv7_down'], fw('p7_w2'), p7_in, self.pool_op(p6_out), False)
        return p3_out, p4_out, p5_out, p6_out, p7_out

    def _head_fwd(self, rec, out, kout, act, gout, pitch):
        """efficientdet.py:335-353 / :374-393; the header's pointwise conv writes fp32 straight into
        the [B, A, .] layout and takes the loss gradient from the padded cell rows of `gout`."""
        B = self.B
        per_img = self.ncells * kout
        feats, sw = list(self.levels), False
        for i in range(len(rec['dw'])):                      # layer-major: the five levels of a depthwise conv = one launch
            ts = self.dw_multi_op(rec['dw'][i], feats, sw)
            feats = self.pw_multi_op([rec['pw'][lvl][i] for lvl in range(len(ts))], ts)
            sw = True
        ts = self.dw_multi_op(rec['hdw'], feats, True)
        # the header's pointwise conv (shared weight, no BN): the five levels = ONE launch each for the forward, the
        # weight gradient (one workspace per level spec, mi355_conv_wgrad_multi) and the data gradient
        grouped = self.prec != 'fp32' and self.deterministic and len(ts) <= ops_conv.MAX_PROBLEMS and \
            os.environ.get('MI355_EFF_GROUP_HDR', '1') != '0'
        fprobs, items = [], []
        for lvl, t in enumerate(ts):
            spec = rec['hpw'][lvl]
            h, w = t.shape[1], t.shape[2]
            off = self.cell_off[lvl]
            pr = fwd_problem(t, out.data_ptr() + 4 * off * kout, 1, 1, 1, 0, out_strides=(per_img, w * kout, kout),
                             K=kout)
            fprobs.append(pr)
            items.append((t, spec, h, w, off))
            self.fwd_launches.append((spec, [(B, h, w)]))
        spec0 = rec['hpw'][0]
        if grouped:
            self._gather(self.fwd_steps, fprobs, spec0.w_fwd, kout, 1, bias=spec0.conv.bias, act=act, out_f32=True)
        else:
            for pr, (t, spec, h, w, off) in zip(fprobs, items):
                self._gather(self.fwd_steps, [pr], spec.w_fwd, kout, 1, bias=spec.conv.bias, act=act, out_f32=True)

        def bwd():
            group, dprobs = [], []
            for t, spec, h, w, off in items:
                dy_ptr = gout.data_ptr() + self.esz * off * pitch
                wp = fwd_problem(t, dy_ptr, 1, 1, 1, 0, K=pitch)
                wp.out_sN = self.ncells * pitch
                group.append((wp, spec, B * h * w))
                gt = self.grad(t)
                assert self.first_write(t)
                dp = ConvProblem()
                dp.in_, dp.out = dy_ptr, gt.data_ptr()
                dp.N, dp.H, dp.W, dp.C, dp.P, dp.Q, dp.sh, dp.sw = B, h, w, pitch, h, w, 1, 1
                dp.TR, dp.TS, dp.dh0, dp.dhs, dp.dw0, dp.dws = 1, 1, 0, -1, 0, -1
                dp.r0, dp.rs, dp.s0, dp.ss, dp.S = 0, 1, 0, 1, 1
                dp.out_mh, dp.out_oh, dp.out_mw, dp.out_ow = 1, 0, 1, 0
                dp.out_sN, dp.out_sH, dp.out_sW = h * w * spec.Cpad, w * spec.Cpad, spec.Cpad
                dp.in_sN = self.ncells * pitch
                dprobs.append(dp)
            if grouped:
                self._wgrad_multi(group)
                self._gather(self.bwd_steps, dprobs, spec0.w_dgr, spec0.Cpad, 1)
            else:
                for (wp, spec, rows), dp in zip(group, dprobs):
                    self._wgrad([wp], spec, rows)
                    self._gather(self.bwd_steps, [dp], spec.w_dgr, spec.Cpad, 1)
        self.tape.append(bwd)

    # ------------------------------------------------------------------ backward
    def _build_backward(self):
        # Same-shape weight gradients of the pointwise convs are independent leaves of the backward pass: the 24 BiFPN node
        # convs (64 -> 64 over 64^2 .. 4^2 maps: 9 us launches that fill a fraction of the chip), the expand / project convs
        # of repeated MBConv blocks.  MI355_EFF_WG_GROUP=1: conv_bwd defers them and five of a kind (or the end of the
        # tape) go out as ONE mi355_conv_wgrad_multi launch (engine.RetinaPlan._defer_wgrad); the gradient buffers a
        # deferred launch reads are complete when its tape entry runs and nothing overwrites them.  Measured and OFF:
        # 494 -> 459 launches per D0 step but 2 300 -> 2 235 img/s (D2 664 -> 650): the side stream, which already
        # finishes behind main, gets its work later, and the grouped launch of five tiny maps is no faster than five.
        self._grouping_wgrads = self._wgrad_grouping_ok() and os.environ.get('MI355_EFF_WG_GROUP', '0') == '1'
        for fn in reversed(self.tape):
            fn()
        self._grouping_wgrads = False
        self._flush_wgrads()
        self._build_param_grads()

    def _step(self, lst, name, *args):
        if name.startswith('__'):                     # python-side pseudo steps (tiny per-sample kernels)
            self.keep.append(args)
            lst.append((None, args, name))
            return
        super(EffDetPlan, self)._step(lst, name, *args)

    SIDE_STEPS = RetinaPlan.SIDE_STEPS + ('mi355_dwconv_wgrad', 'mi355_dwconv_wgrad_f32', 'mi355_dwconv_wgrad_slabs',
                                          'mi355_dwconv_wgrad_slabs_multi')

    def grad_view(self, p, arena):
        off = self.grad_offsets[id(p)]
        flat = self.grad_arenas[arena][off:off + p.numel()]
        if p.dim() == 4 and p.shape[1] > 1 and not p.is_contiguous():
            return flat.as_strided(p.shape, p.stride())
        return flat.view(p.shape)

    def _zero_extra(self):
        return ([self.dw_scratch] if getattr(self, '_dw_atomic', False) else []) + [self._zb]

    def _extra_param_grads(self, arena, stream):
        """Depthwise layers: dW = G * scale, frozen-BN gamma / beta gradients — one table-driven launch for all of
        them (the descriptor table is built once per gradient arena)."""
        tabs = self.__dict__.setdefault('_dw_pg_tables', {})
        tab = tabs.get(arena)
        if tab is None:
            rows = []
            for s in self.dw_specs:
                bn = s.folded_bn

                def dp(t):
                    return 0 if t is None else t.data_ptr()
                if s.slabs is not None:                      # [nslabs][(T + 1) * C] rows: taps, then the dy column sums
                    g_t, csum, nsl, stride = s.slabs.data_ptr(), s.slabs.data_ptr() + s.T * s.C * 4, s.nslabs, (s.T + 1) * s.C
                else:
                    g_t, csum, nsl, stride = dp(s.G_t), dp(s.colsum), 1, 0
                rows.append(DwGradDesc(
                    g_t, dp(s.conv.weight), dp(s.scale), csum if bn is not None else 0,
                    dp(bn.running_mean if bn is not None else None), dp(bn.running_var if bn is not None else None),
                    dp(self.grad_view(s.conv.weight, arena)),
                    dp(self.grad_view(bn.weight, arena) if bn is not None else None),
                    dp(self.grad_view(bn.bias, arena) if bn is not None else None),
                    float(bn.eps) if bn is not None else 0.0, s.C, s.k * s.k, 1, nsl, stride))
            arr = (DwGradDesc * len(rows))(*rows)
            dev_tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone().to(self.dev)
            tab = (dev_tab, len(rows), max(s.C for s in self.dw_specs))
            tabs[arena] = tab
        if tab[1]:
            _lib.call('mi355_dw_param_grads_table', ptr(tab[0]), c_int(tab[1]), c_int(tab[2]), c_vp(stream))
        if self._se_bwd:
            stabs = self.__dict__.setdefault('_se_pg_tables', {})
            stab = stabs.get(arena)
            if stab is None:
                rows = [SeGradDesc(pool.data_ptr(), rpre.data_ptr(), de.data_ptr(), dr.data_ptr(),
                                   self.grad_view(sp.reduce.weight, arena).data_ptr(),
                                   self.grad_view(sp.reduce.bias, arena).data_ptr(),
                                   self.grad_view(sp.expand.weight, arena).data_ptr(),
                                   self.grad_view(sp.expand.bias, arena).data_ptr(), N, sp.C, sp.Cse, 0)
                        for sp, pool, rpre, de, dr, N in self._se_bwd]
                arr = (SeGradDesc * len(rows))(*rows)
                dev_tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone().to(self.dev)
                stab = (dev_tab, len(rows), max(2 * sp.C * sp.Cse + sp.C + sp.Cse for sp, _, _, _, _, _ in self._se_bwd))
                stabs[arena] = stab
            _lib.call('mi355_se_param_grads_table', ptr(stab[0]), c_int(stab[1]), c_int(stab[2]), c_vp(stream))
        if self._fusion_bwd:
            ftabs = self.__dict__.setdefault('_fusion_bwd_tables', {})
            ftab = ftabs.get(arena)
            if ftab is None:
                rows = [FusionBwdDesc(p.data_ptr(), gw.data_ptr(), self.grad_view(p, arena).data_ptr(), n, npt)
                        for p, n, gw, npt in self._fusion_bwd]
                arr = (FusionBwdDesc * len(rows))(*rows)
                ftab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone().to(self.dev)
                ftabs[arena] = ftab
            _lib.call('mi355_fusion_weights_bwd_table', ptr(ftab), c_int(len(self._fusion_bwd)), c_f32(1e-4),
                      c_vp(stream))

    def draw_drop_connect(self):
        """One uniform per (block, image) for this step's stochastic depth (efficientnet.py drop_connect), from a
        plan-owned device generator seeded per data-parallel rank: replicas that all start from
        torch.manual_seed(s) must not drop the same branches of their shards."""
        if not (self.drop_connect and self._dc_rows):
            return
        if self.dc_inject is not None:
            self.dc_u[:self.dc_inject.shape[0]].copy_(self.dc_inject)
            return
        if getattr(self, '_dc_gen', None) is None:
            import torch.distributed as dist
            rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
            self._dc_gen = torch.Generator(device=self.dev)
            self._dc_gen.manual_seed(torch.initial_seed() + 1000003 * rank)
        self.dc_u.uniform_(generator=self._dc_gen)

    def run_heads(self, img):
        self._zf.zero_()
        if self._fusion_table is None and self._fusion_rows:
            arr = (FusionDesc * len(self._fusion_rows))(*self._fusion_rows)
            self._fusion_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone().to(self.dev)
            self._step(self.prep_steps, 'mi355_fusion_weights_table', ptr(self._fusion_table),
                       c_int(len(self._fusion_rows)), c_f32(1e-4))
        if not torch.cuda.is_current_stream_capturing():
            self.draw_drop_connect()               # graph replay: Trainer._step_graph draws before each replay
        super(EffDetPlan, self).run_heads(img)

    def _arena_for_backward(self):
        # gradient accumulation across backward calls without zero_grad is served by autograd
        # adding the returned views; the EfficientDet plan keeps a single launch list
        lo = self.grad_arenas[0].data_ptr()
        hi = lo + self.grad_numel * 4
        for p in self.params:
            g = p.grad
            if g is not None and lo <= g.data_ptr() < hi:
                p.grad = g.clone()                 # detach live .grad tensors from the arena
        return 0
