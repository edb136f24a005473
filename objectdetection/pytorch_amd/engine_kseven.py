"""Launch plan of the KSevenDet family (ksevendet/architecture/ksevendet.py:197-231): ResNet trunk -> PANetFPN ->
simple heads.  The trunk and the top-down pyramid are the RetinaNet dataflow under other module names
(neck/fpn.py:235-290 == architectures/retinanet.py:48-69), so `RetinaPlan`'s stem / stage / pyramid builders are
reused as they are.  New here:

* the optional PANet bottom-up path (fpn.py:266-280): N3 = P3, N_{l+1} = conv3x3(conv3x3 s2(N_l) + P_{l+1});
* the heads (head/simple_head.py:143-174, :281-315): every tower layer is ONE conv shared by the five levels
  followed by a PER-LEVEL BatchNorm and ReLU.  Frozen BN (the default, ksevendet.py:164) cannot be folded into
  the shared weights once, so each level gets its own folded bf16 shadow of the shared weight (scale of ITS BN)
  and runs as its own launch; the parameter-gradient pass adds the five levels' dW into the shared gradient and
  keeps dgamma / dbeta per level.  BN layers left in train mode take the batch-statistics passes of `RetinaPlan`.
  The header convs (no BN) stay one grouped launch over the levels, writing fp32 [B, A, .] directly."""
import ctypes
import os

import torch

from . import _lib, ops_conv, ops_detect
from ._lib import c_i64, c_int, ptr
from .engine import LayerSpec, RetinaPlan
from .ops_conv import ACT_NONE, ACT_RELU, ACT_SIGMOID, dgrad_problems, fwd_problem


class KSevenPlan(RetinaPlan):
    def _meta(self):
        m = self.model
        self.num_classes = m.num_classes
        self.A_cell = m.anchors.num_anchors_per_cell
        self.input_multiple = 32
        self.stage_names = ('layer1', 'layer2', 'layer3', 'layer4')
        self.has_maxpool = True
        self.panet = bool(m.neck.panet_buttomup)

    # ------------------------------------------------------------------ layers
    def _collect_layers(self):
        m, bb, neck = self.model, self.model.backbone, self.model.neck
        self.specs = []

        def add(name, conv, bn=None):
            s = LayerSpec(name, conv, bn)
            self.specs.append(s)
            return s

        for mod in m.modules():
            if isinstance(mod, torch.nn.Conv2d) and not mod.weight.permute(0, 2, 3, 1).is_contiguous():
                raise _lib.Mi355Error('conv weights must be KRSC (channels_last) in memory: call '
                                      'model._channels_last_params()')
        self.stem = add('backbone.conv1', bb.conv1, bb.bn1)
        self.stem.need_dgrad = False
        self.blocks = []
        for li, layer in enumerate([getattr(bb, n) for n in self.stage_names]):
            for bi, blk in enumerate(layer):
                pre = 'backbone.layer%d.%d' % (li + 1, bi)
                rec = {'kind': 'bottleneck' if hasattr(blk, 'conv3') else 'basic'}
                rec['c1'] = add(pre + '.conv1', blk.conv1, blk.bn1)
                rec['c2'] = add(pre + '.conv2', blk.conv2, blk.bn2)
                if rec['kind'] == 'bottleneck':
                    rec['c3'] = add(pre + '.conv3', blk.conv3, blk.bn3)
                rec['ds'] = (add(pre + '.downsample', blk.downsample[0], blk.downsample[1])
                             if blk.downsample is not None else None)
                rec['stage_end'] = (bi == len(layer) - 1)
                rec['stage'] = li
                self.blocks.append(rec)
        # the pyramid under RetinaPlan's names (fpn.py:235-290 <-> retinanet.py:48-69)
        self.fpn = {'P5_1': add('neck.conv_top', neck.conv_top),
                    'P5_2': add('neck.posthoc_modules.0', neck.posthoc_modules[0]),
                    'P4_1': add('neck.topdown_lateral_modules.0', neck.topdown_lateral_modules[0].conv_lateral),
                    'P4_2': add('neck.posthoc_modules.1', neck.posthoc_modules[1]),
                    'P3_1': add('neck.topdown_lateral_modules.1', neck.topdown_lateral_modules[1].conv_lateral),
                    'P3_2': add('neck.posthoc_modules.2', neck.posthoc_modules[2]),
                    'P6': add('neck.extra_pyramid_modules.0', neck.extra_pyramid_modules[0]),
                    'P7_2': add('neck.extra_pyramid_modules.1', neck.extra_pyramid_modules[1])}
        if self.panet:
            self.pan1 = [add('neck.panet_buttomup_conv1_modules.%d' % i, c) for i, c in enumerate(neck.panet_buttomup_conv1_modules)]
            self.pan2 = [add('neck.panet_buttomup_conv2_modules.%d' % i, c) for i, c in enumerate(neck.panet_buttomup_conv2_modules)]
        nlev = len(m.head_feature_pyramid_levels)

        def head(mod, name):
            return self._head_specs(add, mod, name, nlev)
        self.reg_tower, self.reg_out = head(m.regressor, 'regressor')
        self.cls_tower, self.cls_out = head(m.classifier, 'classifier')

    @staticmethod
    def _head_specs(add, mod, name, nlev):
        """(tower[layer][level], header[level]) specs of a simple head: shared weights (simple_head.py:116-119: one conv per
        layer, a folded shadow per level BN; ONE header spec for all levels) or a tower and a header per level (:120-127)."""
        share = bool(mod.share_weights)
        tower = [[add('%s.conv_tower.%d.%d@p%d' % (name, 0 if share else lv, i, lv), mod.conv_tower[0 if share else lv][i],
                      mod.bn_modules[lv][i]) for lv in range(nlev)] for i in range(mod.layers_num)]
        if share:
            out = add(name + '.header.0', mod.header[0])
            outs = [out] * nlev
        else:
            outs = [add('%s.header.%d' % (name, lv), mod.header[lv]) for lv in range(nlev)]
        for o in outs:
            o.Kpad = -(-o.K // 64) * 64                # gradient-side pitch (see RetinaPlan._collect_layers)
        return tower, outs

    def _head_out_levels(self, specs, xs, out, kout, act):
        """Header conv of every level writing fp32 straight into [B, A, .]: one grouped launch; unshared headers bring their
        own weights / bias per problem (Mi355ConvProblem.weight)."""
        if all(s is specs[0] for s in specs):
            return RetinaPlan._head_out_fwd(self, specs[0], xs, out, kout, act)
        per_img = self.ncells * kout
        probs = []
        for sp, x, off in zip(specs, xs, self.cell_off):
            w = x.shape[2]
            pr = fwd_problem(x, out.data_ptr() + 4 * off * kout, 3, 3, 1, 1, out_strides=(per_img, w * kout, kout), K=kout)
            pr.weight = sp.w_fwd.data_ptr()
            pr.bias = 0 if sp.conv.bias is None else sp.conv.bias.data_ptr()
            probs.append(pr)
            self.fwd_launches.append((sp, [(x.shape[0], x.shape[1], x.shape[2])]))
        if self.prec == 'fp32' or len(probs) > ops_conv.MAX_PROBLEMS:      # (per-problem weights are a bf16-launch feature)
            for sp, pr in zip(specs, probs):
                pr.weight = pr.bias = 0
                self._gather(self.fwd_steps, [pr], sp.w_fwd, kout, 9, bias=sp.conv.bias, act=act, out_f32=True)
            return
        self._gather(self.fwd_steps, probs, specs[0].w_fwd, kout, 9, bias=specs[0].conv.bias, act=act, out_f32=True)

    # ------------------------------------------------------------------ pyramid: + PANet bottom-up path
    def _fpn_fwd(self, feats):
        RetinaPlan._fpn_fwd(self, feats)               # -> self.p3 .. self.p7, self.levels
        # neck/fpn.py:283-287: P7 = conv(F.relu(P6, inplace=True)) rectifies P6 IN PLACE, and that same tensor object
        # sits in the output list: the heads see relu(P6) (RetinaNet's heads see the raw P6)
        self.p6_relu_inplace = True
        self.levels = [self.p3, self.p4, self.p5, self.p6relu, self.p7]
        if not self.panet:
            return
        # fpn.py:266-280: outputs [N3 = P3, N4, N5]; the extra levels come from C5 and are unchanged
        self.n4pre = self.conv(self.pan1[0], self.p3, addend=self.p4)
        self.n4 = self.conv(self.pan2[0], self.n4pre)
        self.n5pre = self.conv(self.pan1[1], self.n4, addend=self.p5)
        self.n5 = self.conv(self.pan2[1], self.n5pre)
        self.levels = [self.p3, self.n4, self.n5, self.p6relu, self.p7]

    def _fpn_bwd(self, g_feat):
        if self.panet:
            g3, g4, g5, g6, g7 = g_feat
            g_n5pre = self.conv_bwd(self.pan2[1], self.n5pre, g5)                     # = grad of (conv1(N4) + P5)
            g_p5 = g_n5pre
            self.conv_bwd(self.pan1[1], self.n4, g_n5pre, dx=g4, addend=g4)           # N4 feeds its head and N5
            g_n4pre = self.conv_bwd(self.pan2[0], self.n4pre, g4)
            g_p4 = g_n4pre
            self.conv_bwd(self.pan1[0], self.p3, g_n4pre, dx=g3, addend=g3)           # P3 feeds its head and N4
            g_feat = [g3, g_p4, g_p5, g6, g7]
        return RetinaPlan._fpn_bwd(self, g_feat)

    # ------------------------------------------------------------------ heads
    def _heads_fwd(self):
        B = self.B
        self.cells = [t.shape[1] * t.shape[2] for t in self.levels]
        self.cell_off = [sum(self.cells[:i]) for i in range(len(self.levels))]
        self.ncells = sum(self.cells)
        self.A = self.ncells * self.A_cell
        C = self.num_classes
        self.reg = self.new((B, self.A, 4), torch.float32)
        self.cls = self.new((B, self.A, C), torch.float32)
        self._mark_side = True                         # the box head runs beside the class head (side stream)
        self.reg_t = self._tower_fwd(self.reg_tower)
        self._head_out_levels(self.reg_out, self.reg_t[-1], self.reg, 4 * self.A_cell, ACT_NONE)
        self._mark_side = False
        self.cls_t = self._tower_fwd(self.cls_tower)
        self._head_out_levels(self.cls_out, self.cls_t[-1], self.cls, C * self.A_cell, ACT_SIGMOID)
        self.losses = self.new((2,), torch.float32)
        self.argmax = self.new((B, self.A), torch.int32)
        self.state = self.new((B, self.A), torch.int8)
        self.num_pos = self.new((B,), torch.int32)
        nb = ctypes.c_int64()
        _lib.call('mi355_focal_loss_workspace_bytes', c_int(B), c_int(self.A), c_int(C), c_int(1), ctypes.byref(nb))
        self.loss_ws = self.new((nb.value,), torch.uint8)
        self.loss_ws_bytes = nb.value
        self.glosses = self.new((2,), torch.float32)

    def _levels_grouped(self, layer):
        """A tower conv over the pyramid levels as ONE grouped launch: the levels share the conv but fold their own
        frozen BatchNorm, so every problem brings its own folded weights / bias (Mi355ConvProblem.weight).  Train-mode
        BN (statistics passes per level) and the fp32-accurate mode keep one launch per level."""
        return (self.prec != 'fp32' and self.deterministic and 2 <= len(layer) <= ops_conv.MAX_PROBLEMS and
                not any(sp.bn_train for sp in layer) and os.environ.get('MI355_KS_GROUP', '1') != '0')

    def _conv_levels(self, layer, xs, act):
        s0 = layer[0]
        outs, probs = [], []
        for sp, x in zip(layer, xs):
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
            self.fwd_launches.append((sp, [(N, P, Q)]))
        self._gather(self.fwd_steps, probs, s0.w_fwd, s0.K, s0.T, bias=s0.bias_tensor, act=act)
        return outs

    def _tower_fwd(self, tower):
        """layers_num x [shared conv3x3 -> per-level BN -> ReLU] (simple_head.py:148-153)."""
        xs, outs = self.levels, []
        for layer in tower:
            if self._levels_grouped(layer):
                xs = self._conv_levels(layer, xs, ACT_RELU)
            else:
                xs = [self.conv(spec, x, ACT_RELU) for spec, x in zip(layer, xs)]
            outs.append(xs)
        return outs

    def _heads_bwd(self):
        B = self.B
        kc, kr = self.cls_out[0].Kpad, self.reg_out[0].Kpad
        self.gcls = self.new((B, self.ncells, kc), zero=True)
        self.greg = self.new((B, self.ncells, kr), zero=True)
        self.cls_pitch, self.reg_pitch = kc, kr
        g_feat = [self.new(tuple(lv.shape)) for lv in self.levels]
        self._head_bwd(self.reg_out, self.reg_tower, self.reg_t, self.greg, kr, g_feat, first=True)
        self._head_bwd(self.cls_out, self.cls_tower, self.cls_t, self.gcls, kc, g_feat, first=False)
        return g_feat

    def _head_bwd(self, out_specs, tower, acts, gout, kp, g_feat, first):
        B = self.B
        out_spec = out_specs[0]
        shared_out = all(s is out_spec for s in out_specs)
        xs = acts[-1]
        F = xs[0].shape[-1]                            # head feature width
        wprobs, dprobs, gys = [], [], []
        for x, off in zip(xs, self.cell_off):
            h, w = x.shape[1], x.shape[2]
            gy = self.new((B, h, w, F))
            gys.append(gy)
            dy_ptr = gout.data_ptr() + self.esz * off * kp
            pr = fwd_problem(x, dy_ptr, 3, 3, 1, 1, K=kp)
            pr.out_sN = self.ncells * kp
            wprobs.append(pr)
            dp = ops_conv.ConvProblem()                 # data gradient of the header: dy rows with the image stride
            dp.in_, dp.out, dp.mask = dy_ptr, gy.data_ptr(), x.data_ptr()
            dp.N, dp.H, dp.W, dp.C, dp.P, dp.Q, dp.sh, dp.sw = B, h, w, kp, h, w, 1, 1
            dp.TR, dp.TS, dp.dh0, dp.dhs, dp.dw0, dp.dws = 3, 3, 1, -1, 1, -1
            dp.r0, dp.rs, dp.s0, dp.ss, dp.S = 0, 1, 0, 1, 3
            dp.out_mh, dp.out_oh, dp.out_mw, dp.out_ow = 1, 0, 1, 0
            dp.out_sN, dp.out_sH, dp.out_sW = h * w * F, w * F, F
            dp.in_sN = self.ncells * kp
            dprobs.append(dp)
        if shared_out:
            self._wgrad(wprobs, out_spec, B * self.ncells)
            self._gather(self.bwd_steps, dprobs, out_spec.w_dgr, F, 9)
        elif self.prec == 'fp32' or len(dprobs) > ops_conv.MAX_PROBLEMS:
            for sp, wp, dp, x in zip(out_specs, wprobs, dprobs, xs):
                self._wgrad([wp], sp, B * x.shape[1] * x.shape[2])
                self._gather(self.bwd_steps, [dp], sp.w_dgr, F, 9)
        else:                                           # a header per level: per-spec workspaces, per-problem w_dgr
            self._wgrad_multi([(wp, sp, B * x.shape[1] * x.shape[2]) for sp, wp, x in zip(out_specs, wprobs, xs)])
            for sp, dp in zip(out_specs, dprobs):
                dp.weight, dp.bias = sp.w_dgr.data_ptr(), 0
            self._gather(self.bwd_steps, dprobs, out_spec.w_dgr, F, 9)
        for li in range(len(tower) - 1, -1, -1):
            ins = acts[li - 1] if li > 0 else self.levels
            nxt = []
            if self._levels_grouped(tower[li]):
                # one weight-gradient launch (a workspace per level spec) and one data-gradient launch (per-level w_dgr)
                group, dprobs = [], []
                for lv, (spec, x, gy) in enumerate(zip(tower[li], ins, gys)):
                    N, H, W, C = x.shape
                    pt, pl, P, Q = spec.geometry(H, W)
                    wp = fwd_problem(x, gy.data_ptr(), spec.R, spec.S, spec.stride, (pt, pl), K=spec.Kpad, out_size=(P, Q))
                    group.append((wp, spec, N * P * Q))
                    if li > 0:
                        dx = self.new((N, H, W, spec.Cpad))
                        nxt.append(dx)
                        ps = ops_conv.dgrad_problems(gy, dx.data_ptr(), H, W, spec.Cpad, spec.R, spec.S, spec.stride, pt, mask=x)
                    else:                               # into the pyramid-level gradient (second head: accumulate)
                        gf = g_feat[lv]
                        ps = ops_conv.dgrad_problems(gy, gf.data_ptr(), H, W, spec.Cpad, spec.R, spec.S, spec.stride, pt,
                                                     addend=None if first else gf)
                    assert len(ps) == 1
                    ps[0].weight, ps[0].bias = spec.w_dgr.data_ptr(), 0
                    dprobs.append(ps[0])
                self._wgrad_multi(group)
                s0 = tower[li][0]
                self._gather(self.bwd_steps, dprobs, s0.w_dgr, s0.Cpad, s0.T)
                gys = nxt
                continue
            for lv, (spec, x, gy) in enumerate(zip(tower[li], ins, gys)):
                if li > 0:
                    nxt.append(self.conv_bwd(spec, x, gy, mask=x))
                else:                                   # into the pyramid-level gradient (second head: accumulate)
                    gf = g_feat[lv]
                    self.conv_bwd(spec, x, gy, dx=gf, addend=None if first else gf)
            gys = nxt
