"""Launch plan of KSevenDet with the BiFPN neck (ksevendet/architecture/ksevendet.py:64-86, neck/bifpn.py:75-347,
config/model/ksevendet_resnet_bifpn.yaml): ResNet trunk -> BiFPN modules -> simple heads (one tower per level).

The trunk is `RetinaPlan`'s (stem, stages and their hand-ordered backward); the heads are `KSevenPlan`'s (grouped launches
over the levels, per-level folded BatchNorm, per-level headers); the neck runs on `EffDetPlan`'s op toolkit (taped backward):
1x1 convs with folded frozen BN on the MFMA gather conv, depthwise 3x3 convs on the LDS-tiled depthwise kernels, the node
sums on the BiFPN node kernel with unit weights.  What differs from EfficientDet's BiFPN (efficientdet.py:195-317) and is
reproduced here as bifpn.py:303-347 computes it:

* plain sums (no fusion parameters), ReLU instead of swish: conv_up[i](relu(in[i] + up2x(up[i + 1]))); the separable block
  ends in BatchNorm + ReLU (bifpn.py:61-72), so its output is stored activated and its gradient is masked with it;
* symmetric padding (nn.Conv2d(padding=1), nn.MaxPool2d(3, 2, 1): -inf padding, the stem pool's kernel) instead of the
  TF-"same" forms;
* `first_time`: down_channel[i] on C3..C5, P6 = maxpool(bn(conv1x1(C5))), P7 = maxpool(P6), and the SECOND 1x1 reductions
  `down_channel_two` of C4 / C5 that replace in[1], in[2] for the bottom-up pass (:325-327);
* `last_time` with fewer head levels than neck levels: only conv_down[0 .. out - 2] exist and the topmost outputs are never
  built (:330-341), although the top-down pass still runs through every neck level.
"""
import ctypes

import torch

from . import _lib, ops_conv
from ._lib import c_i64, c_int, ptr
from .engine import LayerSpec, RetinaPlan
from .engine_effdet import DwSpec, EffDetPlan
from .engine_kseven import KSevenPlan
from .ops_conv import ACT_NONE, ACT_RELU, ACT_SIGMOID


class KSevenBiFPNPlan(EffDetPlan, KSevenPlan):
    def _meta(self):
        m = self.model
        self.num_classes = m.num_classes
        self.A_cell = m.anchors.num_anchors_per_cell
        self.input_multiple = 128          # P7 (stride 128) must upsample exactly onto P6 (bifpn.py:317)
        self.stage_names = ('layer1', 'layer2', 'layer3', 'layer4')
        self.has_maxpool = True
        self.panet = False

    # ------------------------------------------------------------------ layers
    def _collect_layers(self):
        m, bb = self.model, self.model.backbone
        self.specs, self.dw_specs, self.se_specs = [], [], []

        def add(name, conv, bn=None):
            s = LayerSpec(name, conv, bn)
            self.specs.append(s)
            return s

        def dw(name, conv):
            if not conv.weight.is_contiguous():
                conv.weight.data = conv.weight.data.contiguous()
            s = DwSpec(name, conv, None)
            self.dw_specs.append(s)
            return s

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
        self.neck_mods = []
        for mi, mod in enumerate(m.neck.bifpn_modules):
            pre = 'neck.bifpn_modules.%d' % mi

            def sep(name, blk):
                return (dw(name + '.dw', blk.depthwise_conv), add(name + '.pw', blk.pointwise_conv, blk.bn))
            rec = {'mod': mod,
                   'up': [sep('%s.conv_up.%d' % (pre, i), b) for i, b in enumerate(mod.conv_up)],
                   'down': [sep('%s.conv_down.%d' % (pre, i), b) for i, b in enumerate(mod.conv_down)]}
            if mod.first_time:
                rec['dc'] = [add('%s.down_channel.%d' % (pre, i), s[0], s[1]) for i, s in enumerate(mod.down_channel)]
                rec['hi'] = add(pre + '.higher_pyramid_conv.0', mod.higher_pyramid_conv[0][0], mod.higher_pyramid_conv[0][1])
                rec['dc2'] = [add('%s.down_channel_two.%d' % (pre, i), s[0], s[1]) for i, s in enumerate(mod.down_channel_two)]
            self.neck_mods.append(rec)
        nlev = len(m.head_feature_pyramid_levels)
        self.reg_tower, self.reg_out = KSevenPlan._head_specs(add, m.regressor, 'regressor', nlev)
        self.cls_tower, self.cls_out = KSevenPlan._head_specs(add, m.classifier, 'classifier', nlev)

    # ------------------------------------------------------------------ ops the EfficientDet toolkit does not have
    def relu_op(self, z):
        """y = relu(z) (the node's activation, bifpn.py:318); backward: gz = gy where z > 0."""
        y = self.new(tuple(z.shape))
        self._step(self.fwd_steps, 'mi355_relu_mask_bf16', ptr(z), ptr(z), ptr(y), c_i64(z.numel()))

        def bwd():
            if not self.has_grad(y):
                return
            gz = self.grad(z)
            assert self.first_write(z)                 # (a node's sum has one consumer: its activation)
            self._step(self.bwd_steps, 'mi355_relu_mask_bf16', ptr(self.grad(y)), ptr(z), ptr(gz), c_i64(z.numel()))
        self.tape.append(bwd)
        return y

    def pw_act_op(self, spec, x, relu):
        """1x1 conv + folded frozen BN (+ ReLU: the separable block's own activation, bifpn.py:66-72).  The output is stored
        activated; its gradient is masked with it before the conv's weight / data gradients."""
        out = self.conv(spec, x, ACT_RELU if relu else ACT_NONE)

        def bwd():
            if not self.has_grad(out):
                return
            dy = self.grad(out)
            if relu:
                self._step(self.bwd_steps, 'mi355_relu_mask_bf16', ptr(dy), ptr(out), ptr(dy), c_i64(dy.numel()))
            if spec.need_dgrad:
                dx = self.grad(x)
                acc = not self.first_write(x)
                self.conv_bwd(spec, x, dy, dx=dx, addend=dx if acc else None)
            else:
                self.conv_bwd(spec, x, dy, dgrad=False)
        self.tape.append(bwd)
        return out

    def pool3_op(self, x):
        """nn.MaxPool2d(3, stride=2, padding=1) (bifpn.py:213, :229-233): -inf padding, first maximum wins."""
        N, H, W, C = x.shape
        P, Q = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        out = self.new((N, P, Q, C))
        idx = self.new((N, P, Q, C), torch.int8)
        dims = (c_int(N), c_int(H), c_int(W), c_int(C))
        self._step(self.fwd_steps, 'mi355_maxpool3x3s2_fwd', ptr(x), ptr(out), ptr(idx), *dims)

        def bwd():
            if not self.has_grad(out):
                return
            gx = self.grad(x)
            if self.first_write(x):
                self._step(self.bwd_steps, 'mi355_maxpool3x3s2_bwd', ptr(self.grad(out)), ptr(idx), ptr(gx), ptr(None), *dims)
            else:                                       # (the pool backward stores: through a scratch tensor, then add)
                tmp = self.new(tuple(x.shape))
                self._step(self.bwd_steps, 'mi355_maxpool3x3s2_bwd', ptr(self.grad(out)), ptr(idx), ptr(tmp), ptr(None), *dims)
                self._axpy(tmp, None, 0, gx, True)
        self.tape.append(bwd)
        return out

    def sep_relu_op(self, pair, z):
        """SeparableConvBlock(relu(z)): ReLU -> depthwise 3x3 (pad 1, no bias) -> pointwise 1x1 + BN + ReLU."""
        dws, pws = pair
        return self.pw_act_op(pws, self.dw_op(dws, self.relu_op(z), False), True)

    # ------------------------------------------------------------------ forward
    def _build_forward(self):
        B = self.B
        self._gbuf, self._gwritten, self.tape = {}, set(), []
        self._se_bwd, self._fusion_bwd = [], []
        self._se_apply_in_dgrad = {}
        self._zf = self.new((64,), torch.float32, zero=True)
        self._zb = self.new((64,), torch.float32, zero=True)
        self._zf_off = self._zb_off = 0
        self.drop_connect, self._dc_rows, self.dc_inject = False, 0, None
        self.dbg = {}
        x = RetinaPlan._stem_fwd(self)
        feats = RetinaPlan._stages_fwd(self, x)
        self.c3, self.c4, self.c5 = feats[1], feats[2], feats[3]
        fe = [self.c3, self.c4, self.c5]
        for mi, rec in enumerate(self.neck_mods):
            fe = self._module_fwd(rec, fe)
            for l, f in enumerate(fe):
                self.dbg['bifpn%d_out%d' % (mi, l)] = f
        self.levels = list(fe)
        KSevenPlan._heads_fwd(self)                    # (allocates cls / reg and the loss state)

        def heads_bwd():                               # the heads' hand-ordered backward as one tape entry
            for t in self.levels:
                assert self.first_write(t)
            self._heads_bwd_into([self.grad(t) for t in self.levels])
        self.tape.append(heads_bwd)

    def _module_fwd(self, rec, inputs):
        """BiFPNModule._forward (bifpn.py:303-347)."""
        mod = rec['mod']
        n = mod.neck_pyramid_num
        if mod.first_time:
            fin = [self.pw_act_op(rec['dc'][i], inputs[i], False) for i in range(mod.in_pyramid_num)]
            fin.append(self.pool3_op(self.pw_act_op(rec['hi'], inputs[-1], False)))
            for _ in range(mod.higher_pyramid_levels_num - 1):
                fin.append(self.pool3_op(fin[-1]))
        else:
            fin = list(inputs)
        up = [None] * n
        up[n - 1] = fin[n - 1]
        for i in range(n - 2, -1, -1):                 # top-down: in[i] + nearest-2x(up[i + 1])
            up[i] = self.sep_relu_op(rec['up'][i], self.node_op(None, fin[i], up[i + 1], True))
        if mod.first_time:
            for i in range(1, mod.in_pyramid_num):
                fin[i] = self.pw_act_op(rec['dc2'][i - 1], inputs[i], False)
        down = [up[0]]
        ndown = n - 1 if not mod.last_time else min(mod.out_pyramid_num, n - 1)
        for i in range(1, ndown):                      # bottom-up: in[i] + up[i] + maxpool(down[i - 1])
            down.append(self.sep_relu_op(rec['down'][i - 1], self.node_op(None, fin[i], up[i], False, self.pool3_op(down[-1]))))
        if not mod.last_time or mod.out_pyramid_num == n:
            down.append(self.sep_relu_op(rec['down'][n - 2], self.node_op(None, fin[n - 1], self.pool3_op(down[n - 2]), False)))
        return [down[i] for i in mod.out_neck_pyramid_idx]

    # ------------------------------------------------------------------ backward
    def _heads_bwd_into(self, g_feat):
        B = self.B
        kc, kr = self.cls_out[0].Kpad, self.reg_out[0].Kpad
        self.gcls = self.new((B, self.ncells, kc), zero=True)
        self.greg = self.new((B, self.ncells, kr), zero=True)
        self.cls_pitch, self.reg_pitch = kc, kr
        KSevenPlan._head_bwd(self, self.reg_out, self.reg_tower, self.reg_t, self.greg, kr, g_feat, first=True)
        KSevenPlan._head_bwd(self, self.cls_out, self.cls_tower, self.cls_t, self.gcls, kc, g_feat, first=False)

    def _build_backward(self):
        for fn in reversed(self.tape):
            fn()
        # the trunk: RetinaPlan._stages_bwd takes the gradient of layer4's output behind its ReLU and the raw sums for
        # layer3 / layer2 (their masks ride in the blocks' own data-gradient launches)
        for t in (self.c3, self.c4, self.c5):
            assert self.has_grad(t)
        g5 = self.grad(self.c5)
        self._step(self.bwd_steps, 'mi355_relu_mask_bf16', ptr(g5), ptr(self.c5), ptr(g5), c_i64(g5.numel()))
        self._grouping_wgrads = False
        g = RetinaPlan._stages_bwd(self, g5, {2: self.grad(self.c4), 1: self.grad(self.c3)})
        RetinaPlan._stem_bwd(self, g)
        self._build_param_grads()
