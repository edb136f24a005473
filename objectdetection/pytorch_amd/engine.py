"""Static launch plans for the RetinaNet hot path (MI355X-first: no tracing compiler, no per-op
autograd graph).

A `RetinaPlan` is built once per (batch, height, width): it owns every activation / gradient
buffer in HBM (persistent, sized for the 288 GB part) and a flat list of pre-marshalled C-ABI
launches (`ctypes` structs built once).  A training step is then

    prep (frozen-BN fold + bf16 weight shadows)  ->  forward launches  ->  fused loss
    ->  loss backward (writes bf16 d/dlogit)     ->  backward launches ->  parameter grads

with no Python tensor math and no host<->device sync in between, so the whole step can also be
captured in a hipGraph (`RetinaPlan.capture`).

Mirrors the forward of architectures/retinanet.py:272-295 and its autograd backward.
"""
import ctypes
import os
import struct

import torch

from . import _lib, ops_conv, ops_detect
from ._lib import c_f32, c_i64, c_int, c_vp, ptr
from .ops_conv import (ACT_NONE, ACT_RELU, ACT_SIGMOID, ConvLaunch, dgrad_problems, fwd_problem,
                       out_hw, pick_split_rows)


GUARD_BYTES = 4096 if os.environ.get('MI355_GUARD', '0') == '1' else 0
GUARD_PATTERN = 0xA5


def _ceil8(v):
    return (v + 7) // 8 * 8


# ---------------------------------------------------------------------------------------------- plan tables
PLAN_MAX_ARGS = 24                       # include/mi355det_plan.h
PLAN_MAIN, PLAN_SIDE, PLAN_SIDE_RR, PLAN_JOIN, PLAN_SIDE2, PLAN_JOIN2 = 0, 1, 2, 3, 4, 5


class PlanStep(ctypes.Structure):
    """Mi355PlanStep."""
    _fields_ = [('fn', ctypes.c_int32), ('where', ctypes.c_int32), ('host', ctypes.c_int32), ('nargs', ctypes.c_int32),
                ('arg', ctypes.c_uint64 * PLAN_MAX_ARGS)]


class PlanState(ctypes.Structure):
    """Mi355PlanState."""
    _fields_ = [('next', ctypes.c_int32), ('failed', ctypes.c_int32), ('side_dirty', ctypes.c_int32),
                ('side2_dirty', ctypes.c_int32), ('rr', ctypes.c_int32), ('forks', ctypes.c_int32),
                ('reserved', ctypes.c_int32 * 2)]


_BYREF = type(ctypes.byref(ctypes.c_int()))
_M64 = (1 << 64) - 1


def _slot(a):
    """The 8-byte argument slot of one ctypes argument of a step (low bytes = the value, as the C side reads it with
    the parameter's own type)."""
    if a is None:
        return 0
    if isinstance(a, ctypes.c_void_p):
        return a.value or 0
    if isinstance(a, (ctypes.c_int, ctypes.c_int64, ctypes.c_uint, ctypes.c_uint64)):
        return a.value & _M64
    if isinstance(a, ctypes.c_float):
        return struct.unpack('<I', struct.pack('<f', a.value))[0]
    if isinstance(a, ctypes.c_double):
        return struct.unpack('<Q', struct.pack('<d', a.value))[0]
    if isinstance(a, _BYREF):
        return ctypes.addressof(a._obj)
    if isinstance(a, (ctypes.Array, ctypes.Structure)):
        return ctypes.addressof(a)
    if isinstance(a, ctypes._Pointer):
        return ctypes.cast(a, ctypes.c_void_p).value or 0
    if isinstance(a, int):
        return a & _M64
    raise TypeError('cannot marshal %r into a plan step' % (a,))


class LayerSpec(object):
    """One conv (+ optional frozen BN) of the network and its device-side shadows."""

    def __init__(self, name, conv, bn=None, tf_same=False):
        self.name, self.conv, self.bn = name, conv, bn
        self.K, self.C, self.R, self.S = conv.weight.shape
        self.stride, self.pad = conv.stride[0], conv.padding[0]
        self.tf_same = tf_same         # TF "same" padding (utils_extra_efficientnet.py:33-52)
        self.Cpad, self.Kpad = _ceil8(self.C), _ceil8(self.K)
        self.T = self.R * self.S
        self.need_dgrad = True
        # train-mode BatchNorm (batch statistics, optionally synchronised over the replicas): the BN cannot be
        # folded into the weights; the plan runs conv -> statistics -> normalise (+ residual, ReLU) passes instead
        self.bn_train = bn is not None and bn.training
        # filled by the plan
        self.w_fwd = self.w_dgr = self.scale = self.shift = None
        self.G = self.colsum = None
        self.nslabs = 0
        self.direct = False        # wgrad writes straight into the parameter-gradient arena

    @property
    def folded_bn(self):
        return self.bn if (self.bn is not None and not self.bn_train) else None

    @property
    def bias_tensor(self):
        return self.shift if self.folded_bn is not None else self.conv.bias

    def geometry(self, H, W):
        """-> (pad_top, pad_left, P, Q) for an H x W input."""
        if getattr(self, 's2d', False):    # space-to-depth stem: 4x4 / stride 1, pad 2 before and 1 after
            return 2, 2, H, W
        if self.tf_same:
            P, Q = -(-H // self.stride), -(-W // self.stride)
            eh = self.stride * (P - 1) + 1 + (self.R - 1) - H
            ew = self.stride * (Q - 1) + 1 + (self.S - 1) - W
            return eh // 2, ew // 2, P, Q
        P, Q = out_hw(H, W, self.R, self.stride, self.pad)[0], out_hw(H, W, self.S, self.stride, self.pad)[1]
        return self.pad, self.pad, P, Q


def group_exchange(world):
    """Issue the BatchNorm statistics all-reduce?  (more than one replica, or the world-size-1 pricing run)"""
    return world > 1 or os.environ.get('MI355_BN_FORCE_ALLREDUCE') == '1'


def precision_of(model):
    """'bf16' (throughput mode: bf16 MFMA operands, fp32 accumulate) or 'fp32' (the reference's own precision,
    train.py:259-321 runs without autocast: fp32 activations / gradients / shadows on v_mfma_f32_16x16x4_f32).
    `model.precision` wins over the MI355_CONV_PRECISION environment default."""
    p = getattr(model, 'precision', None) or os.environ.get('MI355_CONV_PRECISION', 'bf16')
    if p not in ('bf16', 'fp32'):
        raise ValueError("precision must be 'bf16' or 'fp32', got %r" % (p,))
    return p


# entry points that have an fp32-mode twin taking the same arguments (include/mi355det_conv.h)
F32_TWINS = {'mi355_conv_gather': 'mi355_conv_gather_f32', 'mi355_conv_wgrad_slabs': 'mi355_conv_wgrad_slabs_f32',
             'mi355_maxpool3x3s2_fwd': 'mi355_maxpool3x3s2_fwd_f32', 'mi355_maxpool3x3s2_bwd': 'mi355_maxpool3x3s2_bwd_f32',
             'mi355_maxpool3x3s2_bwd_pooled': 'mi355_maxpool3x3s2_bwd_pooled_f32',
             'mi355_sumpool2x2_bf16': 'mi355_sumpool2x2_f32', 'mi355_relu_mask_bf16': 'mi355_relu_mask_f32',
             # EfficientDet (include/mi355det_effdet.h)
             'mi355_dwconv_fwd': 'mi355_dwconv_fwd_f32', 'mi355_dwconv_dgrad': 'mi355_dwconv_dgrad_f32',
             'mi355_dwconv_wgrad': 'mi355_dwconv_wgrad_f32', 'mi355_se_pool': 'mi355_se_pool_f32',
             'mi355_se_scale': 'mi355_se_scale_f32', 'mi355_se_bwd_reduce': 'mi355_se_bwd_reduce_f32',
             'mi355_se_bwd_apply': 'mi355_se_bwd_apply_f32', 'mi355_bifpn_node_fwd': 'mi355_bifpn_node_fwd_f32',
             'mi355_bifpn_node_wgrad': 'mi355_bifpn_node_wgrad_f32', 'mi355_bifpn_node_bwd': 'mi355_bifpn_node_bwd_f32', 'mi355_axpy_bf16': 'mi355_axpy_f32',
             'mi355_drop_connect': 'mi355_drop_connect_f32', 'mi355_drop_connect_bwd': 'mi355_drop_connect_bwd_f32', 'mi355_upsample_bwd_axpy': 'mi355_upsample_bwd_axpy_f32',
             'mi355_maxpool_same_fwd': 'mi355_maxpool_same_fwd_f32', 'mi355_maxpool_same_bwd': 'mi355_maxpool_same_bwd_f32'}
# entry points that never see an activation tensor (fp32 vectors / parameter tables): the same in both modes
F32_NEUTRAL = ('mi355_rccl_', 'mi355_se_fc', 'mi355_se_fc_bwd', 'mi355_se_slabs', 'mi355_fusion_weights', 'mi355_dw_prep', 'mi355_dw_param',
               'mi355_prep_weights', 'mi355_param_grads', 'mi355_bn_nhwc', 'mi355_focal', 'mi355_stream')


class RetinaPlan(object):
    supports_fp32 = True

    def __init__(self, model, B, H, W, device):
        self.model, self.B, self.H, self.W, self.dev = model, B, H, W, device
        self.lib = _lib.lib()
        self.prec = precision_of(model)
        if self.prec == 'fp32' and not self.supports_fp32:
            raise _lib.Mi355Error('%s: the fp32-accurate mode is not built for this plan' % type(self).__name__)
        self.adt = torch.float32 if self.prec == 'fp32' else torch.bfloat16     # activation / gradient storage
        self.esz = 4 if self.prec == 'fp32' else 2
        self.keep = []                 # ctypes objects referenced by the steps
        self.prep_steps, self.fwd_steps, self.bwd_steps, self.pgrad_steps = [], [], [], []
        self.bufs = []
        self.fwd_launches = []         # (spec, [(n, p, q) per problem]) for FLOP accounting
        # weight-gradient reduction over pixel chunks: 'slabs' (default; plain stores into per-chunk slabs,
        # summed in fixed order by the parameter-gradient table launch: bit-reproducible) or 'atomic'
        # (fp32 atomics straight into the gradient).  MI355_DETERMINISTIC=1 forces 'slabs'.
        self.wgrad_mode = os.environ.get('MI355_WGRAD', 'slabs')
        if os.environ.get('MI355_DETERMINISTIC', '0') == '1' or self.prec == 'fp32':
            self.wgrad_mode = 'slabs'
        assert self.wgrad_mode in ('slabs', 'atomic')
        self.deterministic = self.wgrad_mode == 'slabs'
        self.use_side = os.environ.get('MI355_SIDE_STREAM', '1') == '1'
        # MI355_SIDE_STREAMS=2: the weight-gradient launches alternate between TWO side streams (two independent
        # latency-bound backbone wgrads in flight beside the data-gradient chain)
        self.n_side = max(1, int(os.environ.get('MI355_SIDE_STREAMS', '1')))
        self._side, self._side2 = None, None
        self._pstate = PlanState()
        self._side_ids, self._mark_side = set(), False
        # a THIRD stream for short branches of the backward chain (the data gradient of a stage's downsample conv): they
        # run beside the main chain while the side stream carries the weight gradients (MI355_BRANCH_STREAM=0: in line)
        self._branch_ids = set()
        self.use_branch = os.environ.get('MI355_BRANCH_STREAM', '1') == '1'
        self.last_arena = 0
        # one forward, one backward: the activations / assignment state of the LAST forward live in this plan's
        # buffers; `forward_serial` lets a backward detect that another forward overwrote them (retinanet._TrainStep)
        self.forward_serial = 0
        # parameters no launch of this plan reads (BiFPN fusion weights with attention=False): the reference leaves
        # their .grad None, so autograd gets None and the flat optimizer skips them (train.py:320-321)
        self.unused_param_ids = set()
        self.bn_saved = {}
        self._pending_wgrads, self._grouping_wgrads = {}, False
        self._meta()
        if H % self.input_multiple or W % self.input_multiple:
            # the reference's FPN / BiFPN (nn.Upsample x2 + add, retinanet.py:52-61,
            # efficientdet.py:216) needs this too; its Resizer pads to multiples of 32
            raise ValueError('input must be a multiple of %d, got %dx%d' % (self.input_multiple, H, W))
        self._collect_layers()
        self._maybe_s2d_stem()
        self._alloc_params()
        self._build_prep()
        self._build_forward()
        self._build_backward()
        self._param_ptrs = self._ptr_signature()

    def _meta(self):
        self.num_classes = self.model.classificationModel.num_classes
        self.A_cell = self.model.anchors.num_anchors_per_cell
        self.input_multiple = 32
        self.stage_names = ('layer1', 'layer2', 'layer3', 'layer4')
        self.fpn_names = ('P5_1', 'P5_2', 'P4_1', 'P4_2', 'P3_1', 'P3_2', 'P6', 'P7_2')
        self.has_maxpool = True

    # ------------------------------------------------------------------ bookkeeping
    def new(self, shape, dtype=None, zero=False):
        if GUARD_BYTES:
            return self._new_guarded(shape, dtype or self.adt, zero)
        t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype or self.adt, device=self.dev)
        self.bufs.append(t)
        return t

    # MI355_GUARD=1 (a debugging mode, tests/test_gpu_guards.py): every plan buffer sits between two 4 KiB zones filled
    # with a byte pattern; `check_guards()` names the buffers a launch wrote past.  An out-of-bounds store of a
    # hand-written kernel otherwise lands in a neighbouring block of torch's caching allocator and shows up as a wrong
    # number somewhere else, or as a memory fault only when the buffer happens to end a segment.
    def _new_guarded(self, shape, dtype, zero):
        import traceback
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        raw = torch.full((nbytes + 2 * GUARD_BYTES,), GUARD_PATTERN, dtype=torch.uint8, device=self.dev)
        body = raw[GUARD_BYTES:GUARD_BYTES + nbytes]
        if zero:
            body.zero_()
        t = body.view(dtype).view(shape)
        fr = [f for f in traceback.extract_stack(limit=6)[:-2] if 'new' != f.name]
        where = ' < '.join('%s:%d' % (f.name, f.lineno) for f in reversed(fr[-3:]))
        self.__dict__.setdefault('_guards', []).append((raw, nbytes, where, shape, dtype))
        self.bufs.append(t)
        return t

    def check_guards(self):
        """[(where, shape, dtype, side, first bad byte offset)] of the guard zones that no longer hold the pattern."""
        bad = []
        for raw, nbytes, where, shape, dtype in self.__dict__.get('_guards', []):
            for side, zone in (('before', raw[:GUARD_BYTES]), ('after', raw[GUARD_BYTES + nbytes:])):
                ne = (zone != GUARD_PATTERN).nonzero()
                if ne.numel():
                    off = int(ne[0].item()) - (GUARD_BYTES if side == 'before' else 0)
                    bad.append((where, shape, str(dtype), side, off, int(ne.numel())))
        return bad

    def _ptr_signature(self):
        # (cached tensor objects: walking the module tree costs ~1 ms per step; `.data` re-homing, `.to()` and
        # load_state_dict all keep the Parameter / buffer objects and change or keep their storage)
        if getattr(self, '_sig_tensors', None) is None:
            self._sig_tensors = list(self.model.parameters()) + list(self.model.buffers())
        return [t.data_ptr() for t in self._sig_tensors]

    def stale(self):
        return self._param_ptrs != self._ptr_signature()

    def _step(self, lst, name, *args):
        self.keep.append(args)
        if self.prec == 'fp32':
            if name not in F32_TWINS and not name.startswith(F32_NEUTRAL):
                raise _lib.Mi355Error('%s has no fp32-mode twin' % name)
            name = F32_TWINS.get(name, name)
        lst.append((getattr(self.lib, name), args, name))
        if getattr(self, '_mark_side', False):
            self._side_ids.add(id(args))

    def _tail_wgrads_main(self):
        return int(os.environ.get('MI355_TAIL_WGRADS_MAIN', '0'))

    def _join_step(self, lst, branch=False):
        """A join edge inside a launch list: the main stream waits for what the list has sent to the side stream so far
        (branch=True: to the branch stream only)."""
        lst.append((None, (), 'join2:' if branch else 'join:'))

    SIDE_STEPS = ('mi355_conv_wgrad_slabs', 'mi355_conv_wgrad_colsum', 'mi355_conv_wgrad_slabs_f32', 'mi355_conv_wgrad_multi')

    # ------------------------------------------------------------------ replay (include/mi355det_plan.h)
    # A step list is marshalled ONCE into a table of 8-byte argument slots (`PlanStep`); `mi355_plan_run` enqueues a
    # range of it below Python: one ctypes call per range instead of one per launch plus one per fork edge (the
    # Python loop it replaces cost ~20 us per launch: 7.9 ms of host time per 10 ms RetinaNet-R50 step).  Which steps
    # go to the side stream is part of the TABLE (`where`), and the runner issues a fork edge in front of a side step
    # whenever the main stream has received work since the last one: the dependency edges of a replay do not depend
    # on host-side flags, so an eager replay and a hipGraph capture of the same table have the same edges.
    def _compiled(self, steps):
        cache = self.__dict__.setdefault('_tables', {})
        key = id(steps)
        ent = cache.get(key)
        if ent is not None and ent[1] == len(steps) and ent[3] is steps:
            return ent
        n = len(steps)
        arr = (PlanStep * max(n, 1))()
        lookup, arity = self.lib.mi355_plan_lookup, self.lib.mi355_plan_arity
        # MI355_TAIL_WGRADS_MAIN=n: the LAST n weight-gradient launches of a backward list stay on the main stream.  The
        # side stream (weight gradients + tables) finishes behind main (tools/side_balance.py: 0.55 ms at R50 512^2, 0.33 ms
        # at EfficientDet-D0), and at the end of the backward main has nothing left to run beside it.
        keep_main = set()
        ntail = self._tail_wgrads_main()
        if ntail and (steps is self.bwd_steps or steps is getattr(self, 'bwd_steps_alt', None)):
            side_idx = [i for i, st_ in enumerate(steps) if st_[2] in self.SIDE_STEPS]
            keep_main = set(side_idx[-ntail:])
        for i, (fn, args, name) in enumerate(steps):
            st = arr[i]
            if name.startswith('py:'):                         # host-side step (BN statistics all-reduce)
                st.host, st.fn = 1, -1
                continue
            if name in ('join:', 'join2:'):                    # main waits for the side stream(s) / the branch stream
                st.where, st.fn = (PLAN_JOIN if name == 'join:' else PLAN_JOIN2), -1
                continue
            if fn is None:
                raise _lib.Mi355Error('unknown deferred step %r' % (name,))
            idx = lookup(name.encode())
            if idx < 0:
                raise _lib.Mi355Error('%s is not a stream-taking entry point of include/*.h (tools/gen_plan_table.py)' % name)
            if len(args) != arity(idx):
                raise _lib.Mi355Error('%s takes %d arguments before the stream, the step has %d' % (name, arity(idx), len(args)))
            st.fn, st.nargs = idx, len(args)
            if name in self.SIDE_STEPS and i not in keep_main:
                st.where = PLAN_SIDE_RR if self.n_side > 1 else PLAN_SIDE
            elif id(args) in self._side_ids:
                st.where = PLAN_SIDE
            elif id(args) in self._branch_ids:
                st.where = PLAN_SIDE2
            for j, a in enumerate(args):
                st.arg[j] = _slot(a)
        ent = (arr, n, [i for i in range(n) if arr[i].host], steps)
        cache[key] = ent
        return ent

    def _run(self, steps, stream, begin=0, end=None):
        """Enqueue `steps[begin:end]`: main-stream launches on `stream`; with the side stream enabled
        (MI355_SIDE_STREAM=1) the weight-gradient launches (and the box head) go to a second HIP stream, forked behind
        everything issued so far: a wgrad and the data gradient of the same layer only share inputs, so they run
        concurrently and fill each other's ramp-up / tail; `join_side()` is the fence in front of the consumers of dW."""
        if os.environ.get('MI355_PLAN_PY') == '1' or _lib.TRACE_PATH:
            return self._run_py(steps, stream, begin, end)
        arr, n, hosts, _ = self._compiled(steps)
        end = n if end is None else end
        if hosts and torch.cuda.is_current_stream_capturing():
            # (a torch.distributed call per BatchNorm layer cannot be part of a hipGraph; on an RCCL group the exchange
            #  is a table step - mi355_rccl_allreduce_sum_f32 - and is captured with the launches around it)
            raise _lib.Mi355Error('hipGraph capture of a plan with host-side steps (synchronised BatchNorm statistics '
                                  'exchanged through torch.distributed on a non-RCCL group): use Trainer(graph=False)')
        side = side2 = None
        if self._side_active():
            side = self._side_stream().cuda_stream
            if self.n_side > 1 or (self.use_branch and self._branch_ids):
                if self._side2 is None:
                    self._side2 = torch.cuda.Stream(device=self.dev)
                side2 = self._side2.cuda_stream
        run, state = self.lib.mi355_plan_run, self._pstate
        s, ss, ss2, pstate = c_vp(stream), c_vp(side), c_vp(side2), ctypes.byref(state)
        pos = begin
        while True:
            rc = run(arr, c_int(pos), c_int(end), s, ss, ss2, pstate)
            if rc == 0:
                return
            if rc == 1:                                        # stopped in front of a host step
                steps[state.next][0](stream)
                pos = state.next + 1
                continue
            name = steps[state.failed][2] if 0 <= state.failed < n else '?'
            raise _lib.Mi355Error('%s failed (%d): %s' % (name, rc, self.lib.mi355_last_error().decode()))

    def _run_py(self, steps, stream, begin=0, end=None):
        """MI355_PLAN_PY=1: the same replay as a Python loop (one ctypes call per launch and per fork edge): the form
        mi355_plan_run replaced, kept for A/B timing of the host cost (tools/host_cost.py)."""
        s = c_vp(stream)
        side = c_vp(self._side_stream().cuda_stream) if self._side_active() else None
        fork = self.lib.mi355_stream_fork
        which = {id(self.prep_steps): 'prep', id(self.fwd_steps): 'fwd', id(self.bwd_steps): 'bwd'}.get(id(steps), 'steps')
        for i, (fn, args, name) in enumerate(steps[begin:end]):
            if name.startswith('py:'):
                fn(stream)
                continue
            if name == 'join:':
                self.join_side()
                continue
            if name == 'join2:':                               # (branch steps run in line in this mode)
                continue
            if _lib.TRACE_PATH:
                _lib.trace_begin('%s[%d] %s' % (which, begin + i, name))
            if side is not None and (name in self.SIDE_STEPS or id(args) in self._side_ids):
                rc = fork(s, side) or fn(*args, side)
                self._pstate.side_dirty = 1
            else:
                rc = fn(*args, s)
            if _lib.TRACE_PATH and not rc:
                _lib.trace_end()
            if rc:
                raise _lib.Mi355Error('%s failed (%d): %s' % (name, rc, self.lib.mi355_last_error().decode()))

    @property
    def _side_dirty(self):
        return bool(self._pstate.side_dirty or self._pstate.side2_dirty)

    def _side_active(self):
        """Two-stream replay?  Not while a hipGraph is being captured unless MI355_GRAPH_BRANCH=1 (a branched graph
        replays no faster than the single-stream order: the step is GPU-bound)."""
        if not self.use_side:
            return False
        if torch.cuda.is_current_stream_capturing():
            return os.environ.get('MI355_GRAPH_BRANCH', '0') == '1'
        return True

    def _side_stream(self):
        if getattr(self, '_side', None) is None:
            self._side = torch.cuda.Stream(device=self.dev)
        return self._side

    def _join(self, fold_only):
        if self._side is None:
            return
        rc = self.lib.mi355_plan_join(c_vp(self._stream()), c_vp(self._side.cuda_stream),
                                      c_vp(self._side2.cuda_stream if self._side2 is not None else None),
                                      ctypes.byref(self._pstate), c_int(fold_only))
        if rc:
            raise _lib.Mi355Error('plan_join: %s' % self.lib.mi355_last_error().decode())

    def _fold_side2(self):
        """Order the primary side stream behind the second one (before anything that needs ALL weight gradients)."""
        self._join(1)

    def join_side(self):
        """The current stream waits for everything issued to the side stream(s)."""
        self._join(0)

    # ------------------------------------------------------------------ layers / parameters
    def _collect_layers(self):
        m = self.model
        self.specs = []

        def add(name, conv, bn=None):
            s = LayerSpec(name, conv, bn)
            self.specs.append(s)
            return s

        for mod in m.modules():
            if isinstance(mod, torch.nn.Conv2d) and not mod.weight.permute(0, 2, 3, 1).is_contiguous():
                raise _lib.Mi355Error('conv weights must be KRSC (channels_last) in memory: call '
                                      'model._channels_last_params()')
        self._add = add
        self.stem = add('conv1', m.conv1, m.bn1)
        self.stem.need_dgrad = False
        self.blocks = []
        for li, layer in enumerate([getattr(m, n) for n in self.stage_names]):
            for bi, blk in enumerate(layer):
                pre = 'layer%d.%d' % (li + 1, bi)
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
        f = m.fpn
        self.fpn = {n: add('fpn.' + n, getattr(f, n)) for n in self.fpn_names}
        self.reg_tower = [add('regressionModel.conv%d' % i, getattr(m.regressionModel, 'conv%d' % i))
                          for i in range(1, 5)]
        self.reg_out = add('regressionModel.output', m.regressionModel.output)
        self.cls_tower = [add('classificationModel.conv%d' % i, getattr(m.classificationModel, 'conv%d' % i))
                          for i in range(1, 5)]
        self.cls_out = add('classificationModel.output', m.classificationModel.output)
        # gradient-side channel pitch of the head outputs: a multiple of 64 (720 -> 768, 36 -> 64), so that their
        # data gradient (C = pitch) runs on the non-flat kernels; pad channels of d loss / d logit stay zero
        for s in (self.reg_out, self.cls_out):
            s.Kpad = -(-s.K // 64) * 64

    def _maybe_s2d_stem(self):
        """The ResNet stem (7x7 / stride 2 / pad 3 over 3 channels) as a 4x4 / stride-1 conv over the 2x2 space-to-depth
        image (conv_aux.hip, 'space-to-depth stem'): 256 instead of 392 reduction columns, 32-B gather pieces.  The
        spec is re-shaped in place; its master weight / gradient are the derived buffers stem_wp / stem_dwp
        [K][12][4][4], packed before the weight preparation and scattered back into the real dW after the
        parameter-gradient table.  bf16 plans, even input sizes; MI355_STEM_S2D=0 keeps the direct form."""
        s = getattr(self, 'stem', None)
        self.stem_s2d = bool(
            s is not None and self.prec != 'fp32' and os.environ.get('MI355_STEM_S2D', '1') != '0' and
            (s.C, s.R, s.S, s.stride, s.pad) == (3, 7, 7, 2, 3) and not s.tf_same and s.conv.bias is None and
            self.H % 2 == 0 and self.W % 2 == 0)
        if not self.stem_s2d:
            return
        s.true_ct = s.C * s.T                      # FLOP accounting keeps the 3 x 49 taps of the real conv
        s.s2d = True
        s.C, s.Cpad, s.R, s.S, s.T, s.stride, s.pad = 12, 16, 4, 4, 16, 1, 2
        self.stem_wp = self.new((s.K, 12, 4, 4), torch.float32, zero=True)
        self.stem_dwp = self.new((s.K, 12, 4, 4), torch.float32, zero=True)

    def _stem_unpack_grad(self, arena, stream):
        """dW of the real 7x7 stem weight += the gradient of its space-to-depth image (after the table launch)."""
        w = self.stem.conv.weight
        sk, sc, sr, ss = w.stride()
        rc = self.lib.mi355_stem_s2d_unpack_grad(ptr(self.stem_dwp), ptr(self.grad_view(w, arena)), c_i64(sk), c_i64(sc),
                                                 c_i64(sr), c_i64(ss), c_int(self.stem.K), c_vp(stream))
        if rc:
            raise _lib.Mi355Error('stem_s2d_unpack_grad: %s' % self.lib.mi355_last_error().decode())

    def _alloc_params(self):
        """bf16 shadow arena, folded-BN arena, raw-wgrad scratch arena, and the flat parameter-
        gradient arena (one tensor: a single RCCL all-reduce covers every gradient)."""
        params = list(self.model.parameters())
        self.params = params
        offs, total = {}, 0
        for p in params:
            offs[id(p)] = total
            total += p.numel()
        self.grad_offsets = offs
        self.grad_numel = total
        self.grad_numel_padded = (total + 3) // 4 * 4
        if GUARD_BYTES:
            self.grad_arenas = [self._new_guarded((self.grad_numel_padded,), torch.float32, True) for _ in range(2)]
        else:
            self.grad_arenas = [torch.zeros(self.grad_numel_padded, dtype=torch.float32, device=self.dev)
                                for _ in range(2)]
        self.active = 0
        shadow = sum(s.K * s.T * s.Cpad + (s.C * s.T * s.Kpad if s.need_dgrad else 0) for s in self.specs)
        self.shadow = self.new((shadow + 64,), self.adt, zero=True)
        nbn = sum(2 * s.K for s in self.specs if s.folded_bn is not None)
        self.bnbuf = self.new((max(nbn, 1),), torch.float32)
        so, bo = 0, 0
        scratch = 0
        for s in self.specs:
            n = s.K * s.T * s.Cpad
            s.w_fwd = self.shadow[so:so + n].view(s.K, s.T, s.Cpad)
            so += (n + 7) // 8 * 8
            if s.need_dgrad:
                n = s.C * s.T * s.Kpad
                s.w_dgr = self.shadow[so:so + n].view(s.C, s.T, s.Kpad)
                so += (n + 7) // 8 * 8
            if s.folded_bn is not None:
                s.scale = self.bnbuf[bo:bo + s.K]
                s.shift = self.bnbuf[bo + s.K:bo + 2 * s.K]
                bo += 2 * s.K
            s.direct = (not self.deterministic and s.bn is None and s.Cpad == s.C and s.Kpad == s.K)
            if not s.direct and not self.deterministic:
                scratch += s.Kpad * s.T * s.Cpad + s.Kpad
        self.scratch = self.new((max(scratch, 1),), torch.float32, zero=True)
        self._carve_grads(0)

    def grad_view(self, p, arena):
        off = self.grad_offsets[id(p)]
        flat = self.grad_arenas[arena][off:off + p.numel()]
        if p.dim() == 4:        # same KRSC memory order and the SAME strides as the parameter
            return flat.as_strided(p.shape, p.stride())
        return flat.view(p.shape)

    def _carve_grads(self, arena):
        """Point every layer's wgrad / colsum target at `arena` (or the scratch arena)."""
        so = 0
        for s in self.specs:
            w = s.conv.weight
            if s.direct:
                off = self.grad_offsets[id(w)]
                s.G = self.grad_arenas[arena][off:off + w.numel()]
                boff = self.grad_offsets[id(s.conv.bias)]
                s.colsum = self.grad_arenas[arena][boff:boff + s.K]
            elif not self.deterministic:              # (slab mode: _wgrad allocates the slabs of each layer)
                n = s.Kpad * s.T * s.Cpad
                s.G = self.scratch[so:so + n]
                s.colsum = self.scratch[so + n:so + n + s.Kpad]
                so += n + s.Kpad

    # ------------------------------------------------------------------ prep (per step)
    def _layer_desc(self, s, arena=None):
        """Mi355LayerDesc of one conv layer (prep fields; gradient fields when `arena` is given)."""
        d = ops_conv.LayerDesc()
        w = s.conv.weight
        s2d = getattr(s, 's2d', False)
        if s2d:                                # the derived [K][12][4][4] image of the 7x7 stem weight
            d.w = self.stem_wp.data_ptr()
            d.sk, d.sc, d.sr, d.ss = self.stem_wp.stride()
        else:
            d.w = w.data_ptr()
            d.sk, d.sc, d.sr, d.ss = w.stride()
        d.K, d.C, d.R, d.S, d.Cpad, d.Kpad = s.K, s.C, s.R, s.S, s.Cpad, s.Kpad
        d.f32_shadows = 1 if self.prec == 'fp32' else 0
        bn = s.folded_bn                      # (train-mode BN is not folded: plain weights, dW = G)
        if bn is not None:
            d.gamma, d.beta = bn.weight.data_ptr(), bn.bias.data_ptr()
            d.mean, d.var, d.eps = bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.eps
            d.conv_bias = s.conv.bias.data_ptr() if s.conv.bias is not None else 0
            d.scale, d.shift = s.scale.data_ptr(), s.shift.data_ptr()
        d.w_fwd = s.w_fwd.data_ptr()
        d.w_dgr = s.w_dgr.data_ptr() if s.w_dgr is not None else 0
        if arena is not None:
            d.G, d.colsum = s.G.data_ptr(), s.colsum.data_ptr()
            if self.deterministic:
                d.nslabs, d.slab_stride = s.nslabs, s.Kpad * s.T * s.Cpad
            d.dW = self.stem_dwp.data_ptr() if s2d else self.grad_view(w, arena).data_ptr()
            if bn is not None:
                d.dgamma = self.grad_view(bn.weight, arena).data_ptr()
                d.dbeta = self.grad_view(bn.bias, arena).data_ptr()
                d.dconv_bias = self.grad_view(s.conv.bias, arena).data_ptr() if s.conv.bias is not None else 0
            elif s.conv.bias is not None:
                d.dbeta = self.grad_view(s.conv.bias, arena).data_ptr()
        return d

    def _build_prep(self):
        """ONE launch: frozen-BN fold + bf16 forward / dgrad weight shadows of every conv layer."""
        descs, nb = [], 0
        if getattr(self, 'stem_s2d', False):
            w = self.stem.conv.weight
            sk, sc, sr, ss = w.stride()
            self._step(self.prep_steps, 'mi355_stem_s2d_pack', ptr(w), c_i64(sk), c_i64(sc), c_i64(sr), c_i64(ss),
                       c_int(self.stem.K), ptr(self.stem_wp))
        for s in self.specs:
            d = self._layer_desc(s)
            d.block_begin = nb
            nb += (-(-s.Cpad // 64)) * (-(-s.Kpad // 32)) * s.R * s.S      # blocks of (tap, 32 k, 64 c), as the kernel counts them
            descs.append(d)
        self.prep_table = ops_conv.upload_table(descs, self.dev)
        self.bufs.append(self.prep_table)
        self._step(self.prep_steps, 'mi355_prep_weights_table', ptr(self.prep_table), c_int(len(descs)), c_int(nb))

    # ------------------------------------------------------------------ launch builders
    def _gather(self, lst, problems, weight, K, T, bias=None, act=ACT_NONE, out_f32=False, flags=0, stats=None):
        """stats: fp32 [ceil(M / 128)][2][K] - the launch also leaves the per-tile column statistics of its output there
        (Mi355ConvLaunch.stats: one problem, bf16 output, no split-K)."""
        assert len(problems) <= ops_conv.MAX_PROBLEMS
        L = ConvLaunch()
        for j, pr in enumerate(problems):
            L.prob[j] = pr
        L.nprob, L.K, L.weight, L.T = len(problems), K, weight.data_ptr(), T
        L.act, L.bias, L.out_f32, L.flags = act, (0 if bias is None else bias.data_ptr()), int(out_f32), flags
        L.stats = 0 if stats is None else stats.data_ptr()
        self.keep.append(L)
        ksplit = 1 if (self.prec == 'fp32' or stats is not None) else self._pick_ksplit(problems, K, T)
        if ksplit > 1:
            need = ksplit * sum(p.N * p.P * p.Q for p in problems) * K
            ws = self.new((need,), torch.float32)                 # partial-tile slabs of this launch
            self._step(lst, 'mi355_conv_gather_splitk', ctypes.byref(L), c_int(ksplit), ptr(ws))
        else:
            self._step(lst, 'mi355_conv_gather', ctypes.byref(L))

    @staticmethod
    def _pick_ksplit(problems, K, T):
        """Split the taps x channels reduction over blocks when the output has too few tiles to cover the 256 CUs and
        the reduction is long (P6 / P7 convs).  From 64 tiles of 128 x 128 (= 128 of the 128 x 64 tiles the launch
        then runs with) the unsplit three-stage ring kernel is as fast or faster than split + finish (16 x 16 x 16
        maps of R50 at batch 16: 2048 -> 512 1x1 23 vs 32 us, 256 -> 256 3x3 20 vs 27 us, 512 -> 512 3x3 42 vs 42 us;
        MI355_SPLITK_MAX_TILES)."""
        if os.environ.get('MI355_SPLITK', '1') == '0':
            return 1
        C = problems[0].C
        if C % 64:
            return 1
        tiles = sum(-(-(p.N * p.P * p.Q) // 128) for p in problems) * (-(-K // 128))
        ksteps = min(p.TR * p.TS for p in problems) * (C // 64)
        if tiles >= int(os.environ.get('MI355_SPLITK_MAX_TILES', '64')) or ksteps < 16:
            return 1
        return max(1, min(ksteps // 8, 384 // tiles))

    def _wgrad(self, problems, spec, rows, colsum=True):
        """Weight gradient, split over pixel chunks.  'slabs': every chunk stores its partial dW tile
        (and dy column sums) with plain stores into its own slab of the layer's workspace; the
        parameter-gradient table launch sums the slabs in index order while it folds the frozen BN
        (plain stores run ~5x the fp32-atomic rate on gfx950: tools/probe_atomics.hip) and the
        gradients are bit-reproducible.  'atomic': fp32 atomics straight into the gradient."""
        L = ConvLaunch()
        for j, pr in enumerate(problems):
            L.prob[j] = pr
        L.nprob, L.K, L.T = len(problems), spec.Kpad, spec.T
        slab = spec.Kpad * spec.T * spec.Cpad
        tiles = ops_conv.wgrad_tiles(spec.Kpad, spec.Cpad, spec.T, problems)
        split = pick_split_rows(rows, tiles, target_blocks=ops_conv.wgrad_target_blocks(spec.Kpad, spec.Cpad, spec.T, problems),
                                min_rows=int(os.environ.get('MI355_WG_MINROWS', '1024')),
                                granule=int(os.environ.get('MI355_WG_GRANULE', '64')))
        split = self._fit_ring_split(problems, spec, tiles, split)
        self.keep.append(L)
        spec._wgrad_launch = L
        spec._wgrad_end = len(self.bwd_steps) + 1          # launches of the backward list needed for this dW
        if not self.deterministic:
            self._step(self.bwd_steps, 'mi355_conv_wgrad_colsum', ctypes.byref(L), ptr(spec.G),
                       ptr(spec.colsum if colsum else None), c_int(split))
            return
        nslabs = sum(-(-(p.N * p.P * p.Q) // split) for p in problems)
        if spec.G is None or spec.nslabs != nslabs:
            spec.nslabs = nslabs
            spec.G = self.new((nslabs, slab), torch.float32)           # every element is written each step
            spec.colsum = self.new((nslabs, spec.Kpad), torch.float32, zero=True)
        self._step(self.bwd_steps, 'mi355_conv_wgrad_slabs', ctypes.byref(L), ptr(spec.G), c_i64(slab),
                   c_int(nslabs), ptr(spec.colsum if colsum else None), c_int(split))

    def _fit_ring_split(self, problems, spec, tiles, split):
        """MI355_WG_X_FIT=1: launches of the eight-wave ring kernel (one block per CU) cut so that every block runs in ONE
        round (ops_conv.fit_split_rows).  Measured and OFF: the head towers went 140 -> 148 us and the cls output 386 -> 573
        us alone, the step lost 2 % - the launch ends with its LONGEST chunk (which the fit lengthens from 65 to 74
        stages), and the 20 blocks past 256 are the shortest chunks of the longest-first order, which the CUs of the
        short first-round chunks pick up for free."""
        if (len(problems) > 1 and spec.T == 9 and os.environ.get('MI355_WG_X_FIT', '0') == '1' and
                ops_conv.wgrad_row3x_ok(problems, spec.Cpad, spec.Kpad)):
            cap = int(os.environ.get('MI355_WG_X_BLOCKS', '256'))
            return ops_conv.fit_split_rows([p.N * p.P * p.Q for p in problems], tiles, cap, split,
                                           int(os.environ.get('MI355_WG_GRANULE', '64')))
        return split

    # -- same-shape weight gradients of a stage as ONE launch ---------------------------------------------------------
    # The identical bottleneck / basic blocks of a ResNet stage produce weight gradients of the same shape that are
    # independent leaves of the backward pass; alone each is a 25-50 us launch that cannot fill 256 CUs (layer3 conv2
    # of R50 512^2 bs16: 44 us = 437 TFLOP/s alone, 27 us per block = 724 TFLOP/s when five blocks share a launch:
    # tools/bench_wgrad_group.py).  conv_bwd defers them; the stage's end (or five of a kind) flushes a group.
    def _defer_wgrad(self, pr, spec, rows, x, dy):
        # (a launch needs one K, C and tap geometry; image sizes may differ per problem, as in the five-level head launches)
        key = (spec.Kpad, spec.Cpad, spec.R, spec.S, spec.stride, spec.pad)
        group = self._pending_wgrads.setdefault(key, [])
        group.append((pr, spec, rows))
        if len(group) == ops_conv.MAX_PROBLEMS:
            self._flush_wgrads(key)

    def _flush_wgrads(self, key=None):
        for k in ([key] if key is not None else list(self._pending_wgrads)):
            group = self._pending_wgrads.pop(k, [])
            if len(group) == 1:
                self._wgrad([group[0][0]], group[0][1], group[0][2])
            elif group:
                self._wgrad_multi(group)

    def _wgrad_multi(self, group):
        spec0 = group[0][1]
        problems = [g[0] for g in group]
        L = ConvLaunch()
        for j, pr in enumerate(problems):
            L.prob[j] = pr
        L.nprob, L.K, L.T = len(problems), spec0.Kpad, spec0.T
        slab = spec0.Kpad * spec0.T * spec0.Cpad
        tiles = ops_conv.wgrad_tiles(spec0.Kpad, spec0.Cpad, spec0.T, problems)
        split = pick_split_rows(sum(g[2] for g in group), tiles,
                                target_blocks=ops_conv.wgrad_target_blocks(spec0.Kpad, spec0.Cpad, spec0.T, problems),
                                min_rows=int(os.environ.get('MI355_WG_MINROWS', '1024')),
                                granule=int(os.environ.get('MI355_WG_GRANULE', '64')))
        split = self._fit_ring_split(problems, spec0, tiles, split)
        n = len(group)
        ws, cs, cap = (c_vp * n)(), (c_vp * n)(), (c_int * n)()
        for j, (pr, spec, rows) in enumerate(group):
            nslabs = -(-rows // split)
            if spec.G is None or spec.nslabs != nslabs:
                spec.nslabs = nslabs
                spec.G = self.new((nslabs, slab), torch.float32)
                spec.colsum = self.new((nslabs, spec.Kpad), torch.float32, zero=True)
            ws[j], cs[j], cap[j] = spec.G.data_ptr(), spec.colsum.data_ptr(), nslabs
            spec._wgrad_launch = L
            spec._wgrad_end = len(self.bwd_steps) + 1
        self.keep.append((L, ws, cs, cap))
        self._step(self.bwd_steps, 'mi355_conv_wgrad_multi', ctypes.byref(L), ws, c_i64(slab), cap, cs, c_int(split))

    def conv(self, spec, x, act=ACT_NONE, addend=None, add_shift=0):
        N, H, W, C = x.shape
        assert C == spec.Cpad, (spec.name, C, spec.Cpad)
        pt, pl, P, Q = spec.geometry(H, W)
        out = self.new((N, P, Q, spec.Kpad), zero=spec.Kpad != spec.K)
        if spec.bn_train:
            # conv -> raw z; batch statistics; y = act(BN(z) + residual).  (K of a BN layer is a multiple of 8.)
            assert spec.Kpad == spec.K and add_shift == 0, spec.name
            pr = fwd_problem(x, out.data_ptr(), spec.R, spec.S, spec.stride, (pt, pl), K=spec.K, out_size=(P, Q))
            # the statistics of z in the conv's own epilogue (one [sum | square-sum] row per 128-pixel tile, finished by the
            # same reduce launch as the rows of the separate pass): one pass over z less per BN layer (MI355_BN_EPILOGUE=0)
            tiles = None
            # (reductions of whole 64-channel steps only: the short flat ones may take the streaming 1x1 form, whose epilogue
            #  is its own)
            if self.prec == 'bf16' and os.environ.get('MI355_BN_EPILOGUE', '1') == '1' and spec.Cpad % 64 == 0:
                ntile = -(-(N * P * Q) // 128)
                tiles = self.new((ntile, 2 * spec.K), torch.float32)
            self._gather(self.fwd_steps, [pr], spec.w_fwd, spec.K, spec.T, bias=spec.conv.bias, stats=tiles)
            self.fwd_launches.append((spec, [(N, P, Q)]))
            return self.bn_fwd(spec, spec.bn, out, act, addend, tiles=tiles)
        pr = fwd_problem(x, out.data_ptr(), spec.R, spec.S, spec.stride, (pt, pl), K=spec.K,
                         addend=addend, add_shift=add_shift, out_size=(P, Q),
                         out_strides=(P * Q * spec.Kpad, Q * spec.Kpad, spec.Kpad))
        self._gather(self.fwd_steps, [pr], spec.w_fwd, spec.K, spec.T, bias=spec.bias_tensor, act=act)
        self.fwd_launches.append((spec, [(N, P, Q)]))
        return out

    # -- same-shape convs of different layers as ONE launch (per-problem weights, Mi355ConvProblem.weight / .bias) ----
    # The three 3x3 smoothing convs of the pyramid (retinanet.py:53, :57, :61) are independent once the laterals exist;
    # alone the 32^2 and 16^2 ones are 27 / 20 us launches of 128 / 32 tiles that cannot fill 256 CUs.
    def _group_convs_ok(self, specs):
        s0 = specs[0]
        return (os.environ.get('MI355_FPN_GROUP', '1') == '1' and self.prec == 'bf16' and 2 <= len(specs) <= ops_conv.MAX_PROBLEMS and
                type(self).conv is RetinaPlan.conv and type(self).conv_bwd is RetinaPlan.conv_bwd and
                all((sp.K, sp.Kpad, sp.Cpad, sp.R, sp.S, sp.stride, sp.pad) == (s0.K, s0.Kpad, s0.Cpad, s0.R, s0.S, s0.stride, s0.pad)
                    and not sp.bn_train and not getattr(sp, 's2d', False) for sp in specs))

    def conv_group(self, specs, xs, act=ACT_NONE):
        """conv(spec_i, x_i) for every i in one grouped gather launch (largest map first)."""
        if not self._group_convs_ok(specs):
            return [self.conv(sp, x, act) for sp, x in zip(specs, xs)]
        outs, probs = [], []
        for sp, x in zip(specs, xs):
            N, H, W, C = x.shape
            assert C == sp.Cpad, (sp.name, C, sp.Cpad)
            pt, pl, P, Q = sp.geometry(H, W)
            out = self.new((N, P, Q, sp.Kpad), zero=sp.Kpad != sp.K)
            pr = fwd_problem(x, out.data_ptr(), sp.R, sp.S, sp.stride, (pt, pl), K=sp.K, out_size=(P, Q),
                             out_strides=(P * Q * sp.Kpad, Q * sp.Kpad, sp.Kpad))
            pr.weight = sp.w_fwd.data_ptr()
            pr.bias = 0 if sp.bias_tensor is None else sp.bias_tensor.data_ptr()
            outs.append(out)
            probs.append(pr)
            self.fwd_launches.append((sp, [(N, P, Q)]))
        order = sorted(range(len(probs)), key=lambda i: -xs[i].shape[1] * xs[i].shape[2])
        s0 = specs[order[0]]
        self._gather(self.fwd_steps, [probs[i] for i in order], s0.w_fwd, s0.K, s0.T, bias=s0.bias_tensor, act=act)
        return outs

    def conv_bwd_group(self, specs, xs, dys):
        """conv_bwd(spec_i, x_i, dy_i) with the data gradients of all i in one grouped launch -> [dx_i]."""
        if not self._group_convs_ok(specs) or not all(sp.need_dgrad for sp in specs):
            return [self.conv_bwd(sp, x, dy) for sp, x, dy in zip(specs, xs, dys)]
        dxs, dprobs = [], []
        for sp, x, dy in zip(specs, xs, dys):
            N, H, W, C = x.shape
            pt, pl, P, Q = sp.geometry(H, W)
            pr = fwd_problem(x, dy.data_ptr(), sp.R, sp.S, sp.stride, (pt, pl), K=sp.Kpad, out_size=(P, Q))
            if self._grouping_wgrads:
                self._defer_wgrad(pr, sp, N * pr.P * pr.Q, x, dy)
            else:
                self._wgrad([pr], sp, N * pr.P * pr.Q)
            dx = self.new((N, H, W, sp.Cpad))
            ps = dgrad_problems(dy, dx.data_ptr(), H, W, sp.Cpad, sp.R, sp.S, sp.stride, pt)
            assert len(ps) == 1 and pt == pl
            ps[0].weight, ps[0].bias = sp.w_dgr.data_ptr(), 0
            dprobs.append(ps[0])
            dxs.append(dx)
        order = sorted(range(len(dprobs)), key=lambda i: -xs[i].shape[1] * xs[i].shape[2])
        s0 = specs[order[0]]
        self._gather(self.bwd_steps, [dprobs[i] for i in order], s0.w_dgr, s0.Cpad, s0.T)
        return dxs

    # ------------------------------------------------------------------ train-mode / synchronised BatchNorm
    def _pystep(self, lst, fn, name):
        """A host-side step inside a launch list (the cross-replica all-reduce of BN statistics): called as
        fn(stream) in list order, on the stream the neighbouring launches use."""
        def call(stream, fn=fn):
            fn(stream)
            return 0
        lst.append((call, (), name))

    def _bn_sync_group(self, bn):
        """(synchronise?, process group, replicas): statistics are exchanged for the drop-in
        SynchronizedBatchNorm2d (utils/sync_batchnorm/batchnorm.py:63-93) when a process group with more than one
        rank exists; a plain nn.BatchNorm2d in train mode normalises with its replica's own statistics."""
        import torch.distributed as dist
        from .utils.sync_batchnorm.batchnorm import SynchronizedBatchNorm2d
        if isinstance(bn, SynchronizedBatchNorm2d) and dist.is_available() and dist.is_initialized():
            world = dist.get_world_size(bn.group)
            # (MI355_BN_FORCE_ALLREDUCE=1: exchange on a world-size-1 group too, to price the collectives of
            #  BASELINE configs[2] against the real RCCL stream machinery on one GPU: bench.py --sync-bn)
            if world > 1 or os.environ.get('MI355_BN_FORCE_ALLREDUCE') == '1':
                return True, bn.group, world
        # (tests: `bn.force_sync_formula` puts ONE replica on the synchronised normaliser, so that a single process on
        #  the full batch can be compared with the replicas; world 1 -> no exchange)
        return bool(getattr(bn, 'force_sync_formula', False)), None, 1

    def _bn_allreduce(self, buf, group):
        import torch.distributed as dist

        def run(stream, buf=buf, group=group):
            dist.all_reduce(buf, group=group)       # SUM of [sum | square-sum] (batchnorm.py:118: ReduceAddCoalesced)
        return run

    def _rccl_comm(self, group):
        """ncclComm_t of `group` on this device when its backend is RCCL (torch: ProcessGroupNCCL._comm_ptr()), else
        None.  The communicator is made by the first collective on the device: one 1-element all-reduce at build time."""
        cache = self.__dict__.setdefault('_rccl_comms', {})
        key = id(group)
        if key not in cache:
            comm = None
            try:
                import torch.distributed as dist
                if os.environ.get('MI355_BN_NATIVE_RCCL', '1') != '0' and dist.get_backend(group) == 'nccl':
                    dist.all_reduce(torch.zeros(1, device=self.dev), group=group)
                    torch.cuda.synchronize(self.dev)
                    pg = group if group is not None else dist.distributed_c10d._get_default_group()
                    comm = int(pg._get_backend(self.dev)._comm_ptr()) or None
            except Exception:                                   # (gloo, an older torch without _comm_ptr: host steps)
                comm = None
            cache[key] = comm
        return cache[key]

    def _bn_exchange_step(self, lst, buf, group):
        """The cross-replica sum of a layer's statistics vector inside a launch list: a TABLE step on the launch stream
        when the group is an RCCL one (mi355_rccl_allreduce_sum_f32: no return to Python per layer and direction, and
        capturable), else a host step that calls torch.distributed (gloo in the CPU-side tests)."""
        comm = self._rccl_comm(group)
        if comm:
            self._step(lst, 'mi355_rccl_allreduce_sum_f32', c_vp(comm), ptr(buf), c_i64(buf.numel()))
        else:
            self._pystep(lst, self._bn_allreduce(buf, group), 'py:bn_allreduce')

    def bn_fwd(self, spec, bn, z, act, addend, tag=None, tiles=None):
        """Batch statistics of z [N,P,Q,C] -> layer state -> y = act(z*scale + shift (+ addend)).  Records what the
        backward needs under `tag` (default: the spec) in self.bn_saved.  tiles: the producing conv already left the
        partial [sum | square-sum] rows there (one per 128-pixel tile): no statistics pass of its own."""
        if bn.momentum is None or not bn.affine:
            raise _lib.Mi355Error('train-mode BN in the launch plan needs affine=True and a float momentum')
        C = z.shape[-1]
        M = z.numel() // C
        nblk, rpb = c_int(), c_int()
        _lib.call('mi355_bn_nhwc_partial_blocks', c_i64(M), c_int(C), ctypes.byref(nblk), ctypes.byref(rpb))
        partials = self.new((nblk.value, 2 * C), torch.float32)
        sync, group, world = self._bn_sync_group(bn)
        # exchanged statistics carry each replica's own element count (batchnorm.py:105-126 adds the actual sizes: the
        # last batch of an epoch, or a DataParallel scatter that does not divide, gives the replicas different N)
        counted = sync and group_exchange(world)
        sums = self.new((2 * C + (2 if counted else 0),), torch.float32)
        state = self.new((5 * C + (1 if counted else 0),), torch.float32)
        f32 = c_int(1 if self.prec == 'fp32' else 0)
        lst = self.fwd_steps
        fpart, fblk = partials, nblk.value
        if tiles is not None:
            fpart, fblk = tiles, tiles.shape[0]
        else:
            self._step(lst, 'mi355_bn_nhwc_stats', ptr(z), c_vp(0), c_vp(0), f32, c_i64(M), c_int(C), c_int(0), ptr(partials))
        if counted:
            self._step(lst, 'mi355_bn_nhwc_reduce_counted', ptr(fpart), c_int(fblk), c_int(C), c_i64(M), ptr(sums),
                       c_vp(0), c_vp(0))
            self._bn_exchange_step(lst, sums, group)
        else:
            self._step(lst, 'mi355_bn_nhwc_reduce', ptr(fpart), c_int(fblk), c_int(C), ptr(sums), c_vp(0), c_vp(0))
        track = bn.track_running_stats and bn.running_mean is not None
        plain = type(bn) is torch.nn.BatchNorm2d            # nn.BatchNorm2d.forward counts batches; the sync class does not
        self._step(lst, 'mi355_bn_nhwc_finalize', ptr(sums), c_f32(0.0 if counted else float(M) * world), c_int(C), c_f32(bn.eps),
                   c_f32(bn.momentum), c_int(1 if sync else 0), ptr(bn.weight), ptr(bn.bias),
                   ptr(bn.running_mean if track else None), ptr(bn.running_var if track else None),
                   ptr(bn.num_batches_tracked if (track and plain) else None), ptr(state))
        y = self.new(tuple(z.shape))
        self._step(lst, 'mi355_bn_nhwc_apply', ptr(z), ptr(state), ptr(addend), c_int(1 if act == ACT_RELU else 0), f32,
                   c_i64(M), c_int(C), ptr(y))
        assert act in (ACT_NONE, ACT_RELU)
        self.bn_saved[id(spec) if tag is None else tag] = (bn, z, state, partials, M, C, sync, group, world)
        return y

    def bn_bwd(self, spec, g, tag=None):
        """g = dL/d(BN output, pre-activation) -> dz; dgamma / dbeta (local sums) go straight into the arena."""
        bn, z, state, partials, M, C, sync, group, world = self.bn_saved[id(spec) if tag is None else tag]
        arena = getattr(self, '_build_arena', 0)
        gsums = self.new((2 * C,), torch.float32)
        f32 = c_int(1 if self.prec == 'fp32' else 0)
        lst = self.bwd_steps
        self._step(lst, 'mi355_bn_nhwc_stats', ptr(g), ptr(z), ptr(state), f32, c_i64(M), c_int(C), c_int(1), ptr(partials))
        self._step(lst, 'mi355_bn_nhwc_reduce', ptr(partials), c_int(partials.shape[0]), c_int(C), ptr(gsums),
                   ptr(self.grad_view(bn.weight, arena)), ptr(self.grad_view(bn.bias, arena)))
        if sync and group_exchange(world):
            self._bn_exchange_step(lst, gsums, group)
        dz = self.new(tuple(g.shape))
        counted = sync and group_exchange(world)            # (1 / the exchanged element count sits in state[5C])
        self._step(lst, 'mi355_bn_nhwc_bwd_apply', ptr(g), ptr(z), ptr(state), ptr(gsums),
                   c_f32(0.0 if counted else 1.0 / (float(M) * world)),
                   f32, c_i64(M), c_int(C), ptr(dz))
        return dz

    def conv_bwd(self, spec, x, dy, dx=None, addend=None, mask=None, mask_first=False, dgrad=True):
        """wgrad (+ column sums) of `spec` and, optionally, its data gradient into `dx`."""
        N, H, W, C = x.shape
        pt, pl, P, Q = spec.geometry(H, W)
        if spec.bn_train:                                  # dy is the gradient of the BN output: through the BN first
            dy = self.bn_bwd(spec, dy)
        pr = fwd_problem(x, dy.data_ptr(), spec.R, spec.S, spec.stride, (pt, pl), K=spec.Kpad, out_size=(P, Q))
        if self._grouping_wgrads:
            self._defer_wgrad(pr, spec, N * pr.P * pr.Q, x, dy)
        else:
            self._wgrad([pr], spec, N * pr.P * pr.Q)      # + fused column sums (bias / beta grads)
        if not dgrad:
            return None
        if dx is None:
            dx = self.new((N, H, W, spec.Cpad))
        assert pt == pl, 'asymmetric-pad dgrad is only needed by the stem, which has no dgrad'
        probs = dgrad_problems(dy, dx.data_ptr(), H, W, spec.Cpad, spec.R, spec.S, spec.stride, pt,
                               addend=addend, mask=mask)
        for i in range(0, len(probs), ops_conv.MAX_PROBLEMS):        # stride 3: 9 output-parity classes
            self._gather(self.bwd_steps, probs[i:i + ops_conv.MAX_PROBLEMS], spec.w_dgr, spec.Cpad, spec.T,
                         flags=1 if mask_first else 0)
        return dx

    # ------------------------------------------------------------------ forward
    def _build_forward(self):
        x = self._stem_fwd()
        feats = self._stages_fwd(x)
        self._fpn_fwd(feats)                       # -> self.levels
        self._heads_fwd()

    def _stem_fwd(self):
        B, H, W = self.B, self.H, self.W
        self.img_nhwc = self.new((B, H // 2, W // 2, 16)) if self.stem_s2d else self.new((B, H, W, 8))
        x = self.conv(self.stem, self.img_nhwc, ACT_RELU)
        self.c1 = x
        if not self.has_maxpool:                   # RetinaNetTiny: retinanet.py:583-586
            return x
        P, Q = (x.shape[1] - 1) // 2 + 1, (x.shape[2] - 1) // 2 + 1
        self.mp = self.new((B, P, Q, 64))
        self.mp_idx = self.new((B, P, Q, 64), torch.int8)
        self._step(self.fwd_steps, 'mi355_maxpool3x3s2_fwd', ptr(x), ptr(self.mp), ptr(self.mp_idx),
                   c_int(B), c_int(x.shape[1]), c_int(x.shape[2]), c_int(64))
        return self.mp

    def _stages_fwd(self, x):
        feats = []
        # The downsample conv of a stage's first block (utils_resnet.py:72-73) only depends on the block input: it runs on
        # the side stream - idle in the forward until the box head - beside conv1 / conv2, and the join edge in front of
        # the conv that adds it as the residual orders main behind it.  MI355_DS_SIDE=1 (off by default: measured within +-0.3 %
        # of the in-line order at 512^2 and 640^2 - the side stream's launch only moves bytes the chain waits for anyway).  Frozen BN only: the
        # statistics passes of a train-mode BN would pull the whole chain over.
        ds_side = os.environ.get('MI355_DS_SIDE', '0') == '1' and self.use_side and not self._mark_side

        def downsample(rec, x):
            if rec['ds'] is None:
                return x, False
            side = ds_side and not rec['ds'].bn_train
            first = len(self.fwd_steps)
            res = self.conv(rec['ds'], x)
            if side:
                for st in self.fwd_steps[first:]:
                    self._side_ids.add(id(st[1]))
            return res, side
        for rec in self.blocks:
            rec['x'] = x
            res, joined = downsample(rec, x)
            if rec['kind'] == 'bottleneck':
                rec['a'] = self.conv(rec['c1'], x, ACT_RELU)
                rec['b'] = self.conv(rec['c2'], rec['a'], ACT_RELU)
                if joined:
                    self._join_step(self.fwd_steps)
                rec['out'] = self.conv(rec['c3'], rec['b'], ACT_RELU, addend=res)
            else:
                rec['a'] = self.conv(rec['c1'], x, ACT_RELU)
                if joined:
                    self._join_step(self.fwd_steps)
                rec['out'] = self.conv(rec['c2'], rec['a'], ACT_RELU, addend=res)
            x = rec['out']
            if rec['stage_end']:
                feats.append(x)
        return feats

    def _fpn_fwd(self, feats):
        """retinanet.py:48-69."""
        c3, c4, c5 = feats[1], feats[2], feats[3]
        self.c3, self.c4, self.c5 = c3, c4, c5
        f = self.fpn
        # P6 = conv(C5) -> ReLU -> P7 only depends on C5: three short launches (split-K convs over 8^2 / 4^2 maps) that
        # run on the side stream beside the laterals and the smoothing convs; the join edge in front of the heads orders
        # main behind them (MI355_FPN_SIDE=0: in line, after the smoothing convs as before)
        side_branch = os.environ.get('MI355_FPN_SIDE', '1') == '1' and self.use_side and not self._mark_side

        def extra_levels():
            first = len(self.fwd_steps)
            self.p6 = self.conv(f['P6'], c5)
            self.p6relu = self.new(tuple(self.p6.shape))
            self._step(self.fwd_steps, 'mi355_relu_mask_bf16', ptr(self.p6), ptr(self.p6), ptr(self.p6relu),
                       c_i64(self.p6.numel()))
            self.p7 = self.conv(f['P7_2'], self.p6relu)
            if side_branch:
                for st in self.fwd_steps[first:]:
                    self._side_ids.add(id(st[1]))
        if side_branch:
            extra_levels()
        self.p5lat = self.conv(f['P5_1'], c5)
        self.p4lat = self.conv(f['P4_1'], c4, addend=self.p5lat, add_shift=1)
        self.p3lat = self.conv(f['P3_1'], c3, addend=self.p4lat, add_shift=1)
        self.p3, self.p4, self.p5 = self.conv_group([f['P3_2'], f['P4_2'], f['P5_2']], [self.p3lat, self.p4lat, self.p5lat])
        if side_branch:
            self._join_step(self.fwd_steps)
        else:
            extra_levels()
        self.levels = [self.p3, self.p4, self.p5, self.p6, self.p7]

    def _heads_fwd(self):
        B = self.B
        self.cells = [t.shape[1] * t.shape[2] for t in self.levels]
        self.cell_off = [sum(self.cells[:i]) for i in range(len(self.levels))]
        self.ncells = sum(self.cells)
        self.A = self.ncells * self.A_cell
        C = self.num_classes
        self.reg = self.new((B, self.A, 4), torch.float32)
        self.cls = self.new((B, self.A, C), torch.float32)
        # the box head only shares the pyramid with the class head: its launches go to the side stream
        # (when enabled) and run beside the class tower; run_heads joins before the loss
        self._mark_side = True
        self.reg_t = self._tower_fwd(self.reg_tower)
        self._head_out_fwd(self.reg_out, self.reg_t[-1], self.reg, 4 * self.A_cell, ACT_NONE)
        self._mark_side = False
        self.cls_t = self._tower_fwd(self.cls_tower)
        self._head_out_fwd(self.cls_out, self.cls_t[-1], self.cls, C * self.A_cell, ACT_SIGMOID)
        # loss state
        self.losses = self.new((2,), torch.float32)
        self.argmax = self.new((B, self.A), torch.int32)
        self.state = self.new((B, self.A), torch.int8)
        self.num_pos = self.new((B,), torch.int32)
        nb = ctypes.c_int64()
        _lib.call('mi355_focal_loss_workspace_bytes', c_int(B), c_int(self.A), c_int(C), c_int(1),
                  ctypes.byref(nb))
        self.loss_ws = self.new((nb.value,), torch.uint8)
        self.loss_ws_bytes = nb.value
        self.glosses = self.new((2,), torch.float32)

    def _level_views(self, buf):
        """Per-level NHWC views of a level-major buffer [sum_l B*h_l*w_l, 256]."""
        views, off = [], 0
        for lv in self.levels:
            n = self.B * lv.shape[1] * lv.shape[2]
            views.append(buf[off:off + n].view(self.B, lv.shape[1], lv.shape[2], buf.shape[1]))
            off += n
        return views

    def _tower_fwd(self, tower):
        """4 x (conv3x3 + ReLU) with shared weights over the 5 levels: ONE grouped launch per conv
        (retinanet.py:92-104 runs 5 launch sets per conv)."""
        rows = self.B * self.ncells
        xs = self.levels
        outs = []
        for spec in tower:
            buf = self.new((rows, 256))
            ys = self._level_views(buf)
            probs = [fwd_problem(x, y.data_ptr(), 3, 3, 1, 1, K=256) for x, y in zip(xs, ys)]
            self._gather(self.fwd_steps, probs, spec.w_fwd, 256, 9, bias=spec.conv.bias, act=ACT_RELU)
            self.fwd_launches.append((spec, [(x.shape[0], x.shape[1], x.shape[2]) for x in xs]))
            outs.append(ys)
            xs = ys
        return outs

    def _head_out_fwd(self, spec, xs, out, kout, act):
        """Output conv writing fp32 straight into the [B, A, .] layout of retinanet.py:108-115 /
        :163-178 (NHWC is already (y, x, anchor, class) order: no permute/cat pass)."""
        per_img = self.ncells * kout
        probs = []
        for x, off in zip(xs, self.cell_off):
            w = x.shape[2]
            probs.append(fwd_problem(x, out.data_ptr() + 4 * off * kout, 3, 3, 1, 1,
                                     out_strides=(per_img, w * kout, kout), K=kout))
        self._gather(self.fwd_steps, probs, spec.w_fwd, kout, 9, bias=spec.conv.bias, act=act, out_f32=True)
        self.fwd_launches.append((spec, [(x.shape[0], x.shape[1], x.shape[2]) for x in xs]))

    # ------------------------------------------------------------------ backward
    def _wgrad_grouping_ok(self):
        return (os.environ.get('MI355_WG_GROUP', '1') == '1' and self.deterministic and self.prec == 'bf16' and
                type(self).conv_bwd is RetinaPlan.conv_bwd)

    def _build_backward(self):
        g_feat = self._heads_bwd()
        self._grouping_wgrads = self._wgrad_grouping_ok()      # the three 3x3 smoothing convs of the pyramid share a launch
        g, extra = self._fpn_bwd(g_feat)
        self._grouping_wgrads = False
        self._flush_wgrads()
        g = self._stages_bwd(g, extra)
        self._stem_bwd(g)
        self._build_param_grads()

    def _heads_bwd(self):
        B = self.B
        kc = self.cls_out.Kpad                     # 720 (9 anchors x 80 classes), padded to 8
        kr = self.reg_out.Kpad                     # 40 (36 padded)
        self.gcls = self.new((B, self.ncells, kc), zero=True)       # bf16 d/dlogit, cell rows
        self.greg = self.new((B, self.ncells, kr), zero=True)       # bf16, padded pitch
        self.cls_pitch, self.reg_pitch = kc, kr
        # output convs, then the towers (levels grouped), feature grads accumulated
        g_feat = [self.new(tuple(lv.shape)) for lv in self.levels]
        self._head_bwd(self.reg_out, self.reg_tower, self.reg_t, self.greg, kr, g_feat, first=True)
        self._head_bwd(self.cls_out, self.cls_tower, self.cls_t, self.gcls, kc, g_feat, first=False)
        return g_feat

    def _fpn_bwd(self, g_feat):
        """retinanet.py:48-69 reversed -> (gradient of the last stage output, {stage: lateral grad})."""
        g_p3, g_p4, g_p5, g_p6, g_p7 = g_feat
        f = self.fpn
        # RetinaNet: the heads read the raw P6 (retinanet.py:64-67, P7_1 = nn.ReLU() out of place): only the P7 branch
        # passes the ReLU mask.  KSevenDet's `F.relu(P6, inplace=True)` (neck/fpn.py:286) rectifies the tensor that is
        # ALSO handed to the heads: head gradient and P7 gradient are masked together (mask after the addend).
        # The P7 -> P6 -> C5 data gradients (short split-K launches over 4^2 / 8^2 / 16^2 maps) only meet the rest of the
        # pyramid's backward at C5: they run on the branch stream beside the smoothing / lateral gradients.  Their weight
        # gradients are deferred to the end of the pyramid (grouping on), i.e. behind the join: P6's reads g_p6.
        branch = (self.use_branch and self.use_side and self.n_side == 1 and self._grouping_wgrads and
                  os.environ.get('MI355_FPN_BRANCH', '1') == '1')
        first = len(self.bwd_steps)
        self.conv_bwd(f['P7_2'], self.p6relu, g_p7, dx=g_p6, addend=g_p6, mask=self.p6,
                      mask_first=not getattr(self, 'p6_relu_inplace', False))
        g_c5 = self.conv_bwd(f['P6'], self.c5, g_p6)
        if branch:
            assert all(st[2] not in self.SIDE_STEPS for st in self.bwd_steps[first:]), 'weight gradients were to be deferred'
            for st in self.bwd_steps[first:]:
                self._branch_ids.add(id(st[1]))
        g_p3lat, g_p4lat, g_p5lat = self.conv_bwd_group([f['P3_2'], f['P4_2'], f['P5_2']],
                                                        [self.p3lat, self.p4lat, self.p5lat], [g_p3, g_p4, g_p5])
        self._sumpool(g_p3lat, g_p4lat)
        g_c3 = self.conv_bwd(f['P3_1'], self.c3, g_p3lat)
        self._sumpool(g_p4lat, g_p5lat)
        g_c4 = self.conv_bwd(f['P4_1'], self.c4, g_p4lat)
        # c5 has no later consumer: finish it here (sum of P6 and P5_1 grads, ReLU of layer4)
        if branch:
            self._join_step(self.bwd_steps, branch=True)
        self.conv_bwd(f['P5_1'], self.c5, g_p5lat, dx=g_c5, addend=g_c5, mask=self.c5)
        return g_c5, {2: g_c4, 1: g_c3}            # raw FPN-lateral grads of the stage outputs

    def _stages_bwd(self, g, extra):
        """Backbone, last block first.  g = gradient w.r.t. the block output's pre-ReLU sum."""
        group = self._wgrad_grouping_ok()
        group_l1 = os.environ.get('MI355_WG_GROUP_L1', '1') == '1'     # (A/B: layer1's weight gradients issued block by block)
        for i in range(len(self.blocks) - 1, -1, -1):
            rec = self.blocks[i]
            self._grouping_wgrads = group and (group_l1 or rec['stage'] != 0)
            x = rec['x']
            first_block = (i == 0)
            x_extra = None
            if i > 0 and self.blocks[i - 1]['stage_end'] and self.blocks[i - 1]['stage'] in extra:
                x_extra = extra[self.blocks[i - 1]['stage']]
            # x = max-pool output for the very first block (its ReLU mask is applied by the pool backward)
            mask_x = None if (first_block and self.has_maxpool) else x
            # the data gradient of the downsample conv (79 - 95 us at R50 512^2: a 1x1 / stride-2 conv = four parity
            # classes) only meets the conv3 -> conv2 -> conv1 chain of its block at the final add: it is issued first and
            # runs on the branch stream beside that chain; main joins it in front of conv1's data gradient
            branch = (rec['ds'] is not None and self.use_branch and self.use_side and self.n_side == 1 and
                      not rec['ds'].bn_train)
            res_g = None
            if branch:
                first = len(self.bwd_steps)
                res_g = self.conv_bwd(rec['ds'], x, g, dx=x_extra, addend=x_extra)
                for st in self.bwd_steps[first:]:
                    if st[2] == 'mi355_conv_gather':
                        self._branch_ids.add(id(st[1]))
            if rec['kind'] == 'bottleneck':
                g_b = self.conv_bwd(rec['c3'], rec['b'], g, mask=rec['b'])
                g_a = self.conv_bwd(rec['c2'], rec['a'], g_b, mask=rec['a'])
                last = rec['c1']
            else:
                g_a = self.conv_bwd(rec['c2'], rec['a'], g, mask=rec['a'])
                last = rec['c1']
            if branch:
                self._join_step(self.bwd_steps, branch=True)
            elif rec['ds'] is not None:
                res_g = self.conv_bwd(rec['ds'], x, g, dx=x_extra, addend=x_extra)
            else:
                assert x_extra is None
                res_g = g
            g = self.conv_bwd(last, x, g_a, addend=res_g, mask=mask_x)
            self._grouping_wgrads = False
            if i == 0 or self.blocks[i - 1]['stage'] != rec['stage']:
                self._flush_wgrads()                       # the stage is done: its deferred weight gradients, grouped
        return g

    def _stem_bwd(self, g):
        """max-pool backward (+ ReLU mask of the stem output, read off the pooled map: a quarter of the bytes), stem
        wgrad only."""
        if self.has_maxpool:
            g_c1 = self.new(tuple(self.c1.shape))
            self._step(self.bwd_steps, 'mi355_maxpool3x3s2_bwd_pooled', ptr(g), ptr(self.mp_idx), ptr(g_c1),
                       ptr(self.mp), c_int(self.B), c_int(self.c1.shape[1]), c_int(self.c1.shape[2]), c_int(64))
            g = g_c1
        self.conv_bwd(self.stem, self.img_nhwc, g, dgrad=False)

    def _sumpool(self, src, dst):
        N, H, W, C = src.shape
        self._step(self.bwd_steps, 'mi355_sumpool2x2_bf16', ptr(src), ptr(dst), c_int(N), c_int(H),
                   c_int(W), c_int(C), c_int(1))

    def _head_bwd(self, out_spec, tower, acts, gout, kp, g_feat, first):
        B = self.B
        rows = B * self.ncells
        # output conv: dy = gout viewed per level with the image stride of the [B, cells, kp] buffer
        xs = acts[-1]
        wprobs, dprobs = [], []
        gbuf = self.new((rows, 256))
        gys = self._level_views(gbuf)
        for x, gy, off in zip(xs, gys, self.cell_off):
            h, w = x.shape[1], x.shape[2]
            dy_ptr = gout.data_ptr() + self.esz * off * kp
            pr = fwd_problem(x, dy_ptr, 3, 3, 1, 1, K=kp)
            pr.out_sN = self.ncells * kp
            wprobs.append(pr)
            # dgrad: input = dy [B,h,w,kp] with batch stride, output = grad of the tower's last act
            dp = ops_conv.ConvProblem()
            dp.in_, dp.out, dp.mask = dy_ptr, gy.data_ptr(), x.data_ptr()
            dp.N, dp.H, dp.W, dp.C, dp.P, dp.Q, dp.sh, dp.sw = B, h, w, kp, h, w, 1, 1
            dp.TR, dp.TS, dp.dh0, dp.dhs, dp.dw0, dp.dws = 3, 3, 1, -1, 1, -1
            dp.r0, dp.rs, dp.s0, dp.ss, dp.S = 0, 1, 0, 1, 3
            dp.out_mh, dp.out_oh, dp.out_mw, dp.out_ow = 1, 0, 1, 0
            dp.out_sN, dp.out_sH, dp.out_sW = h * w * 256, w * 256, 256
            dp.in_sN = self.ncells * kp
            dprobs.append(dp)
        self._wgrad(wprobs, out_spec, rows)
        self._gather(self.bwd_steps, dprobs, out_spec.w_dgr, 256, 9)
        # tower convs 4..1
        for li in range(3, -1, -1):
            spec = tower[li]
            xs = acts[li - 1] if li > 0 else self.levels
            wprobs = [fwd_problem(x, gy.data_ptr(), 3, 3, 1, 1, K=256) for x, gy in zip(xs, gys)]
            self._wgrad(wprobs, spec, rows)
            if li > 0:
                nbuf = self.new((rows, 256))
                nys = self._level_views(nbuf)
                probs = []
                for x, gy, ny in zip(xs, gys, nys):
                    probs += dgrad_problems(gy, ny.data_ptr(), x.shape[1], x.shape[2], 256, 3, 3, 1, 1, mask=x)
                self._gather(self.bwd_steps, probs, spec.w_dgr, 256, 9)
                gbuf, gys = nbuf, nys
            else:
                probs = []
                for x, gy, gf in zip(xs, gys, g_feat):
                    probs += dgrad_problems(gy, gf.data_ptr(), x.shape[1], x.shape[2], 256, 3, 3, 1, 1,
                                            addend=None if first else gf)
                self._gather(self.bwd_steps, probs, spec.w_dgr, 256, 9)

    def _build_param_grads(self):
        """BN-folded layers (and padded ones): dW = G*scale, dgamma, dbeta from the raw wgrad —
        ONE table-driven launch per step (table built lazily per gradient arena)."""
        self.pgrad_specs = [s for s in self.specs if not s.direct]
        self.pgrad_tables = {}

    def _run_param_grads(self, arena, stream, segment=None):
        """`segment` = (lo, hi) arena range: only the layers whose weights live there (overlapped DP
        exchange); None = every layer."""
        if not self.pgrad_specs:
            return
        key = (arena, segment)
        tab = self.pgrad_tables.get(key)
        if tab is None:
            self._carve_grads(arena)
            seen, descs, nb, chain_tail = {}, [], 0, {}
            for s in self.pgrad_specs:
                seen[id(s.conv)] = seen.get(id(s.conv), 0) + 1
            for s in self.pgrad_specs:
                if segment is not None and not (segment[0] <= self.grad_offsets[id(s.conv.weight)] < segment[1]):
                    continue
                d = self._layer_desc(s, arena)
                d.atomic = 1 if seen[id(s.conv)] > 1 else 0
                d.block_begin = nb
                nb += s.K
                descs.append(d)
                if d.atomic and self.deterministic:
                    # descriptors that write the same dW (a conv shared by the pyramid levels): a chain in table order,
                    # the head adds the members' slabs in that order and stores once (no atomics: bit-reproducible)
                    prev = chain_tail.get(id(s.conv))
                    if prev is None:
                        d.atomic = 0
                    else:
                        descs[prev].share_next = len(descs)          # 1-based index of this descriptor
                        d.atomic = 2
                    chain_tail[id(s.conv)] = len(descs) - 1
            if self._overwrite_grads():
                for d in descs:
                    if d.atomic == 0 and d.dW != self._stem_dwp_ptr():
                        d.atomic = 3                   # sole writer: plain stores (begin_backward zeroes only the gaps)
            tab = (ops_conv.upload_table(descs, self.dev), len(descs), nb) if descs else None
            self.pgrad_tables[key] = tab or ()
        if not tab:
            return
        rc = self.lib.mi355_param_grads_table(ptr(tab[0]), c_int(tab[1]), c_int(tab[2]), c_vp(stream))
        if rc:
            raise _lib.Mi355Error('param_grads_table: %s' % self.lib.mi355_last_error().decode())
        if getattr(self, 'stem_s2d', False):
            off = self.grad_offsets[id(self.stem.conv.weight)]
            if segment is None or segment[0] <= off < segment[1]:
                self._stem_unpack_grad(arena, stream)          # same stream, behind the table launch

    def backward_segments(self, min_bytes=32 << 20):
        """Cut the backward pass for an overlapped data-parallel exchange: [(n_steps, lo, hi)] in execution
        order, meaning "after the first n_steps backward launches (+ the parameter-gradient table of
        arena range [lo, hi)) the gradients in [lo, hi) are final".  Ranges are cut at top-level module
        boundaries (parameters() order is stem, stages, pyramid, heads; backward runs the other way), each
        at least `min_bytes` so that a collective stays xGMI-efficient."""
        cache = self.__dict__.setdefault('_segments', {})
        if min_bytes in cache:
            return cache[min_bytes]
        ready = {}
        for s in self.specs:
            end = getattr(s, '_wgrad_end', len(self.bwd_steps))
            for p in (s.conv.weight, s.conv.bias, getattr(s.bn, 'weight', None), getattr(s.bn, 'bias', None)):
                if p is not None:
                    ready[id(p)] = max(ready.get(id(p), 0), end)
        bounds = []                                   # (offset, ready step) per top-level child, in order
        for _, child in self.model.named_children():
            ps = list(child.parameters())
            if ps:
                bounds.append((self.grad_offsets[id(ps[0])],
                               max(ready.get(id(p), len(self.bwd_steps)) for p in ps)))
        segs, hi, rdy = [], self.grad_numel_padded, 0
        for off, r in reversed(bounds):
            rdy = max(rdy, r)
            if (hi - off) * 4 >= min_bytes or off == bounds[0][0]:
                lo = 0 if off == bounds[0][0] else off
                segs.append([rdy, lo, hi])
                hi, rdy = lo, 0
        for i in range(1, len(segs)):                 # execution order must be monotone
            segs[i][0] = max(segs[i][0], segs[i - 1][0])
        segs[-1][0] = len(self.bwd_steps)
        cache[min_bytes] = [tuple(x) for x in segs]
        return cache[min_bytes]

    # ------------------------------------------------------------------ execution
    def _stream(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    def run_heads(self, img):
        """prep + forward launches; fills self.cls / self.reg."""
        st = self._stream()
        self.forward_serial += 1
        img = img.contiguous()
        if img.dtype != torch.float32:
            img = img.float()
        self._img_ref = img
        if getattr(self, 'stem_s2d', False):
            rc = self.lib.mi355_nchw_to_nhwc_s2d_bf16(ptr(img), ptr(self.img_nhwc), c_int(self.B), c_int(self.H),
                                                      c_int(self.W), c_vp(st))
        else:
            to_nhwc = self.lib.mi355_nchw_to_nhwc_f32 if self.prec == 'fp32' else self.lib.mi355_nchw_to_nhwc_bf16
            rc = to_nhwc(ptr(img), ptr(self.img_nhwc), c_int(self.B), c_int(3),
                         c_int(self.H), c_int(self.W), c_int(8), c_vp(st))
        if rc:
            raise _lib.Mi355Error('nchw_to_nhwc: %s' % self.lib.mi355_last_error().decode())
        self._run(self.prep_steps, st)
        self.join_side()                               # (EfficientDet: half of the depthwise weight prep ran there)
        self._run(self.fwd_steps, st)
        self.join_side()

    def run_loss(self, anchors, annot):
        annot = annot.contiguous().float()
        self._annot = annot
        self._anchors = anchors
        M = annot.shape[1]
        _lib.call('mi355_focal_loss_fwd', ptr(self.cls), ptr(self.reg), ptr(anchors), ptr(annot),
                  c_int(self.B), c_int(self.A), c_int(self.num_classes), c_int(M),
                  c_f32(ops_detect.NEG_THR), c_f32(ops_detect.POS_THR), c_f32(ops_detect.ALPHA),
                  ptr(self.losses), ptr(self.argmax), ptr(self.state), ptr(self.num_pos),
                  ptr(self.loss_ws), c_i64(self.loss_ws_bytes), c_vp(self._stream()))
        return self.losses

    # ------------------------------------------------------------------ a forward overwritten before its backward
    def restore_forward(self, anchors, img, annot, versions):
        """The plan keeps ONE set of activations per input shape; the reference's autograd graph keeps them per call.
        When another forward of the same shape ran between a loss and its backward (a validation / `return_head` call,
        kseven_inference_debuger.py:310-317; a second micro-batch), the backward re-runs the forward of ITS batch first:
        the same launches on the same inputs and weights restore the activations bit for bit.  Not possible - and raised,
        as before - when the forward has side effects or draws: train-mode BatchNorm (running statistics), drop-connect;
        or when a parameter was modified in place since (the reference's autograd raises there too)."""
        why = None
        if self.bn_saved:
            why = 'the plan has train-mode BatchNorm layers (a second forward would move the running statistics again)'
        elif getattr(self, 'drop_connect', False):
            why = 'drop-connect is active (a second forward would draw other branches)'
        elif versions != tuple(p._version for p in self.params):
            why = 'a parameter was modified in place since that forward'
        if why:
            raise RuntimeError('mi355det: another forward ran on this launch plan (shape %dx%dx%d) between this '
                               'loss\'s forward and its backward, and the forward cannot be re-run: %s; call '
                               'backward() before the next forward of the same input shape' % (self.B, self.H, self.W, why))
        with torch.no_grad():
            self.run_heads(img)
            self.run_loss(anchors, annot)
        return self.forward_serial

    def _stem_dwp_ptr(self):
        return self.stem_dwp.data_ptr() if getattr(self, 'stem_s2d', False) else -1

    def _overwrite_grads(self):
        """The parameter-gradient tables STORE a layer's gradients instead of adding to a zeroed arena (0.3 GB less
        traffic per RetinaNet-R50 step: the 145 MB memset and the read half of the read-modify-write).  Only where every
        descriptor is the sole writer of its tensors: the plain RetinaNet plan in slab mode (no shared-conv chains, no
        atomics, no extra tables).  MI355_PG_OVERWRITE=0 disables it."""
        if '_ow' not in self.__dict__:
            seen = {}
            for s in self.pgrad_specs:
                seen[id(s.conv)] = seen.get(id(s.conv), 0) + 1
            self._ow = (type(self) is RetinaPlan and self.deterministic and bool(self.pgrad_specs) and
                        all(v == 1 for v in seen.values()) and os.environ.get('MI355_PG_OVERWRITE', '1') == '1')
        return self._ow

    def _gap_views(self, arena):
        """Overwrite mode: the arena ranges NO table descriptor stores (frozen / unused parameters, alignment, the
        space-to-depth stem's weight whose gradient is added by the unpack launch), merged."""
        self._carve_grads(arena)
        base, cov = self.grad_arenas[arena].data_ptr(), []
        for s in self.pgrad_specs:
            d = self._layer_desc(s, arena)
            if d.dW == self._stem_dwp_ptr():       # (this descriptor keeps adding: all of its targets are gaps)
                continue
            if d.dW:
                cov.append(((d.dW - base) // 4, s.conv.weight.numel()))
            for q in (d.dgamma, d.dbeta, d.dconv_bias):
                if q:
                    cov.append(((q - base) // 4, s.K))
        cov.sort()
        gaps, pos = [], 0
        for lo, n in cov:
            assert 0 <= lo and lo + n <= self.grad_numel and lo >= pos, 'overlapping gradient targets'
            if lo > pos:
                gaps.append((pos, lo))
            pos = lo + n
        if pos < self.grad_arenas[arena].numel():
            gaps.append((pos, self.grad_arenas[arena].numel()))
        return [self.grad_arenas[arena][lo:hi] for lo, hi in gaps]

    def _zero_tensors(self, arena):
        """Everything a backward pass accumulates into, zeroed by ONE launch in front of it (mi355_zero_ranges)."""
        ts = self._gap_views(arena) if self._overwrite_grads() else [self.grad_arenas[arena]]
        if self.scratch.numel() > 1:                   # (atomic mode's raw gradients; the slab mode has none)
            ts.append(self.scratch)
        if getattr(self, 'stem_s2d', False):
            ts.append(self.stem_dwp)
        return ts + self._zero_extra()

    def _zero_for_backward(self, arena, stream):
        cache = self.__dict__.setdefault('_zero_tables', {})
        ent = cache.get(arena)
        if ent is None:
            ts = [t for t in self._zero_tensors(arena) if t.numel()]
            rows = []
            for t in ts:
                assert t.is_contiguous() and (t.numel() * t.element_size()) % 4 == 0 and t.data_ptr() % 4 == 0
                rows += [t.data_ptr(), t.numel() * t.element_size()]
            tab = torch.tensor(rows, dtype=torch.int64).to(self.dev) if rows else None
            big = max([t.numel() * t.element_size() for t in ts] + [0])
            ent = cache[arena] = (tab, len(ts), ts, max(1, min(1024, big >> 16)))
        if ent[1]:
            _lib.call('mi355_zero_ranges', ptr(ent[0]), c_int(ent[1]), c_int(ent[3]), c_vp(stream))

    def begin_backward(self, g_cls, g_reg):
        """Zero the gradient arena, loss backward (d loss / d head outputs).  Returns the arena index."""
        st = self._stream()
        arena = self._arena_for_backward()
        self.last_arena = arena
        gl = self.glosses
        if g_cls is not None and g_reg is not None and g_cls.dtype == torch.float32 and g_reg.dtype == torch.float32 and \
                g_cls.device == self.dev and g_reg.data_ptr() == g_cls.data_ptr() + 4:
            gl = g_cls                                 # (the trainer's pair of ones: adjacent floats, no staging copies)
        else:
            gl[0:1].copy_(g_cls.reshape(1) if g_cls is not None else torch.zeros(1, device=self.dev))
            gl[1:2].copy_(g_reg.reshape(1) if g_reg is not None else torch.zeros(1, device=self.dev))
        self._zero_for_backward(arena, st)
        annot = self._annot
        _lib.call('mi355_focal_loss_bwd', ptr(self.cls), ptr(self.reg), ptr(self._anchors), ptr(annot),
                  ptr(self.argmax), ptr(self.state), ptr(self.num_pos), ptr(gl),
                  c_int(self.B), c_int(self.A), c_int(self.num_classes), c_int(annot.shape[1]),
                  c_f32(ops_detect.ALPHA), c_int(2 if self.prec == 'fp32' else 1), ptr(self.gcls), ptr(self.greg),
                  c_int(self.A_cell), c_int(self.cls_pitch), c_int(self.reg_pitch), c_vp(st))
        self._bwd_pos = 0
        return arena

    def run_backward_segment(self, seg):
        """Backward launches up to seg[0], then the parameter gradients of arena range [seg[1], seg[2]).
        With the side stream the table launch queues there behind this range's wgrads (stream order), i.e.
        beside the data gradients of the layers still to come; `exchange_stream()` is then the stream on
        which the range is final."""
        st = self._stream()
        arena = self.last_arena
        steps = self.bwd_steps if arena == 0 else self.bwd_steps_alt
        self._run(steps, st, self._bwd_pos, seg[0])
        self._bwd_pos = seg[0]
        self._fold_side2()
        self._table_stream = None
        if self._side_active() and self._side_dirty and self._tail_wgrads_main():
            rc = self.lib.mi355_stream_fork(c_vp(st), c_vp(self._side.cuda_stream))   # (a table may read a slab written on main)
            if rc:
                raise _lib.Mi355Error('stream_fork: %s' % self.lib.mi355_last_error().decode())
        if self._side_active() and self._side_dirty:
            # The side stream is the critical path of the backward (tools/side_balance.py: it finishes 0.55 ms behind
            # main at R50 512^2), and a range's table - HBM-bound, 80 - 150 us - only needs that range's weight
            # gradients.  MI355_TABLE_STREAM=1 sends it to the branch stream behind an edge from the side stream, beside
            # the weight gradients of the layers still to come.  Measured and OFF: the side stream finished at the same
            # 6.31 ms (the weight gradients slow down by what the table takes from them) and the step lost 0.7 %: the
            # backward is throughput-bound, only short latency-bound launches gain from another stream.
            if (self.use_branch and self.n_side == 1 and not torch.cuda.is_current_stream_capturing() and
                    os.environ.get('MI355_TABLE_STREAM', '0') == '1'):
                if self._side2 is None:
                    self._side2 = torch.cuda.Stream(device=self.dev)
                rc = self.lib.mi355_stream_fork(c_vp(self._side.cuda_stream), c_vp(self._side2.cuda_stream))
                if rc:
                    raise _lib.Mi355Error('stream_fork: %s' % self.lib.mi355_last_error().decode())
                self._run_param_grads(arena, self._side2.cuda_stream, (seg[1], seg[2]))
                self._pstate.side2_dirty = 1
                self._table_stream = self._side2
            else:
                self._run_param_grads(arena, self._side.cuda_stream, (seg[1], seg[2]))
        else:
            self._run_param_grads(arena, st, (seg[1], seg[2]))

    def exchange_stream(self):
        """Stream on which the gradients of the last finished range are complete."""
        if getattr(self, '_table_stream', None) is not None:
            return self._table_stream
        if self._side_active() and self._side_dirty:
            return self._side
        return torch.cuda.current_stream(self.dev)

    def run_backward(self, g_cls, g_reg, want_views=True):
        arena = self.begin_backward(g_cls, g_reg)
        st = self._stream()
        if type(self)._build_backward is RetinaPlan._build_backward and self.pgrad_specs:
            for seg in self.backward_segments(int(os.environ.get('MI355_SEG_MB', '32')) << 20):   # tables ride the side stream
                self.run_backward_segment(seg)
            self.join_side()
        else:
            self._run(self.bwd_steps if arena == 0 else self.bwd_steps_alt, st)
            self.join_side()
            self._run_param_grads(arena, st)
        self._extra_param_grads(arena, st)
        if not want_views:
            return None
        return [self.grad_view(p, arena) if (p.requires_grad and id(p) not in self.unused_param_ids) else None
                for p in self.params]

    def skip_ranges(self):
        """Sorted, merged arena ranges [lo, hi) of the parameters torch's clip_grad_norm_ / AdamW would skip
        (`grad is None`): requires_grad == False or unused by this plan."""
        out = []
        for p in self.params:
            if not p.requires_grad or id(p) in self.unused_param_ids:
                lo = self.grad_offsets[id(p)]
                if out and out[-1][1] == lo:
                    out[-1][1] = lo + p.numel()
                else:
                    out.append([lo, lo + p.numel()])
        return [tuple(r) for r in out]

    def _zero_extra(self):
        """More tensors the backward accumulates into (subclasses)."""
        return []

    def _extra_param_grads(self, arena, stream):
        pass

    def _arena_for_backward(self):
        lo = self.grad_arenas[0].data_ptr()
        hi = lo + self.grad_numel * 4
        for p in self.params:
            g = p.grad
            if g is not None and lo <= g.data_ptr() < hi:
                if not hasattr(self, 'bwd_steps_alt'):
                    self._build_alt_backward()
                return 1
        return 0

    def _build_alt_backward(self):
        """Second copy of the backward launches targeting gradient arena 1 (used only when live
        `.grad` tensors still alias arena 0, i.e. gradient accumulation without zero_grad)."""
        main_steps, main_keep = self.bwd_steps, self.keep
        self.bwd_steps = []
        self._carve_grads(1)
        self._build_arena = 1                      # (train-mode BN: dgamma / dbeta pointers of the alternate arena)
        self._build_backward()
        self._build_arena = 0
        self.bwd_steps_alt = self.bwd_steps
        self.bwd_steps = main_steps
        self._carve_grads(0)


class RetinaPlanTwoLevel(RetinaPlan):
    """RetinaNetTiny (retinanet.py:447-669: no max-pool, layer1..3, FPN over (C2, C3) -> [P2, P3, P4])
    and RetinaNetP45P6 (retinanet.py:682-918: layer1..4, FPN over (C4, C5) -> [P4, P5, P6 = conv5x5 s3]).
    Same stem/stage/head launches as RetinaPlan; only the pyramid wiring differs."""

    def _meta(self):
        RetinaPlan._meta(self)
        m = self.model
        self.stage_names = tuple(m._stage_names)
        self.fpn_names = tuple(m._fpn_names)       # (top_1, top_2, low_1, low_2, extra)
        self.has_maxpool = hasattr(m, 'maxpool')
        self.input_multiple = m._input_multiple

    def _fpn_fwd(self, feats):
        top1, top2, low1, low2, extra = (self.fpn[n] for n in self.fpn_names)
        self.clow, self.ctop = feats[-2], feats[-1]
        self.toplat = self.conv(top1, self.ctop)
        self.ptop = self.conv(top2, self.toplat)
        self.lowlat = self.conv(low1, self.clow, addend=self.toplat, add_shift=1)
        self.plow = self.conv(low2, self.lowlat)
        self.pextra = self.conv(extra, self.ctop)
        self.levels = [self.plow, self.ptop, self.pextra]

    def _fpn_bwd(self, g_feat):
        top1, top2, low1, low2, extra = (self.fpn[n] for n in self.fpn_names)
        g_low, g_top, g_extra = g_feat
        g_ctop = self.conv_bwd(extra, self.ctop, g_extra)
        g_toplat = self.conv_bwd(top2, self.toplat, g_top)
        g_lowlat = self.conv_bwd(low2, self.lowlat, g_low)
        self._sumpool(g_lowlat, g_toplat)
        g_clow = self.conv_bwd(low1, self.clow, g_lowlat)
        self.conv_bwd(top1, self.ctop, g_toplat, dx=g_ctop, addend=g_ctop, mask=self.ctop)
        return g_ctop, {len(self.stage_names) - 2: g_clow}
