"""Tensor-level wrappers over the conv entry points of the C ABI (include/mi355det_conv.h).

Activations are NHWC bf16 tensors `[N,H,W,C]` (C a multiple of 8); weights are bf16 shadow copies
`w_fwd [K][R*S][C]` / `w_dgr [C][R*S][Kpad]` produced by `prep_weight` from the fp32 masters.
"""
import ctypes

import torch

from ._lib import c_f32, c_i64, c_int, c_vp, call, ptr, require_gpu, stream_of

MAX_PROBLEMS = 5


class ConvProblem(ctypes.Structure):
    _fields_ = [('in_', c_vp), ('out', c_vp), ('addend', c_vp), ('mask', c_vp),
                ('N', c_int), ('H', c_int), ('W', c_int), ('C', c_int),
                ('P', c_int), ('Q', c_int), ('sh', c_int), ('sw', c_int),
                ('TR', c_int), ('TS', c_int), ('dh0', c_int), ('dhs', c_int),
                ('dw0', c_int), ('dws', c_int),
                ('r0', c_int), ('rs', c_int), ('s0', c_int), ('ss', c_int), ('S', c_int),
                ('out_mh', c_int), ('out_oh', c_int), ('out_mw', c_int), ('out_ow', c_int),
                ('out_sN', c_i64), ('out_sH', c_i64), ('out_sW', c_i64),
                ('add_sN', c_i64), ('add_sH', c_i64), ('add_sW', c_i64),
                ('add_shift', c_int), ('reserved', c_int), ('in_sN', c_i64),
                ('weight', c_vp), ('bias', c_vp)]       # optional per-problem weights / bias (bf16 gather launches)


class ConvLaunch(ctypes.Structure):
    _fields_ = [('prob', ConvProblem * MAX_PROBLEMS), ('nprob', c_int), ('K', c_int),
                ('weight', c_vp), ('T', c_int), ('act', c_int), ('bias', c_vp),
                ('out_f32', c_int), ('flags', c_int), ('stats', c_vp)]


class LayerDesc(ctypes.Structure):
    """Mi355LayerDesc (include/mi355det_conv.h): one row of the device-resident layer table."""
    _fields_ = [('w', c_vp), ('sk', c_i64), ('sc', c_i64), ('sr', c_i64), ('ss', c_i64),
                ('K', c_int), ('C', c_int), ('R', c_int), ('S', c_int), ('Cpad', c_int), ('Kpad', c_int),
                ('block_begin', c_int), ('eps', c_f32), ('atomic', c_int), ('nslabs', c_int),
                ('f32_shadows', c_int), ('share_next', c_int),
                ('gamma', c_vp), ('beta', c_vp), ('mean', c_vp), ('var', c_vp), ('conv_bias', c_vp),
                ('scale', c_vp), ('shift', c_vp), ('w_fwd', c_vp), ('w_dgr', c_vp),
                ('G', c_vp), ('colsum', c_vp), ('dW', c_vp), ('dgamma', c_vp), ('dbeta', c_vp),
                ('dconv_bias', c_vp), ('slab_stride', c_i64)]


def upload_table(descs, device):
    """ctypes descriptor list -> device uint8 tensor (plain data, built once per plan)."""
    arr = (LayerDesc * len(descs))(*descs)
    buf = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
    return buf.to(device)


ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2


def _dp(t):
    return 0 if t is None else t.data_ptr()


def out_hw(h, w, k, stride, pad):
    return (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1


def fwd_problem(x, out_ptr, R, S, stride, pad, out_strides=None, K=None, addend=None, add_shift=0,
                mask=None, out_size=None):
    """Forward-conv problem over NHWC `x`; output NHWC [N,P,Q,K] unless `out_strides`
    (sN, sH, sW in elements) says otherwise.  `pad` may be (pad_top, pad_left) with an explicit
    `out_size` (TF-"same" asymmetric padding)."""
    N, H, W, C = x.shape
    if isinstance(pad, tuple):
        pad_t, pad_l = pad
        P, Q = out_size
    else:
        pad_t = pad_l = pad
        P, Q = out_hw(H, W, R, stride, pad)[0], out_hw(H, W, S, stride, pad)[1]
    pr = ConvProblem()
    pr.in_, pr.out, pr.addend, pr.mask = x.data_ptr(), out_ptr, _dp(addend), _dp(mask)
    pr.N, pr.H, pr.W, pr.C, pr.P, pr.Q, pr.sh, pr.sw = N, H, W, C, P, Q, stride, stride
    pr.TR, pr.TS, pr.dh0, pr.dhs, pr.dw0, pr.dws = R, S, -pad_t, 1, -pad_l, 1
    pr.r0, pr.rs, pr.s0, pr.ss, pr.S = 0, 1, 0, 1, S
    pr.out_mh, pr.out_oh, pr.out_mw, pr.out_ow = 1, 0, 1, 0
    if out_strides is None:
        out_strides = (P * Q * K, Q * K, K)
    pr.out_sN, pr.out_sH, pr.out_sW = out_strides
    if addend is not None:
        pr.add_sN, pr.add_sH, pr.add_sW = addend.stride(0), addend.stride(1), addend.stride(2)
    pr.add_shift = add_shift
    return pr


def dgrad_problems(dy, dx_ptr, H, W, Cin, R, S, stride, pad, addend=None, mask=None):
    """Data-gradient problems: one per output-parity class (stride^2 classes; 1 for stride 1).
    dy: NHWC [N,P,Q,Kpad]; dx: NHWC [N,H,W,Cin]."""
    N, P, Q, Kp = dy.shape
    probs = []
    for ph in range(stride):
        for pw in range(stride):
            rf, sf = (ph + pad) % stride, (pw + pad) % stride
            TR = len(range(rf, R, stride))
            TS = len(range(sf, S, stride))
            Pp, Qp = (H - ph + stride - 1) // stride, (W - pw + stride - 1) // stride
            if Pp <= 0 or Qp <= 0:
                continue
            pr = ConvProblem()
            pr.in_, pr.out, pr.addend, pr.mask = dy.data_ptr(), dx_ptr, _dp(addend), _dp(mask)
            pr.N, pr.H, pr.W, pr.C, pr.P, pr.Q, pr.sh, pr.sw = N, P, Q, Kp, Pp, Qp, 1, 1
            pr.TR, pr.TS = TR, TS
            pr.dh0, pr.dhs = (ph + pad - rf) // stride, -1
            pr.dw0, pr.dws = (pw + pad - sf) // stride, -1
            pr.r0, pr.rs, pr.s0, pr.ss, pr.S = rf, stride, sf, stride, S
            pr.out_mh, pr.out_oh, pr.out_mw, pr.out_ow = stride, ph, stride, pw
            pr.out_sN, pr.out_sH, pr.out_sW = H * W * Cin, W * Cin, Cin
            if addend is not None:      # addend indexed on the dx grid (h, w) = (p*stride+ph, ...)
                pr.add_sN = addend.stride(0)
                pr.add_sH, pr.add_sW = addend.stride(1) * stride, addend.stride(2) * stride
                pr.addend = addend.data_ptr() + addend.element_size() * (ph * addend.stride(1) + pw * addend.stride(2))
            probs.append(pr)
    return probs


def launch_gather(problems, weight, K, T, stream_tensor, bias=None, act=ACT_NONE, out_f32=False, ksplit=1, flags=0,
                  stats=None):
    f32 = weight.dtype == torch.float32            # fp32-accurate mode: every operand fp32 (conv_f32.hip)
    for i in range(0, len(problems), MAX_PROBLEMS):
        chunk = problems[i:i + MAX_PROBLEMS]
        L = ConvLaunch()
        for j, pr in enumerate(chunk):
            L.prob[j] = pr
        L.nprob, L.K, L.weight, L.T = len(chunk), K, weight.data_ptr(), T
        L.act, L.bias, L.out_f32, L.flags = act, _dp(bias), 1 if out_f32 else 0, flags
        L.stats = _dp(stats)
        if f32:
            call('mi355_conv_gather_f32', ctypes.byref(L), stream_of(stream_tensor))
        elif ksplit > 1:      # reduction split over blocks, partial tiles in a scratch workspace
            ws = torch.empty(ksplit * sum(p.N * p.P * p.Q for p in chunk) * K, dtype=torch.float32,
                             device=stream_tensor.device)
            call('mi355_conv_gather_splitk', ctypes.byref(L), c_int(ksplit), ptr(ws), stream_of(stream_tensor))
        else:
            call('mi355_conv_gather', ctypes.byref(L), stream_of(stream_tensor))


def launch_wgrad(problems, K, T, dw, split_rows, stream_tensor, colsum=None, f32=False):
    L = ConvLaunch()
    for j, pr in enumerate(problems):
        L.prob[j] = pr
    L.nprob, L.K, L.T = len(problems), K, T
    if f32:                                        # fp32-accurate mode (atomic form: slab stride 0)
        call('mi355_conv_wgrad_slabs_f32', ctypes.byref(L), ptr(dw), c_i64(0), c_int(0), ptr(colsum),
             c_int(split_rows), stream_of(stream_tensor))
        return
    call('mi355_conv_wgrad_colsum', ctypes.byref(L), ptr(dw), ptr(colsum), c_int(split_rows),
         stream_of(stream_tensor))


def wgrad_row3_ok(problems, cpad, kpad=128):
    """Launch takes the row-of-taps wgrad kernel (mirrors the host check in conv_wgrad.hip)."""
    import os
    if os.environ.get('MI355_WG_ROW3', '1') == '0' or cpad % 64:
        return False
    dyn_on = os.environ.get('MI355_WG_ROW3_DYN', '1') != '0'

    def width_ok(w):
        pow2 = (w & (w - 1)) == 0
        if w <= 64 and pow2:
            return True
        # widths that do not divide the 64-pixel stage: the table-driven edge form, where it beats the generic kernel
        return dyn_on and (pow2 or (kpad >= 128 and cpad <= 256))
    return all(p.TR == 3 and p.TS == 3 and p.sh == 1 and p.sw == 1 and p.dhs == 1 and p.dws == 1 and p.dw0 == -1
               and p.Q == p.W and width_ok(p.W) for p in problems)


def wgrad_row3x_ok(problems, cpad, kpad):
    """Launch takes the eight-wave ring form of the row-of-taps kernel (conv_wgrad_row3x_kernel: one block per CU;
    mirrors the host check in conv_wgrad.hip)."""
    import os
    if os.environ.get('MI355_WG_ROW3X', '1') == '0' or cpad % 128 or kpad < 128:
        return False
    return wgrad_row3_ok(problems, cpad, kpad) and all(p.W <= 64 and (p.W & (p.W - 1)) == 0 for p in problems)


def wgrad_target_blocks(kpad, cpad, taps, problems=None):
    """Blocks a weight-gradient launch should be cut into: two four-wave blocks per CU, or one eight-wave block per CU
    for the ring form (MI355_WG_BLOCKS / MI355_WG_X_BLOCKS)."""
    import os
    if problems is not None and taps == 9 and wgrad_row3x_ok(problems, cpad, kpad):
        return int(os.environ.get('MI355_WG_X_BLOCKS', '256'))
    return int(os.environ.get('MI355_WG_BLOCKS', '512'))


def wgrad_tiles(kpad, cpad, taps, problems=None):
    """Output tiles of one pixel chunk of the wgrad launch (mirrors conv_wgrad.hip)."""
    kt = -(-kpad // 128)
    if problems is not None and taps == 9 and wgrad_row3x_ok(problems, cpad, kpad):
        return kt * 3 * (cpad // 128)
    if problems is not None and taps == 9 and wgrad_row3_ok(problems, cpad, kpad):
        return kt * 3 * (cpad // 64)
    return kt * (-(-taps * cpad // 128)) if cpad < 128 else kt * taps * (-(-cpad // 128))


def fit_split_rows(problem_rows, tiles_per_chunk, max_blocks, rows, granule=64):
    """The smallest split >= `rows` (a multiple of `granule`) with sum_i ceil(M_i / split) * tiles <= max_blocks.
    A grouped launch cuts every problem on its own, so the chunk count is sum_i ceil(M_i / split), not
    ceil(sum_i M_i / split): the five pyramid levels of a head tower at 512^2 gave 23 chunks x 12 tiles = 276 blocks for
    a target of 256 - with ONE eight-wave block per CU (the ring weight-gradient kernel) the last 20 blocks ran as a
    second round behind the 65-stage chunks of the first."""
    rows = max(granule, -(-rows // granule) * granule)
    top = max(problem_rows)
    while rows < top and sum(-(-m // rows) for m in problem_rows) * tiles_per_chunk > max_blocks:
        rows += granule
    return rows


def pick_split_rows(total_rows, tiles_per_chunk, target_blocks=512, min_rows=512, granule=64):
    """Rows of the pixel reduction per block: ~2 four-wave blocks per CU, but never so few rows that
    the partial-tile epilogue (64 KiB per block) outweighs the MFMA work.  (Measured on R50 512^2
    bs16: 512 blocks beat 256..1024; with the slabs summed by the table kernel a 1024-row floor
    (the engine's default) beats 512..2048 by 1-2 %; one eight-wave block per CU with two row
    groups, MI355_WG_GROUPS=2 + granule 128, halves the slabs but runs 2 % slower end to end.)"""
    chunks = max(1, target_blocks // max(1, tiles_per_chunk))
    rows = -(-total_rows // chunks)
    import os
    relax_blocks = int(os.environ.get('MI355_WG_RELAX_BLOCKS', '128'))
    if tiles_per_chunk * -(-total_rows // min_rows) < relax_blocks:
        # the floor would leave most CUs without a block (EfficientDet's 1x1 convs: one 64x64 tile, 64 chunks of
        # 1024 rows on 256 CUs): relax it to 256 rows / one block per TWO CUs, whichever comes first (one block per CU
        # wrote twice the slabs for launches that are latency-bound either way: D0 2 208 -> 2 222 img/s; no relaxing at
        # all: 2 177)
        min_rows = max(256, min(min_rows, -(-total_rows * tiles_per_chunk // max(relax_blocks, 1))))
        min_rows = -(-min_rows // granule) * granule
    rows = max(min_rows, -(-rows // granule) * granule)
    return rows


# ---------------------------------------------------------------- simple single-problem helpers
def conv_fwd(x, w_fwd, K, R, S, stride, pad, bias=None, act=ACT_NONE, addend=None, add_shift=0,
             out_f32=False, ksplit=1, stats=None):
    require_gpu(x, w_fwd)
    N, H, W, C = x.shape
    P, Q = out_hw(H, W, R, stride, pad)[0], out_hw(H, W, S, stride, pad)[1]
    out = torch.empty((N, P, Q, K), dtype=torch.float32 if (out_f32 or x.dtype == torch.float32) else torch.bfloat16,
                      device=x.device)
    pr = fwd_problem(x, out.data_ptr(), R, S, stride, pad, K=K, addend=addend, add_shift=add_shift)
    launch_gather([pr], w_fwd, K, R * S, x, bias=bias, act=act, out_f32=out_f32, ksplit=ksplit, stats=stats)
    return out


def conv_dgrad(dy, w_dgr, H, W, Cin, R, S, stride, pad, addend=None, mask=None):
    require_gpu(dy, w_dgr)
    N = dy.shape[0]
    dx = torch.empty((N, H, W, Cin), dtype=dy.dtype, device=dy.device)
    probs = dgrad_problems(dy, dx.data_ptr(), H, W, Cin, R, S, stride, pad, addend=addend, mask=mask)
    launch_gather(probs, w_dgr, Cin, R * S, dy)
    return dx


def conv_wgrad(x, dy, R, S, stride, pad, split_rows=None, out=None, colsum=None):
    """-> G fp32 [Kpad][R*S][C] = sum_m dy[m][k] * x[m @ tap][c] (accumulates into `out`)."""
    require_gpu(x, dy)
    N, H, W, C = x.shape
    Kp = dy.shape[3]
    T = R * S
    if out is None:
        out = torch.zeros((Kp, T, C), dtype=torch.float32, device=x.device)
    pr = fwd_problem(x, dy.data_ptr(), R, S, stride, pad, K=Kp)
    M = N * pr.P * pr.Q
    if split_rows is None:
        split_rows = pick_split_rows(M, wgrad_tiles(Kp, C, T))
    launch_wgrad([pr], Kp, T, out, split_rows, x, colsum=colsum, f32=x.dtype == torch.float32)
    return out


# ---------------------------------------------------------------- aux kernels
def nchw_to_nhwc(img, cpad):
    require_gpu(img)
    img = img.contiguous().float()
    N, C, H, W = img.shape
    out = torch.empty((N, H, W, cpad), dtype=torch.bfloat16, device=img.device)
    call('mi355_nchw_to_nhwc_bf16', ptr(img), ptr(out), c_int(N), c_int(C), c_int(H), c_int(W),
         c_int(cpad), stream_of(img))
    return out


def prep_weight(w, cpad, scale, w_fwd, w_dgr):
    """w: fp32 [K,C,R,S] (any strides) -> bf16 shadows (either may be None)."""
    K, C, R, S = w.shape
    sk, sc, sr, ss = w.stride()
    call('mi355_prep_weight', ptr(w), c_i64(sk), c_i64(sc), c_i64(sr), c_i64(ss), c_int(K), c_int(C),
         c_int(R), c_int(S), c_int(cpad), ptr(scale), ptr(w_fwd), ptr(w_dgr), stream_of(w))


def bn_fold(bn, scale, shift, conv_bias=None):
    call('mi355_bn_fold', ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var),
         ptr(conv_bias), c_f32(bn.eps), c_int(bn.num_features), ptr(scale), ptr(shift), stream_of(scale))


def colsum(g, out):
    M = g.numel() // g.shape[-1]
    call('mi355_colsum_bf16', ptr(g), c_i64(M), c_int(g.shape[-1]), ptr(out), stream_of(g))


def conv_param_grads(G, w, cpad, scale, colsum_, bn, dW, dgamma, dbeta, accumulate=False):
    K, C, R, S = w.shape
    sk, sc, sr, ss = w.stride()
    assert dW.stride() == w.stride()
    call('mi355_conv_param_grads', ptr(G), ptr(w), c_i64(sk), c_i64(sc), c_i64(sr), c_i64(ss),
         c_int(K), c_int(C), c_int(R), c_int(S), c_int(cpad), ptr(scale), ptr(colsum_),
         ptr(bn.running_mean if bn is not None else None),
         ptr(bn.running_var if bn is not None else None), c_f32(bn.eps if bn is not None else 0.0),
         ptr(dW), ptr(dgamma), ptr(dbeta), c_int(1 if accumulate else 0), c_vp(0), c_vp(0), stream_of(G))


def maxpool_fwd(x):
    N, H, W, C = x.shape
    P, Q = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((N, P, Q, C), dtype=torch.bfloat16, device=x.device)
    idx = torch.empty((N, P, Q, C), dtype=torch.int8, device=x.device)
    call('mi355_maxpool3x3s2_fwd', ptr(x), ptr(out), ptr(idx), c_int(N), c_int(H), c_int(W), c_int(C),
         stream_of(x))
    return out, idx


def maxpool_bwd(gout, idx, in_shape, relu_mask=None, pooled=None):
    """relu_mask: the pool's input (a ReLU output) - or `pooled`, the pool's output, which carries the same mask."""
    N, H, W, C = in_shape
    gin = torch.empty(in_shape, dtype=torch.bfloat16, device=gout.device)
    if pooled is not None:
        call('mi355_maxpool3x3s2_bwd_pooled', ptr(gout), ptr(idx), ptr(gin), ptr(pooled), c_int(N), c_int(H),
             c_int(W), c_int(C), stream_of(gout))
    else:
        call('mi355_maxpool3x3s2_bwd', ptr(gout), ptr(idx), ptr(gin), ptr(relu_mask), c_int(N), c_int(H),
             c_int(W), c_int(C), stream_of(gout))
    return gin


def sumpool2x2(src, dst, accumulate):
    N, H, W, C = src.shape
    call('mi355_sumpool2x2_bf16', ptr(src), ptr(dst), c_int(N), c_int(H), c_int(W), c_int(C),
         c_int(1 if accumulate else 0), stream_of(src))


def relu_mask(x, ref, out=None):
    out = torch.empty_like(x) if out is None else out
    call('mi355_relu_mask_bf16', ptr(x), ptr(ref), ptr(out), c_i64(x.numel()), stream_of(x))
    return out
