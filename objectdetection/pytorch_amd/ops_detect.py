"""Tensor-level wrappers over the detection entry points of the C ABI (include/mi355det.h)."""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import c_f32, c_i64, c_int, c_vp, call, ptr, require_gpu, stream_of


def _c(t, dtype=torch.float32):
    """Contiguous, 16-B aligned tensor of `dtype` (copies only when needed)."""
    if t.dtype != dtype:
        t = t.to(dtype)
    t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


# ------------------------------------------------------------------ anchors (anchors.py:38-60)
def anchors_generate(height, width, strides, sizes, ratios, scales, device):
    strides = [int(s) for s in strides]
    nlev = len(strides)
    ratios = np.ascontiguousarray(np.asarray(ratios, dtype=np.float64))
    scales = np.ascontiguousarray(np.asarray(scales, dtype=np.float64))
    per_cell = len(ratios) * len(scales)
    base = np.zeros((nlev, per_cell, 4), dtype=np.float64)
    for l in range(nlev):
        call('mi355_anchors_base', base[l].ctypes.data_as(c_vp), ctypes.c_double(float(sizes[l])),
             ratios.ctypes.data_as(c_vp), c_int(len(ratios)), scales.ctypes.data_as(c_vp),
             c_int(len(scales)))
    offs = (ctypes.c_int64 * (nlev + 1))()
    cstr = (ctypes.c_int * nlev)(*strides)
    call('mi355_anchor_count', c_int(height), c_int(width), c_int(nlev), cstr, c_int(per_cell), offs)
    total = int(offs[nlev])
    out = torch.empty((1, total, 4), dtype=torch.float32, device=device)
    require_gpu(out)
    if total:
        call('mi355_anchors_generate', ptr(out), c_int(height), c_int(width), c_int(nlev), cstr,
             base.ctypes.data_as(c_vp), c_int(per_cell), stream_of(out))
    return out


# ------------------------------------------------------------------ focal loss (losses.py:30-168)
NEG_THR, POS_THR, ALPHA = 0.35, 0.45, 0.25     # losses.py:79, 80, 31 (gamma = 2 is in the kernel)


def focal_loss_fwd(cls, reg, anchors, annot):
    """-> (losses [2], argmax [B,A] int32, state [B,A] int8, num_pos [B] int32)."""
    require_gpu(cls, reg, anchors, annot)
    cls, reg, annot = _c(cls), _c(reg), _c(annot)
    anchors = _c(anchors).reshape(-1, 4)
    B, A, C = cls.shape
    M = annot.shape[1]
    assert reg.shape == (B, A, 4) and anchors.shape[0] == A and annot.shape == (B, M, 5)
    dev = cls.device
    nbytes = ctypes.c_int64()
    call('mi355_focal_loss_workspace_bytes', c_int(B), c_int(A), c_int(C), c_int(M),
         ctypes.byref(nbytes))
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    losses = torch.empty(2, dtype=torch.float32, device=dev)
    argmax = torch.empty((B, A), dtype=torch.int32, device=dev)
    state = torch.empty((B, A), dtype=torch.int8, device=dev)
    num_pos = torch.empty(B, dtype=torch.int32, device=dev)
    call('mi355_focal_loss_fwd', ptr(cls), ptr(reg), ptr(anchors), ptr(annot), c_int(B), c_int(A),
         c_int(C), c_int(M), c_f32(NEG_THR), c_f32(POS_THR), c_f32(ALPHA), ptr(losses), ptr(argmax),
         ptr(state), ptr(num_pos), ptr(ws), c_i64(nbytes.value), stream_of(cls))
    return losses, argmax, state, num_pos, (cls, reg, anchors, annot)


def focal_loss_bwd(saved, argmax, state, num_pos, grad_losses, mode=0):
    cls, reg, anchors, annot = saved
    B, A, C = cls.shape
    M = annot.shape[1]
    dt = torch.float32 if mode == 0 else torch.bfloat16
    gcls = torch.empty((B, A, C), dtype=dt, device=cls.device)
    greg = torch.empty((B, A, 4), dtype=dt, device=cls.device)
    grad_losses = _c(grad_losses)
    call('mi355_focal_loss_bwd', ptr(cls), ptr(reg), ptr(anchors), ptr(annot), ptr(argmax),
         ptr(state), ptr(num_pos), ptr(grad_losses), c_int(B), c_int(A), c_int(C), c_int(M),
         c_f32(ALPHA), c_int(mode), ptr(gcls), ptr(greg), c_int(0), c_int(0), c_int(0), stream_of(cls))
    return gcls, greg


class FocalLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cls, reg, anchors, annot):
        losses, argmax, state, num_pos, saved = focal_loss_fwd(cls, reg, anchors, annot)
        ctx.saved = saved
        ctx.aux = (argmax, state, num_pos)
        ctx.mark_non_differentiable(argmax, state, num_pos)
        return losses[0:1], losses[1:2], argmax, state, num_pos

    @staticmethod
    def backward(ctx, g_cls, g_reg, *_):
        argmax, state, num_pos = ctx.aux
        z = torch.zeros(1, dtype=torch.float32, device=argmax.device)
        g = torch.cat([g_cls.reshape(1) if g_cls is not None else z,
                       g_reg.reshape(1) if g_reg is not None else z]).float()
        gcls, greg = focal_loss_bwd(ctx.saved, argmax, state, num_pos, g, mode=0)
        return gcls, greg, None, None


# ------------------------------------------------------------------ decode / clip (utils.py)
def bbox_decode(boxes, deltas, mean, std, clip_hw=None):
    require_gpu(boxes, deltas)
    boxes, deltas = _c(boxes), _c(deltas)
    B, A = deltas.shape[0], deltas.shape[1]
    batched = 1 if boxes.shape[0] == B and B > 1 else 0
    assert boxes.shape[-2] == A
    out = torch.empty_like(deltas)
    m = (ctypes.c_float * 4)(*[float(x) for x in mean])
    s = (ctypes.c_float * 4)(*[float(x) for x in std])
    h, w = clip_hw if clip_hw is not None else (0, 0)
    call('mi355_bbox_decode', ptr(boxes), c_int(batched), ptr(deltas), c_int(B), c_int(A), m, s,
         c_int(0 if clip_hw is None else 1), c_int(int(h)), c_int(int(w)), ptr(out), stream_of(out))
    return out


def clip_boxes_(boxes, height, width):
    require_gpu(boxes)
    if not (boxes.is_contiguous() and boxes.dtype == torch.float32 and boxes.data_ptr() % 16 == 0):
        tmp = _c(boxes).clone()
        clip_boxes_(tmp, height, width)
        boxes.copy_(tmp)
        return boxes
    call('mi355_clip_boxes', ptr(boxes), c_int(boxes.shape[0]), c_int(boxes.shape[1]),
         c_int(int(height)), c_int(int(width)), stream_of(boxes))
    return boxes


# ------------------------------------------------------------------ filter + NMS (retinanet.py:300-318)
def class_max(cls2d):
    require_gpu(cls2d)
    cls2d = _c(cls2d)
    A, C = cls2d.shape
    score = torch.empty(A, dtype=torch.float32, device=cls2d.device)
    label = torch.empty(A, dtype=torch.int32, device=cls2d.device)
    call('mi355_class_max', ptr(cls2d), c_int(A), c_int(C), ptr(score), ptr(label), stream_of(cls2d))
    return score, label


def score_filter(score, thr):
    require_gpu(score)
    score = _c(score)
    A = score.shape[0]
    idx = torch.empty(A, dtype=torch.int32, device=score.device)
    cnt = torch.empty(1, dtype=torch.int32, device=score.device)
    call('mi355_score_filter', ptr(score), c_int(A), c_f32(thr), ptr(idx), ptr(cnt), stream_of(score))
    return idx[:int(cnt.item())]          # host sync, as the reference's boolean indexing


import os as _os
NMS_HEAD = int(_os.environ.get('MI355_NMS_HEAD', '4096'))               # candidates of a round's matrix pass (two-phase schedule)
NMS_DIRECT_MAX = int(_os.environ.get('MI355_NMS_DIRECT_MAX', '8192'))   # up to here one matrix pass over all candidates


def _aligned(nbytes, device):
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
    return ws, ws.data_ptr() + (-ws.data_ptr()) % 256


def nms(boxes, scores, iou_threshold):
    """Same contract as torchvision.ops.nms: int64 indices, descending score.

    Up to NMS_DIRECT_MAX candidates: one launch sequence (mi355_nms).  More: greedy NMS only compares a box with KEPT
    boxes of higher score, so the suppression bit matrix is built for a HEAD of the sorted candidates only, the TAIL
    is tested against the boxes the head kept (tail x kept pairs), and the compacted survivors go round again —
    46 706 candidates with 4 052 kept: ~0.1e9 IoU tests instead of 1.09e9.  One host read of the survivor count per
    round (this function ends in one anyway: the reference's boolean indexing)."""
    require_gpu(boxes, scores)
    boxes, scores = _c(boxes), _c(scores)
    K = boxes.shape[0]
    dev = boxes.device
    st = stream_of(boxes)
    keep = torch.empty(max(K, 1), dtype=torch.int32, device=dev)
    cnt = torch.zeros(2, dtype=torch.int32, device=dev)          # [kept, survivors of the current round]
    if K <= NMS_DIRECT_MAX:
        nbytes = ctypes.c_int64()
        call('mi355_nms_workspace_bytes', c_int(K), ctypes.byref(nbytes))
        ws, wp = _aligned(nbytes.value, dev)
        call('mi355_nms', ptr(boxes), ptr(scores), c_int(K), c_f32(iou_threshold), ptr(keep), ptr(cnt),
             c_vp(wp), c_i64(nbytes.value), st)
        return keep[:int(cnt[0].item())].long()
    k2 = 1 << max(K - 1, 2047).bit_length()
    sws, swp = _aligned(8 * k2, dev)
    cur_b = torch.empty((K, 4), dtype=torch.float32, device=dev)
    cur_i = torch.empty(K, dtype=torch.int32, device=dev)
    call('mi355_nms_sort', ptr(boxes), ptr(scores), c_int(K), ptr(cur_b), ptr(cur_i), c_vp(swp), c_i64(8 * k2), st)
    nxt_b, nxt_i = torch.empty_like(cur_b), torch.empty_like(cur_i)
    pbytes = ctypes.c_int64()
    call('mi355_nms_presorted_workspace_bytes', c_int(NMS_DIRECT_MAX), ctypes.byref(pbytes))
    pws, pwp = _aligned(pbytes.value, dev)
    fws, fwp = _aligned(12 * ((K + 63) // 64) + 64, dev)
    rem, kept_before, first = K, 0, True
    while rem > 0:
        head = rem if rem <= NMS_DIRECT_MAX else NMS_HEAD
        call('mi355_nms_presorted', ptr(cur_b), ptr(cur_i), c_int(head), c_f32(iou_threshold), ptr(keep), ptr(cnt),
             c_int(0 if first else 1), c_vp(pwp), c_i64(pbytes.value), st)
        first = False
        tail = rem - head
        if tail == 0:
            break
        call('mi355_nms_filter_tail', ptr(boxes), ptr(keep), ptr(cnt), c_int(kept_before),
             c_vp(cur_b.data_ptr() + 16 * head), c_vp(cur_i.data_ptr() + 4 * head), c_int(tail), c_f32(iou_threshold),
             ptr(nxt_b), ptr(nxt_i), c_vp(cnt.data_ptr() + 4), c_vp(fwp), c_i64(12 * ((K + 63) // 64) + 64), st)
        kept_before, rem = (int(v) for v in cnt.tolist())        # the round's host read
        cur_b, nxt_b, cur_i, nxt_i = nxt_b, cur_b, nxt_i, cur_i
    return keep[:int(cnt[0].item())].long()
