"""Input side of the training path on the device.

Mirrors the names of the reference's `datasettool/dataloader.py`: `Normalizer` (:905-924), the zero
canvas of `Resizer` resize mode 1 (:872-880, :886) and `collater` (:655-691) — but as ONE launch of
`mi355_collate_images` plus one of `mi355_collate_annots` over the whole batch (include/mi355det_input.h),
fed with the decoded uint8 images: the dataset workers ship 3 bytes per pixel and the fp32 NCHW batch
the models take is produced in HBM.  The image resampling (skimage.transform.resize, :866) is not part of
this path: samples arrive at their final size.  There is no CPU fallback.
"""
import ctypes

import numpy as np
import torch

from .. import _lib
from .._lib import c_int, c_vp, ptr

IN_U8, IN_SCALE255, IN_NORMALIZE = 1, 2, 4


class Normalizer(object):
    """The reference's constants (dataloader.py:908-910, float64).  On this path the arithmetic runs inside
    the collate launch; the object only carries mean / std."""

    def __init__(self, inference_mode=False, mean=None, std=None, **kwargs):
        self.mean = np.array([[[0.485, 0.456, 0.406]]]) if mean is None else np.asarray(mean, np.float64).reshape(1, 1, 3)
        self.std = np.array([[[0.229, 0.224, 0.225]]]) if std is None else np.asarray(std, np.float64).reshape(1, 1, 3)
        self.inference_mode = inference_mode


class DeviceCollater(object):
    """collater(data) for samples that are still raw: `data` is a list of dicts with
         'img'   uint8 [h,w,3] (decoded image; scaled by 1/255 and normalised on the device) or
                 float32 [h,w,3] (already in [0,1] when `normalizer` is set, else taken as is),
         'annot' float64/float32 [m,5] rows (x1,y1,x2,y2,label), may be empty,
         'scale' the resize factor already applied to the image (optional; multiplies annot[:, :4], :886).
    Returns {'img': [B,3,H,W] fp32, 'annot': [B,M,5] fp32 padded with -1 ([B,1,5] of -1 when no sample has
    a box), 'scale': [...]} on `device`.  H, W = the batch maximum, or the `Resizer` canvas when
    height/width are given (rounded up to base_size, dataloader.py:872-876)."""

    def __init__(self, normalizer=None, height=None, width=None, base_size=32, device=None, scale_annots=False):
        self.normalizer = normalizer
        self.height, self.width, self.base_size = height, width, base_size
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.scale_annots = scale_annots
        self._staging = [None, None]          # pinned host buffers, alternated: the H2D copy is asynchronous
        self._events = [None, None]
        self._turn = 0

    def _canvas(self, imgs):
        H = max(int(i.shape[0]) for i in imgs)
        W = max(int(i.shape[1]) for i in imgs)
        if self.height is not None:
            bs = self.base_size
            ch, cw = self.height + (bs - self.height % bs) % bs, self.width + (bs - self.width % bs) % bs
            if H > ch or W > cw:
                raise ValueError('sample of %dx%d does not fit the %dx%d canvas' % (H, W, ch, cw))
            H, W = ch, cw
        return H, W

    def _stage(self, nbytes):
        k = self._turn
        self._turn ^= 1
        if self._events[k] is not None:
            self._events[k].synchronize()      # the copy that last used this buffer has landed
        if self._staging[k] is None or self._staging[k].numel() < nbytes:
            self._staging[k] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8).pin_memory()
        return k, self._staging[k]

    def __call__(self, data):
        imgs = [np.ascontiguousarray(s['img'].numpy() if torch.is_tensor(s['img']) else s['img']) for s in data]
        annots = [np.asarray(s['annot'].numpy() if torch.is_tensor(s['annot']) else s['annot'], dtype=np.float64).reshape(-1, 5)
                  for s in data]
        scales = [s.get('scale', 1.0) for s in data]
        B = len(imgs)
        if B == 0:
            raise ValueError('empty batch')
        kinds = {i.dtype for i in imgs}
        if kinds not in ({np.dtype(np.uint8)}, {np.dtype(np.float32)}):
            raise TypeError('images must be all uint8 or all float32 HWC, got %s' % sorted(map(str, kinds)))
        for i in imgs:
            if i.ndim != 3 or i.shape[2] != 3:
                raise ValueError('image of shape %s, expected [h,w,3]' % (i.shape,))
        u8 = imgs[0].dtype == np.uint8
        flags = (IN_U8 | IN_SCALE255 if u8 else 0) | (IN_NORMALIZE if self.normalizer is not None else 0)
        H, W = self._canvas(imgs)
        # ---- one pinned staging buffer: [image bytes ... | annotation rows | scales | image meta | annot meta]
        offs, pos = [], 0
        for i in imgs:
            offs.append(pos)
            pos += (i.nbytes + 15) // 16 * 16
        img_bytes = pos
        rows = sum(a.shape[0] for a in annots)
        ann_off = pos
        pos += rows * 40
        scale_off = pos
        pos += B * 8
        meta_off = pos
        pos += B * 5 * 8
        k, host = self._stage(pos)
        hb = host.numpy()
        for i, o in zip(imgs, offs):
            hb[o:o + i.nbytes] = i.reshape(-1).view(np.uint8)
        if rows:
            hb[ann_off:ann_off + rows * 40] = np.concatenate(annots, axis=0).reshape(-1).view(np.uint8)
        hb[scale_off:scale_off + B * 8] = np.asarray(scales, dtype=np.float64).view(np.uint8)
        meta = np.zeros((B * 5,), dtype=np.int64)
        im_meta, an_meta = meta[:B * 3].reshape(B, 3), meta[B * 3:].reshape(B, 2)
        first = 0
        for b, (i, a) in enumerate(zip(imgs, annots)):
            im_meta[b] = (offs[b], i.shape[0], i.shape[1])
            an_meta[b] = (first, a.shape[0])
            first += a.shape[0]
        hb[meta_off:meta_off + B * 40] = meta.view(np.uint8)
        with torch.cuda.device(self.device):
            dev = torch.empty(pos, dtype=torch.uint8, device=self.device)
            dev.copy_(host[:pos], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._events[k] = ev
            st = c_vp(torch.cuda.current_stream().cuda_stream)
            base = dev.data_ptr()
            out = torch.empty((B, 3, H, W), dtype=torch.float32, device=self.device)
            mean = std = None
            if self.normalizer is not None:
                mean = (ctypes.c_double * 3)(*np.asarray(self.normalizer.mean, np.float64).reshape(3))
                std = (ctypes.c_double * 3)(*np.asarray(self.normalizer.std, np.float64).reshape(3))
            _lib.call('mi355_collate_images', c_vp(base), c_vp(base + meta_off), c_int(B), c_int(H), c_int(W),
                      c_int(flags), mean, std, ptr(out), st)
            M = max(1, max(a.shape[0] for a in annots))
            ann = torch.empty((B, M, 5), dtype=torch.float32, device=self.device)
            _lib.call('mi355_collate_annots', c_vp(base + ann_off), c_vp(base + meta_off + B * 24),
                      c_vp(base + scale_off) if self.scale_annots else c_vp(0), c_int(B), c_int(M), ptr(ann), st)
        return {'img': out, 'annot': ann, 'scale': scales}


def collater(data, device=None):
    """Drop-in for dataloader.py:655-691: samples already normalised on the host (float32 HWC) are padded and
    permuted on the device; raw uint8 samples go through `DeviceCollater(Normalizer())` instead."""
    return DeviceCollater(normalizer=None, device=device)(data)
