"""Times the device-side input launch (mi355_collate_images) on 16 x 512 x 512 decoded images resident in HBM
and the whole DeviceCollater call from host arrays (staging copy + H2D + launches).  python tools/bench_input.py"""
import os, sys, time, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from objectdetection.pytorch_amd import _lib
from objectdetection.pytorch_amd._lib import c_int, c_vp, ptr
from objectdetection.pytorch_amd.datasettool.dataloader import DeviceCollater, Normalizer

B, H, W = 16, 512, 512
dev = torch.device('cuda', 0)
rng = np.random.default_rng(0)
imgs = [rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8) for _ in range(B)]
packed = torch.from_numpy(np.stack(imgs).reshape(-1)).to(dev)
meta = torch.tensor([[b * H * W * 3, H, W] for b in range(B)], dtype=torch.int64, device=dev)
out = torch.empty((B, 3, H, W), device=dev)
mean = (ctypes.c_double * 3)(0.485, 0.456, 0.406); std = (ctypes.c_double * 3)(0.229, 0.224, 0.225)
st = torch.cuda.current_stream()
def launch():
    _lib.call('mi355_collate_images', ptr(packed), ptr(meta), c_int(B), c_int(H), c_int(W), c_int(7), mean, std, ptr(out), c_vp(st.cuda_stream))
for _ in range(5): launch()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); [launch() for _ in range(50)]; e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
nbytes = B * H * W * (3 + 12)
print('collate_images u8 16x512x512: %.1f us/launch, %.2f MB algorithmic -> %.2f TB/s' % (us, nbytes / 1e6, nbytes / us / 1e6))
col = DeviceCollater(Normalizer())
data = [{'img': i, 'annot': np.zeros((8, 5))} for i in imgs]
for _ in range(3): col(data)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): col(data)
torch.cuda.synchronize(); print('DeviceCollater from host arrays: %.2f ms/batch (staging memcpy + 12.6 MB H2D + launches)' % ((time.perf_counter() - t0) * 100))
x = torch.from_numpy(np.stack(imgs)).float()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    y = ((x / 255.0 - torch.tensor([0.485, 0.456, 0.406])) / torch.tensor([0.229, 0.224, 0.225])).permute(0, 3, 1, 2).contiguous().to(dev)
torch.cuda.synchronize(); print('host normalise + permute + 50 MB fp32 H2D (the reference shape of the work): %.2f ms/batch' % ((time.perf_counter() - t0) * 200))
