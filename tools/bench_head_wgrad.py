"""The head-tower weight gradient of RetinaNet-R50 512^2 bs16 (five pyramid levels, C = K = 256, 3x3: the largest single
launch class of the step) timed alone, slab mode as the plans run it.  MI355_WG_ROW3X=0/1 selects the four-wave /
eight-wave ring kernel, MI355_X3_ABLATE the diagnostic builds of the ring kernel (conv_wgrad.hip).
NO_COLSUM=1: without the column sums (bias gradient).
Usage: python tools/bench_head_wgrad.py [iters] [K]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from objectdetection.pytorch_amd import ops_conv, _lib                   # noqa: E402
from objectdetection.pytorch_amd._lib import c_i64, c_int, c_vp, ptr     # noqa: E402

dev = torch.device('cuda', 0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
K = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B, C = 16, 256
sizes = [64, 32, 16, 8, 4]
torch.manual_seed(0)
xs = [torch.randn(B, s, s, C, device=dev).to(torch.bfloat16) for s in sizes]
dys = [torch.randn(B, s, s, K, device=dev).to(torch.bfloat16) for s in sizes]
probs = [ops_conv.fwd_problem(x, dy.data_ptr(), 3, 3, 1, 1, K=K) for x, dy in zip(xs, dys)]
rows = sum(B * s * s for s in sizes)
tiles = ops_conv.wgrad_tiles(K, C, 9, probs)
split = ops_conv.pick_split_rows(rows, tiles, target_blocks=ops_conv.wgrad_target_blocks(K, C, 9, probs), min_rows=1024, granule=64)
if os.environ.get('MI355_WG_X_FIT', '0') == '1' and ops_conv.wgrad_row3x_ok(probs, C, K):          # as engine._fit_ring_split
    split = ops_conv.fit_split_rows([B * s * s for s in sizes], tiles, int(os.environ.get('MI355_WG_X_BLOCKS', '256')), split)
nslabs = sum(-(-(B * s * s) // split) for s in sizes)
L = ops_conv.ConvLaunch()
for j, pr in enumerate(probs):
    L.prob[j] = pr
L.nprob, L.K, L.T = len(probs), K, 9
slab = K * 9 * C
ws = torch.empty(nslabs, slab, device=dev)
cs = torch.zeros(nslabs, K, device=dev)
st = c_vp(torch.cuda.current_stream().cuda_stream)
args = (ctypes.byref(L), ptr(ws), c_i64(slab), c_int(nslabs), None if os.environ.get('NO_COLSUM') else ptr(cs), c_int(split), st)
for _ in range(3):
    _lib.call('mi355_conv_wgrad_slabs', *args)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    _lib.call('mi355_conv_wgrad_slabs', *args)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / iters * 1e3
fl = 2.0 * rows * K * C * 9
print('head wgrad C%d K%d: ROW3X=%s ABLATE=%s tiles/chunk %d split %d slabs %d: %.1f us = %.0f TFLOP/s'
      % (C, K, os.environ.get('MI355_WG_ROW3X', '1'), os.environ.get('MI355_X3_ABLATE', '0'), tiles, split, nslabs, us, fl / us / 1e6))
