"""Micro-benchmark of one conv shape through the C ABI: fwd gather / dgrad / wgrad.
Usage: python tools/bench_conv.py N H W C K R stride [iters] [what=fwd|dgrad|wgrad]"""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from objectdetection.pytorch_amd import ops_conv

N, H, W, C, K, R, stride = [int(v) for v in sys.argv[1:8]]
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 20
what = sys.argv[9] if len(sys.argv) > 9 else 'fwd'
pad = R // 2
dev = torch.device('cuda', 0)
torch.manual_seed(0)
x = torch.randn(N, H, W, C, device=dev).to(torch.bfloat16)
w = (torch.randn(K, R * R, C, device=dev) * 0.05).to(torch.bfloat16)
wd = (torch.randn(C, R * R, K, device=dev) * 0.05).to(torch.bfloat16)
b = torch.randn(K, device=dev)
P, Q = ops_conv.out_hw(H, W, R, stride, pad)
dy = torch.randn(N, P, Q, K, device=dev).to(torch.bfloat16)
msk = torch.randn(N, H, W, C, device=dev).to(torch.bfloat16)
G = torch.zeros(K, R * R, C, device=dev)


def run():
    if what == 'fwd':
        return ops_conv.conv_fwd(x, w, K, R, R, stride, pad, bias=b, act=ops_conv.ACT_RELU)
    if what == 'dgrad':
        return ops_conv.conv_dgrad(dy, wd, H, W, C, R, R, stride, pad, mask=msk)
    return ops_conv.conv_wgrad(x, dy, R, R, stride, pad, out=G)


run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / iters * 1e3
fl = 2.0 * N * P * Q * K * C * R * R
by = (x.numel() + N * P * Q * K) * 2
print('%s N%d %dx%d C%d K%d R%d s%d: %.1f us  %.1f TF/s  %.2f TB/s(min traffic)' % (what, N, H, W, C, K, R, stride, us, fl / us / 1e6, by / us / 1e6))
