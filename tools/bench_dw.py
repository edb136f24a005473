"""One depthwise-conv entry point on one shape, N launches (for rocprofv3 --pmc / --kernel-trace).
Usage: python tools/bench_dw.py wgrad|fwd|dgrad N H W C k stride in_swish [iters]"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from objectdetection.pytorch_amd import ops_effdet as E
op, N, H, W, C, k, s, sw = sys.argv[1], *[int(v) for v in sys.argv[2:9]]
iters = int(sys.argv[9]) if len(sys.argv) > 9 else 20
dev = torch.device('cuda', 0)
g = torch.Generator(device='cpu').manual_seed(0)
x = torch.randn((N, H, W, C), generator=g).to(dev).to(torch.bfloat16)
pt, P = E.same_pad(H, k, s); pl, Q = E.same_pad(W, k, s)
dy = torch.randn((N, P, Q, C), generator=g).to(dev).to(torch.bfloat16)
w_t = (torch.randn((k * k, C), generator=g) * 0.3).to(dev)
bias = torch.zeros(C, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def run():
    if op == 'wgrad': E.dwconv_wgrad_slabs(x, sw, dy, k, s)
    elif op == 'fwd': E.dwconv_fwd(x, sw, w_t, bias, k, s)
    else: E.dwconv_dgrad(dy, w_t, tuple(x.shape), k, s, zin=x if sw else None)
run(); torch.cuda.synchronize()
e0.record()
for _ in range(iters): run()
e1.record(); torch.cuda.synchronize()
nbytes = (x.numel() + dy.numel()) * 2 + (x.numel() * 2 if op == 'dgrad' and sw else 0)
us = e0.elapsed_time(e1) * 1e3 / iters
print('%s N%d %dx%d C%d k%d s%d sw%d: %.1f us/launch, %.0f GB/s' % (op, N, H, W, C, k, s, sw, us, nbytes / us / 1e3))
