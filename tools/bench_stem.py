"""Stem as run today (7x7 stride 2 on [B,512,512,8]) vs its space-to-depth form (4x4 stride 1 on [B,256,256,16]): forward and
weight-gradient launch times.  Usage: python tools/bench_stem.py [B S]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from objectdetection.pytorch_amd import ops_conv
B, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 512)
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(0)
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
K = 64
for name, (C, R, stride, pad, H) in (('7x7 s2 C8', (8, 7, 2, 3, S)), ('4x4 s1 C16 (space-to-depth)', (16, 4, 1, (2, 2), S // 2))):
    x = torch.randn((B, H, H, C), generator=g).to(dev).to(torch.bfloat16)
    w_fwd = (torch.randn((K, R * R, C), generator=g) * 0.05).to(dev).to(torch.bfloat16)
    P = S // 2
    if isinstance(pad, tuple):
        out = torch.empty((B, P, P, K), dtype=torch.bfloat16, device=dev)
        pr = ops_conv.fwd_problem(x, out.data_ptr(), R, R, stride, pad, K=K, out_size=(P, P))
        f = lambda: ops_conv.launch_gather([pr], w_fwd, K, R * R, x, act=ops_conv.ACT_RELU)
    else:
        f = lambda: ops_conv.conv_fwd(x, w_fwd, K, R, R, stride, pad, act=ops_conv.ACT_RELU)
    dy = torch.randn((B, P, P, K), generator=g).to(dev).to(torch.bfloat16)
    G = torch.zeros((K, R * R, C), dtype=torch.float32, device=dev)
    prw = ops_conv.fwd_problem(x, dy.data_ptr(), R, R, stride, pad, K=K, **({'out_size': (P, P)} if isinstance(pad, tuple) else {}))
    split = ops_conv.pick_split_rows(B * P * P, ops_conv.wgrad_tiles(K, C, R * R), min_rows=1024)
    fw = lambda: ops_conv.launch_wgrad([prw], K, R * R, G, split, x)
    print('%-30s fwd %.1f us   wgrad %.1f us (split %d)' % (name, timed(f), timed(fw), split))
