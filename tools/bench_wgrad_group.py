"""Would grouping the same-shape weight gradients of a ResNet stage into one launch pay?  Times the wgrad of one
block (batch 16) against ONE launch over n blocks' worth of rows (batch 16 n), slab mode as the plans run it."""
import sys, os, ctypes
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from objectdetection.pytorch_amd import ops_conv, _lib
from objectdetection.pytorch_amd._lib import c_i64, c_int, c_vp, ptr

dev = torch.device('cuda', 0)


def run(N, H, W, C, K, R, stride, n):
    pad = R // 2
    T = R * R
    res = []
    for mult in (1, n):
        B = N * mult
        x = torch.randn(B, H, W, C, device=dev).to(torch.bfloat16)
        P = (H + 2 * pad - R) // stride + 1
        dy = torch.randn(B, P, P, K, device=dev).to(torch.bfloat16)
        pr = ops_conv.fwd_problem(x, dy.data_ptr(), R, R, stride, pad, K=K)
        M = B * P * P
        tiles = ops_conv.wgrad_tiles(K, C, T, [pr])
        split = ops_conv.pick_split_rows(M, tiles, 512, 1024, 64)
        nslabs = -(-M // split)
        L = ops_conv.ConvLaunch()
        L.prob[0] = pr
        L.nprob, L.K, L.T = 1, K, T
        ws = torch.empty(nslabs, K * T * C, device=dev)
        cs = torch.zeros(nslabs, K, device=dev)
        st = c_vp(torch.cuda.current_stream().cuda_stream)
        args = (ctypes.byref(L), ptr(ws), c_i64(K * T * C), c_int(nslabs), ptr(cs), c_int(split), st)
        _lib.call('mi355_conv_wgrad_slabs', *args)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            _lib.call('mi355_conv_wgrad_slabs', *args)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    fl = 2.0 * N * ((H + 2 * pad - R) // stride + 1) ** 2 * K * C * T
    print('C%-4d K%-4d %dx%d s%d %3dx%-3d  one block %6.1f us (%5.0f TF/s) | %d blocks in one launch %6.1f us = %6.1f us per block (%5.0f TF/s)'
          % (C, K, R, R, stride, H, W, res[0], fl / res[0] / 1e6, n, res[1], res[1] / n, fl * n / res[1] / 1e6))


for case in ((16, 32, 32, 256, 256, 3, 1, 6), (16, 32, 32, 1024, 256, 1, 1, 6), (16, 32, 32, 256, 1024, 1, 1, 6),
             (16, 64, 64, 128, 128, 3, 1, 4), (16, 64, 64, 512, 128, 1, 1, 4), (16, 64, 64, 128, 512, 1, 1, 4),
             (16, 16, 16, 512, 512, 3, 1, 3), (16, 16, 16, 2048, 512, 1, 1, 3), (16, 16, 16, 512, 2048, 1, 1, 3),
             (16, 128, 128, 64, 64, 3, 1, 3), (16, 128, 128, 256, 64, 1, 1, 3), (16, 128, 128, 64, 256, 1, 1, 3)):
    run(*case)
