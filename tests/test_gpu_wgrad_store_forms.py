"""GPU: the weight-gradient slabs do not depend on HOW the epilogue stores them.

The kernels of csrc/conv_wgrad.hip write a block's partial tile with 16-byte non-temporal stores (WLaunch.wide; 4-byte
stores for an unaligned workspace or MI355_WG_WIDE=0, plain instead of non-temporal with MI355_WG_NT=0).  The switches are
read once per process, so each form runs in a process of its own on the same seeded operands (one after the other) and
prints a digest of every slab byte: generic kernel (1x1, flat C < 128 form, ragged K), row-of-taps kernel (a width that
does not divide the stage, K <= 64 form) and the eight-wave ring kernel (grouped pyramid levels)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes, hashlib, sys
import torch
sys.path.insert(0, %r)
from objectdetection.pytorch_amd import ops_conv
from objectdetection.pytorch_amd._lib import c_i64, c_int, call, ptr, stream_of
dev = torch.device('cuda', 0)
def mk(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g).to(dev).to(torch.bfloat16)
h = hashlib.sha256()
#        levels (N, H, W)                      C    K    R  split
CASES = [([(2, 16, 16)],                       256, 512, 1, 128),      # generic kernel, 1x1
         ([(2, 16, 16)],                       64,  136, 3, 192),      # flat form (C < 128), ragged K
         ([(2, 12, 20)],                       128, 256, 3, 128),      # row-of-taps, width 20 (table-driven edges)
         ([(2, 16, 16)],                       64,  64,  3, 128),      # row-of-taps, K <= 64 form
         ([(2, 32, 32), (2, 16, 16), (2, 8, 8), (2, 4, 4)], 256, 256, 3, 512)]    # ring kernel, grouped levels
for ci, (levels, C, K, R, split) in enumerate(CASES):
    keep, probs, rows = [], [], []
    for li, (n, hh, ww) in enumerate(levels):
        x, dy = mk((n, hh, ww, C), 100 * ci + li), mk((n, hh, ww, K), 100 * ci + 50 + li)
        keep += [x, dy]
        probs.append(ops_conv.fwd_problem(x, dy.data_ptr(), R, R, 1, R // 2, K=K))
        rows.append(n * hh * ww)
    L = ops_conv.ConvLaunch()
    for j, pr in enumerate(probs):
        L.prob[j] = pr
    L.nprob, L.K, L.T = len(probs), K, R * R
    nsl = sum(-(-r // split) for r in rows)
    slab = K * R * R * C
    ws = torch.full((nsl, slab), float('nan'), device=dev)
    cs = torch.zeros((nsl, K), device=dev)
    call('mi355_conv_wgrad_slabs', ctypes.byref(L), ptr(ws), c_i64(slab), c_int(nsl), ptr(cs), c_int(split), stream_of(keep[0]))
    torch.cuda.synchronize()
    assert not torch.isnan(ws).any(), 'case %%d: a slab element was not written' %% ci
    h.update(ws.cpu().numpy().tobytes())
    h.update(cs.cpu().numpy().tobytes())
print('DIGEST', h.hexdigest())
''' % ROOT


def _digest(env):
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, '-c', CHILD], env=e, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return [l for l in out.stdout.splitlines() if l.startswith('DIGEST')][-1]


def test_slab_bytes_do_not_depend_on_the_store_form():
    ref = _digest({'MI355_WG_WIDE': '0'})
    assert _digest({'MI355_WG_WIDE': '1', 'MI355_WG_NT': '0'}) == ref
    assert _digest({'MI355_WG_WIDE': '1', 'MI355_WG_NT': '1'}) == ref
