"""ctypes loader for libmi355det.so (the C ABI in include/mi355det.h).

Fails loudly: a missing library or a missing symbol is an ImportError-like RuntimeError the
moment an op is used; nothing silently falls back to PyTorch/CPU.
"""
import ctypes
import os
import re
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libmi355det.so')
HEADER_DIR = os.path.join(os.path.dirname(os.path.dirname(_HERE)), 'include')

_lock = threading.Lock()
_lib = None


class Mi355Error(RuntimeError):
    pass


def declared_symbols():
    """Every `mi355_*` function declared in include/*.h (used by the symbol-export test)."""
    names = []
    for fn in sorted(os.listdir(HEADER_DIR)):
        if fn.endswith('.h'):
            text = open(os.path.join(HEADER_DIR, fn)).read()
            names += re.findall(r'\b(mi355_[a-z0-9_]+)\s*\(', text)
    return sorted(set(names))


def lib():
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise Mi355Error(
                        'mi355det: %s not built. Run `python -c "import __graft_entry__ as g; '
                        'g.build()"` or `make -C objectdetection/pytorch_amd/csrc`. '
                        'There is no CPU fallback.' % LIB_PATH)
                # torch first: it bundles its own libamdhip64.so, and the library resolves its HIP calls against whichever
                # runtime the process loaded FIRST.  Loaded before torch (build() then smoke() in one process), it bound
                # the system runtime of /opt/rocm beside torch's: two HIP runtimes in one process, and the launches of this
                # library failed with "no ROCm-capable device is detected" on a box where torch saw the GPU.
                import torch  # noqa: F401
                handle = ctypes.CDLL(LIB_PATH)
                handle.mi355_last_error.restype = ctypes.c_char_p
                _lib = handle
    return _lib


# MI355_TRACE=<file>: fault attribution (tools/find_fault.py).  Every launch is announced in the file (flushed to disk)
# BEFORE it is enqueued and the device is synchronised behind it, so that when the runtime aborts the process on a GPU
# memory fault the last line names the launch that faulted.  Plans replay through the Python loop in this mode.
TRACE_PATH = os.environ.get('MI355_TRACE') or None
_trace_fh = None


def trace_begin(tag):
    global _trace_fh
    if _trace_fh is None:
        _trace_fh = open(TRACE_PATH, 'a')
    _trace_fh.write(tag + '\n')
    _trace_fh.flush()
    os.fsync(_trace_fh.fileno())


def trace_end():
    import torch
    torch.cuda.synchronize()


def call(name, *args):
    """Call `name(*args)`; raise Mi355Error with the library's message on a non-zero code."""
    fn = getattr(lib(), name)
    if TRACE_PATH:
        trace_begin('call %s' % name)
    rc = fn(*args)
    if TRACE_PATH and rc == 0:
        trace_end()
    if rc != 0:
        msg = lib().mi355_last_error()
        raise Mi355Error('%s failed (%d): %s' % (name, rc, msg.decode() if msg else '?'))


c_int = ctypes.c_int
c_i64 = ctypes.c_int64
c_f32 = ctypes.c_float
c_f64 = ctypes.c_double
c_vp = ctypes.c_void_p


def ptr(t):
    """Borrowed device pointer of a tensor (None -> NULL)."""
    return c_vp(0 if t is None else t.data_ptr())


def stream_of(t):
    import torch
    return c_vp(torch.cuda.current_stream(t.device).cuda_stream)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise Mi355Error('mi355det: tensor on %s; the HIP path needs a ROCm GPU tensor '
                             '(no CPU fallback in the product path)' % t.device)
