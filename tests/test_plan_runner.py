"""The launch-plan runner below Python (include/mi355det_plan.h, csrc/plan_run.hip): table generation, argument
marshalling and the run / stop-at-host-step / resume / fail protocol.  No GPU: the only entry point run here is
mi355_zero_ranges with n == 0 (returns before any launch) and with a bad argument (rejected before any launch)."""
import ctypes
import os
import struct
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plan_table_is_generated_from_the_headers():
    rc = subprocess.call([sys.executable, os.path.join(ROOT, 'tools', 'gen_plan_table.py'), '--check'])
    assert rc == 0, 'csrc/plan_table.inc is stale: run tools/gen_plan_table.py'


def test_every_stream_entry_point_has_a_thunk():
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import gen_plan_table
    from objectdetection.pytorch_amd import _lib
    lib = _lib.lib()
    names = gen_plan_table.stream_entry_points()
    assert len(names) >= 100 and 'mi355_conv_gather' in names and 'mi355_stream_fork' not in names
    for i, n in enumerate(names):
        assert lib.mi355_plan_lookup(n.encode()) == i
        assert 0 <= lib.mi355_plan_arity(i) <= 24
    assert lib.mi355_plan_lookup(b'mi355_no_such_entry') == -1
    assert lib.mi355_plan_arity(len(names)) == -1
    # arity = parameters in front of the stream (include/mi355det_conv.h: mi355_conv_gather(launch, stream))
    assert lib.mi355_plan_arity(lib.mi355_plan_lookup(b'mi355_conv_gather')) == 1
    assert lib.mi355_plan_arity(lib.mi355_plan_lookup(b'mi355_focal_loss_bwd')) == 19


def test_slot_marshalling():
    from objectdetection.pytorch_amd import engine
    s = engine._slot
    assert s(None) == 0 and s(ctypes.c_void_p(None)) == 0 and s(ctypes.c_void_p(0x7f0012345678)) == 0x7f0012345678
    assert s(ctypes.c_int(-3)) & 0xFFFFFFFF == 0xFFFFFFFD
    assert s(ctypes.c_int64(-3)) == (1 << 64) - 3
    assert s(ctypes.c_float(0.35)) == struct.unpack('<I', struct.pack('<f', 0.35))[0]
    assert s(ctypes.c_double(0.35)) == struct.unpack('<Q', struct.pack('<d', 0.35))[0]
    st = engine.PlanState()
    assert s(ctypes.byref(st)) == ctypes.addressof(st)
    arr = (ctypes.c_int * 4)()
    assert s(arr) == ctypes.addressof(arr)
    with pytest.raises(TypeError):
        s('text')


def _table(lib, rows):
    from objectdetection.pytorch_amd import engine
    arr = (engine.PlanStep * len(rows))()
    zr = lib.mi355_plan_lookup(b'mi355_zero_ranges')
    for st, (kind, n) in zip(arr, rows):
        if kind == 'host':
            st.host, st.fn = 1, -1
        elif kind in ('join', 'join2'):                     # edges inside a list: no launch, fn is ignored
            st.fn, st.where = -1, (engine.PLAN_JOIN if kind == 'join' else engine.PLAN_JOIN2)
        else:
            st.fn, st.nargs = zr, 3
            st.where = {'side': engine.PLAN_SIDE, 'side2': engine.PLAN_SIDE2}.get(kind, engine.PLAN_MAIN)
            st.arg[0], st.arg[1], st.arg[2] = 0, n & ((1 << 64) - 1), 1
    return arr


def test_run_protocol_without_a_gpu():
    from objectdetection.pytorch_amd import _lib, engine
    lib = _lib.lib()
    state = engine.PlanState()
    run = lib.mi355_plan_run
    vp = ctypes.c_void_p
    tab = _table(lib, [('main', 0), ('side', 0), ('host', 0), ('main', 0), ('main', -1), ('main', 0)])

    def go(b, e):
        return run(tab, ctypes.c_int(b), ctypes.c_int(e), vp(None), vp(None), vp(None), ctypes.byref(state))
    assert go(0, 2) == 0                       # side steps run on main when the run has no side stream
    assert state.side_dirty == 0 and state.forks == 0
    assert go(0, 6) == 1 and state.next == 2   # stops in front of the host step
    assert go(3, 4) == 0                       # the caller resumes behind it
    rc = go(3, 6)                              # n = -1 is rejected by the entry point: the failing index is reported
    assert rc == -1 and state.failed == 4
    assert b'zero_ranges' in lib.mi355_last_error()
    bad = _table(lib, [('main', 0)])
    bad[0].nargs = 2                           # arity mismatch is caught before the call
    assert run(bad, ctypes.c_int(0), ctypes.c_int(1), vp(None), vp(None), vp(None), ctypes.byref(state)) == -1
    assert state.failed == 0
    # join edges and branch-stream steps degrade to the main stream when the run has no side streams
    tab2 = _table(lib, [('main', 0), ('side2', 0), ('join2', 0), ('side', 0), ('join', 0), ('main', 0)])
    assert run(tab2, ctypes.c_int(0), ctypes.c_int(6), vp(None), vp(None), vp(None), ctypes.byref(state)) == 0
    assert state.side_dirty == 0 and state.side2_dirty == 0 and state.forks == 0
    # join with nothing dirty is a no-op without a GPU
    assert lib.mi355_plan_join(vp(None), vp(None), vp(None), ctypes.byref(state), ctypes.c_int(0)) == 0
