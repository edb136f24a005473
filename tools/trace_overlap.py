#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --output-format csv  ->  how many kernels are in flight, and which ones run alone.

Usage: python tools/trace_overlap.py <dir with *_kernel_trace.csv> [first_fraction last_fraction]
Looks at the window [first, last] of the traced wall time (default 0.5 .. 0.95: steady-state steps) and prints
the share of that window with 0 / 1 / 2 / >= 3 kernels in flight, plus the kernels that account for the time
spent alone (one kernel in flight) — the launches a second stream could still overlap."""
import csv
import glob
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    f0, f1 = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.5, 0.95)
    paths = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
    assert paths, 'no *kernel_trace.csv under ' + d
    rows = list(csv.DictReader(open(paths[0])))
    ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows]
    t_lo = min(e[0] for e in ev)
    t_hi = max(e[1] for e in ev)
    w0, w1 = t_lo + f0 * (t_hi - t_lo), t_lo + f1 * (t_hi - t_lo)
    pts = []
    for s, e, n in ev:
        if e <= w0 or s >= w1:
            continue
        pts.append((max(s, w0), 1, n))
        pts.append((min(e, w1), -1, n))
    pts.sort(key=lambda p: (p[0], p[1]))
    live = defaultdict(int)
    depth = 0
    hist = defaultdict(float)
    alone = defaultdict(float)
    last = w0
    for t, dlt, n in pts:
        dt = t - last
        if dt > 0:
            hist[min(depth, 3)] += dt
            if depth == 1:
                alone[next(k for k, v in live.items() if v > 0)] += dt
        last = t
        depth += dlt
        live[n] += dlt
    hist[min(depth, 3)] += w1 - last
    tot = w1 - w0
    print('window %.2f ms of %.2f ms traced, %d kernels' % (tot / 1e6, (t_hi - t_lo) / 1e6, len(pts) // 2))
    for k in range(4):
        print('  %s kernels in flight: %5.1f %%' % ('>=3' if k == 3 else ' %d ' % k, 100 * hist[k] / tot))
    print('time alone, by kernel (%% of window):')
    for n, v in sorted(alone.items(), key=lambda kv: -kv[1])[:14]:
        print('  %5.1f %%  %s' % (100 * v / tot, n[:110]))


if __name__ == '__main__':
    main()
