#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats --output-format csv  ->  a readable per-kernel table (profiles/*.txt).
Usage: python tools/summarize_stats.py <dir with *_kernel_stats.csv> "<header line(s)>" > profiles/rNN_....txt"""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    for h in sys.argv[2:]:
        print('# ' + h)
    paths = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)
    assert paths, 'no *kernel_stats.csv under ' + d
    rows = list(csv.DictReader(open(paths[0])))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print('%-88s %8s %12s %10s %10s %6s' % ('kernel', 'calls', 'total_ms', 'avg_us', 'max_us', 'pct'))
    for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
        print('%-88s %8d %12.3f %10.1f %10.1f %6.2f' % (r['Name'][:88], int(r['Calls']), float(r['TotalDurationNs']) / 1e6,
                                                         float(r['AverageNs']) / 1e3, float(r['MaxNs']) / 1e3,
                                                         100 * float(r['TotalDurationNs']) / tot))


if __name__ == '__main__':
    main()
