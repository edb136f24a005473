"""rocprofv3 PMC passes -> profiles/<round>_pmc_traffic.json (HBM-side bytes of the conv kernels per step).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o run -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --graph 0
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o run -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --graph 0
    python tools/collect_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write 5 profiles/r03_pmc_traffic_<tag>.json [the bench command]

(<tag> = bench.py's traffic_tag: arch[_rDEPTH]_SIZE_bBATCH_PRECISION[_syncbn]; steps = warm-up + timed steps of the
profiled command: every step of the run is counted.)

(separate passes: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2.)  Corrections of
MI355X_MICROARCH.md "HBM": both counters are in KiB; on gfx950 FETCH_SIZE reports HALF the bytes of wide
(16 B/lane) coalesced reads, which is what every conv kernel issues, so the read side is doubled
(checked on adamw_flat_kernel: 36.2 M params x 16 B read = 579 MB = 2 x FETCH_SIZE); WRITE_SIZE is exact."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from csrc_hash import csrc_hash   # noqa: E402


def agg(d):
    out = collections.defaultdict(lambda: [0, 0.0])
    for path in glob.glob(d + '/*counter_collection.csv'):
        for r in csv.DictReader(open(path)):
            k = r['Kernel_Name']
            out[k][0] += 1
            out[k][1] += float(r['Counter_Value'])
    return out


def main():
    fdir, wdir, steps, dst = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    f, w = agg(fdir), agg(wdir)
    # the number of training steps the profiled command really ran (bench.py adds host-cost probes to its warm-up and
    # timed steps): one adamw_prepare launch per optimizer step
    marks = [v[0] for k, v in f.items() if k.startswith('adamw_prepare_kernel')]
    if marks and marks[0] != steps:
        print('collect_traffic: %d optimizer steps in the trace (argument said %d)' % (marks[0], steps))
        steps = marks[0]
    kern = {}
    for k in sorted(f, key=lambda k: -f[k][1]):
        rd = 2.0 * f[k][1] * 1024 / steps
        wr = w.get(k, [0, 0.0])[1] * 1024 / steps
        kern[k[:96]] = {'launches_per_step': round(f[k][0] / steps, 2), 'read_bytes_per_step': round(rd),
                        'write_bytes_per_step': round(wr)}
    conv = {k: v for k, v in kern.items() if 'conv_gather' in k or 'conv_wgrad' in k or 'splitk_finish' in k or 'conv_pw_stream' in k}
    tot = sum(v['read_bytes_per_step'] + v['write_bytes_per_step'] for v in conv.values())
    n = sum(v['launches_per_step'] for v in conv.values())
    ours = {k: v for k, v in kern.items() if not (k.startswith('void at::') or k.startswith('__amd_rocclr'))}
    allb = sum(v['read_bytes_per_step'] + v['write_bytes_per_step'] for v in ours.values())
    res = {'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), %d steps' % steps,
           'command': ' '.join(sys.argv[5:]) or None, 'csrc_sha16': csrc_hash(),
           'all_kernels_bytes_per_step': round(allb),
           'all_launches_per_step': round(sum(v['launches_per_step'] for v in ours.values()), 2),
           'corrections': 'KiB -> bytes; FETCH_SIZE x 2 (gfx950 wide coalesced reads); WRITE_SIZE exact',
           'conv_kernels_bytes_per_step': round(tot), 'conv_launches_per_step': n,
           'conv_bytes_per_launch': round(tot / max(n, 1)), 'kernels': kern}
    json.dump(res, open(dst, 'w'), indent=1)
    print(json.dumps({k: res[k] for k in ('conv_kernels_bytes_per_step', 'conv_launches_per_step', 'conv_bytes_per_launch',
                                          'all_kernels_bytes_per_step', 'all_launches_per_step', 'csrc_sha16')}))


if __name__ == '__main__':
    main()
