"""rocprofv3 SQ-counter pass -> per-kernel wave-cycle split (profiles/<round>_pmc_sq_waves.txt).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY \\
              SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc_sq -o run -- \\
              python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-infer
    python tools/collect_sq.py gpurun_out/pmc_sq > profiles/r03_pmc_sq_waves.txt

Fractions of SQ_WAVE_CYCLES (MI355X_MICROARCH.md, 'rocprofv3 PMC slots': WAIT_ANY = parked on s_waitcnt / barrier,
WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing; disjoint); mfma = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES (a relative
scale between kernels: the gfx950 counter is not normalised per SIMD); lds_conf = SQ_LDS_BANK_CONFLICT / SQ_BUSY_CYCLES."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for path in glob.glob(d + '/*counter_collection.csv'):
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name']
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        key = (r.get('Dispatch_Id'), k)
        if key not in seen:
            seen.add(key)
            cnt[k] += 1
rows = []
for k, c in acc.items():
    wc = c.get('SQ_WAVE_CYCLES', 0.0)
    busy = c.get('SQ_BUSY_CYCLES', 0.0)
    if wc <= 0:
        continue
    rows.append((wc, k, cnt[k], c.get('SQ_WAIT_ANY', 0) / wc, c.get('SQ_WAIT_INST_ANY', 0) / wc, c.get('SQ_WAIT_INST_LDS', 0) / wc,
                 c.get('SQ_ACTIVE_INST_ANY', 0) / wc, c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(busy, 1), c.get('SQ_LDS_BANK_CONFLICT', 0) / max(busy, 1)))
rows.sort(reverse=True)
print('%-78s %6s %9s %9s %9s %9s %8s %8s' % ('kernel', 'launch', 'wait_any', 'wait_inst', 'wait_lds', 'active', 'mfma', 'lds_conf'))
for wc, k, n, a, b, c_, e, m, l in rows[:40]:
    name = k.replace('(anonymous namespace)::', '').replace('void ', '')
    print('%-78s %6d %9.2f %9.2f %9.2f %9.2f %8.2f %8.3f' % (name[:78], n, a, b, c_, e, m, l))
