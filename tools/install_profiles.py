"""gpurun_out/prof_* (written by tools/collect_profiles.sh on the GPU box) -> profiles/<round>_*, with headers that say
what each file is and which bench line of the same call it belongs to.  Usage: python tools/install_profiles.py r02"""
import json, shutil, sys
O, P = 'gpurun_out/', 'profiles/'
R = sys.argv[1] if len(sys.argv) > 1 else 'r02'


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


b = {k: last_json(O + 'prof_bench_%s.json' % k) for k in ('n1', 'r50_640', 'effdet_d0', 'effdet_d2')}
for k in b:
    open(P + '%s_bench_%s.json' % (R, k), 'w').write(json.dumps(b[k]) + '\n')
n1, r = b['n1'], b['n1']['roofline']
body = open(O + 'prof_rocprofv3_kernel_stats.txt').read().splitlines()
hdr = ['# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-infer   (final code of the round)',
       '# RetinaNet-R50 512x512 bs16: 25 training steps + the conv-only replays of the roofline probe; two HIP streams: kernel times overlap',
       '# (a kernel that shares the chip with the other stream runs longer than alone: tools/probe_pool.hip measures the max-pool backward at 42 us alone, 140 us here)',
       '# bench line of the un-profiled run in the same call (profiles/%s_bench_n1.json): value %.1f img/s, %.3f ms/step (median %.3f), conv %.3f ms/step = %.1f TFLOP/s (frac %.4f)'
       % (R, n1['value'], n1['ms_per_step'], n1['ms_per_step_median'], r['conv_ms_per_step'], r['achieved'], r['frac'])]
open(P + '%s_rocprofv3_kernel_stats.txt' % R, 'w').write('\n'.join(hdr + body[1:]) + '\n')
shutil.copy(O + 'prof_rocprofv3_kernel_stats.csv', P + '%s_rocprofv3_kernel_stats.csv' % R)
d0, r0 = b['effdet_d0'], b['effdet_d0']['roofline']
body = open(O + 'prof_effdet_d0_kernel_stats.txt').read().splitlines()
hdr = ['# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --arch efficientdet-d0 --steps 20 --warmup 5 --no-cpu-baseline --no-infer   (final code of the round)',
       '# EfficientDet-D0 512x512 bs16 (BASELINE configs[3]), drop-connect on; 25 steps on two HIP streams (kernel times overlap); 494 plan launches per step',
       '# bench line of the un-profiled run in the same call (profiles/%s_bench_effdet_d0.json): value %.1f img/s, %.3f ms/step; roofline hbm: %.1f GB/s algorithmic (frac %.4f)'
       % (R, d0['value'], d0['ms_per_step'], r0['achieved'], r0['frac'])]
open(P + '%s_effdet_d0_kernel_stats.txt' % R, 'w').write('\n'.join(hdr + body[1:]) + '\n')
open(P + '%s_conv_per_launch.txt' % R, 'w').write(open(O + 'prof_conv_per_launch.txt').read())
body = [l for l in open(O + 'prof_effdet_d0_per_launch.txt').read().splitlines() if not l.startswith('#')]
hdr = ['# python tools/profile_steps.py efficientdet-d0 512 16   (final code of the round): every launch of one D0 training step replayed ALONE, 10x back to back, HIP events',
       '# (warm caches: the sum is a lower bound for the single-stream order of the step; bytes = the operands of the launch once)']
open(P + '%s_effdet_d0_per_launch.txt' % R, 'w').write('\n'.join(hdr + body) + '\n')
for k in b:
    print(k, b[k]['value'], b[k]['ms_per_step'], b[k]['roofline']['achieved'], b[k]['roofline']['frac'], b[k]['roofline'].get('conv_ms_per_step'))
print(n1.get('infer'))
