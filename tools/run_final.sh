# Final pass of a round on the GPU box: the whole GPU test suite, the profile collection (tools/collect_profiles.sh),
# the supplementary bench lines, the PMC traffic passes and the SyncBN kernel profile.  Usage: bash tools/run_final.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/t_all.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/t_all.log
if [ $rc -ge 124 ]; then echo "stopping: runtime abort"; exit 1; fi
bash tools/collect_profiles.sh
timeout -k 10 300 python bench.py --sync-bn --size 640 --steps 10 --warmup 3 > $O/prof_bench_r50_640_syncbn.json 2>/dev/null
timeout -k 10 300 python bench.py --precision fp32 --steps 5 --warmup 2 --no-infer --no-cpu-baseline > $O/prof_bench_r50_512_fp32.json 2>/dev/null
timeout -k 10 300 python bench.py --arch ksevendet --depth 18 --steps 20 --warmup 5 > $O/prof_bench_ksevendet_r18_fpn.json 2>/dev/null
timeout -k 10 300 python bench.py --arch ksevendet --depth 50 --steps 20 --warmup 5 > $O/prof_bench_ksevendet_r50_fpn.json 2>/dev/null
timeout -k 10 300 python bench.py --arch ksevendet-bifpn --depth 18 --steps 20 --warmup 5 > $O/prof_bench_ksevendet_r18_bifpn.json 2>/dev/null
for f in r50_640_syncbn r50_512_fp32 ksevendet_r18_fpn ksevendet_r50_fpn ksevendet_r18_bifpn; do python -c "import json; d=json.loads(open('$O/prof_bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('sync_bn', {}).get('bn_ms_per_step'))"; done
bash tools/collect_traffic.sh r03 r50 d0 d2 || echo "traffic collection failed"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/$O/prof_sbn; (cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_sbn -- python bench.py --sync-bn --size 640 --steps 10 --warmup 3 --no-roofline > $O/prof_sbn.log 2>&1 && python tools/summarize_stats.py $O/prof_sbn "rocprofv3 --kernel-trace --stats -- python bench.py --sync-bn --size 640 --steps 10 --warmup 3 --no-roofline" > $O/prof_syncbn_kernel_stats.txt; find $O/prof_sbn -name "*trace.csv" -delete; head -12 $O/prof_syncbn_kernel_stats.txt | cut -c1-150)
