R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_bn_plan.py tests/test_gpu_effdet.py tests/test_gpu_kseven.py tests/test_gpu_retinanet.py tests/test_gpu_syncbn.py -q > $O/t6.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -E "passed|failed|^FAILED" $O/t6.log | tail -8
if [ $rc -ge 124 ]; then echo "stopping: runtime abort"; exit 1; fi
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout -k 10 300 python bench.py --sync-bn --size 640 --steps 10 --warmup 3 > $O/b_syncbn2.json 2>$O/b_syncbn2.err)
python -c "import json; d=json.loads(open('$R/$O/b_syncbn2.json').read().strip().splitlines()[-1]); print('syncbn', d['value'], d['ms_per_step'], d['host_loop_ms_per_step'], d['sync_bn'])"
rm -rf $R/$O/prof_sbn; (cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_sbn -- python bench.py --sync-bn --size 640 --steps 10 --warmup 3 --no-roofline > $O/prof_sbn.log 2>&1 && python tools/summarize_stats.py $O/prof_sbn "rocprofv3 --kernel-trace --stats -- python bench.py --sync-bn --size 640 --steps 10 --warmup 3 --no-roofline" > $O/prof_syncbn_kernel_stats.txt; find $O/prof_sbn -name "*trace.csv" -delete; head -16 $O/prof_syncbn_kernel_stats.txt | cut -c1-160)
