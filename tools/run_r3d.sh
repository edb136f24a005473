R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_trainer.py tests/test_gpu_fullsize.py tests/test_gpu_bn_plan.py -q > $O/t7.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -E "passed|failed|^FAILED" $O/t7.log | tail -8
if [ $rc -ge 124 ]; then echo "stopping: runtime abort"; exit 1; fi
for x in 0 1 0 1; do MI355_WG_XCD=$x timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-infer --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('XCD=$x', d['value'], d['ms_per_step_median'], d['roofline']['conv_ms_per_step'])"; done
timeout -k 10 300 python bench.py --sync-bn --size 640 --steps 10 --warmup 3 > $O/b_syncbn3.json 2>$O/b_syncbn3.err
python -c "import json; d=json.loads(open('$O/b_syncbn3.json').read().strip().splitlines()[-1]); print('syncbn', d['value'], d['ms_per_step'], d['sync_bn']['bn_ms_per_step'])"
bash tools/collect_traffic.sh r03 r50 || exit 1
bash tools/run_fault_hunt.sh bench
