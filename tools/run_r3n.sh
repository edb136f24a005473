R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_retinanet.py tests/test_gpu_kseven.py tests/test_gpu_trainer.py tests/test_gpu_fullsize.py tests/test_gpu_bn_plan.py -q > $O/t14.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -E "passed|failed|^FAILED" $O/t14.log | tail -8
if [ $rc -ge 124 ]; then echo "stopping: runtime abort"; exit 1; fi
for x in 0 1 0 1; do MI355_TABLE_STREAM=$x timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-infer --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TABLE_STREAM=$x', d['value'], d['ms_per_step_median'], d['roofline']['conv_ms_per_step'])"; done
timeout -k 10 200 python tools/side_balance.py 2>&1 | tail -2
for x in 0 1; do MI355_TABLE_STREAM=$x timeout -k 10 200 python bench.py --size 640 --steps 20 --warmup 5 --no-infer --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('640 TABLE_STREAM=$x', d['value'], d['ms_per_step_median'])"; done
python tools/nccl_rehearse.py 2>&1 | tail -1
