R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out; mkdir -p $O
timeout -k 10 200 python tools/side_balance.py 2>&1 | tail -6
for x in 1 0 1 0; do MI355_WG_GROUP_L1=$x timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-infer --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GROUP_L1=$x', d['value'], d['ms_per_step_median'], d['roofline']['conv_ms_per_step'])"; done
MI355_WG_GROUP_L1=0 timeout -k 10 200 python tools/side_balance.py 2>&1 | tail -6
