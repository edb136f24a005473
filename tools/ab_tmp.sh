R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in 0 1 0 1 0 1; do MI355_GATHER_ONE=$v timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-infer --no-roofline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one $v', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; done
