R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 800 python -m pytest tests/test_gpu_conv.py tests/test_gpu_retinanet.py -x -q -m gpu 2>&1 | tail -3 &&
for w in 0 1 0 1; do MI355_WG_WIDE=$w timeout -k 10 200 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-infer --no-roofline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('WIDE', '$w', d['value'], d['ms_per_step'])"; done
