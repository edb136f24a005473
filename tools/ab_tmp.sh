R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_wgrad_store_forms.py -x -q -m gpu -k "wgrad or grouped or store" 2>&1 | tail -3 &&
for v in 1 1 0; do MI355_X3_M32=$v timeout -k 10 120 python tools/bench_head_wgrad.py 200 | tail -1; done &&
MI355_X3_M32=0 timeout -k 10 500 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "ring" 2>&1 | tail -2 &&
timeout -k 10 120 python tools/bench_head_wgrad.py 100 768 | tail -1
