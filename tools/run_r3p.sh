R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out; mkdir -p $O
b() { timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-infer --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step_median'], d['roofline']['conv_ms_per_step'])"; }
b with_stats_code; b with_stats_code
touch objectdetection/pytorch_amd/csrc/conv_gather.hip; make -C objectdetection/pytorch_amd/csrc -j8 EXTRA=-DMI355_NO_EPI_STATS > $O/build_b.log 2>&1 || { tail -3 $O/build_b.log; exit 1; }
b without; b without
touch objectdetection/pytorch_amd/csrc/conv_gather.hip; make -C objectdetection/pytorch_amd/csrc -j8 > $O/build_a.log 2>&1
b with_stats_code; b with_stats_code
timeout -k 10 300 python -m pytest tests/test_gpu_conv.py -q -k epilogue_statistics 2>&1 | tail -2
