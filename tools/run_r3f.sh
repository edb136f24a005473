R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out; mkdir -p $O; rm -f gpucore.*
timeout -k 10 300 python bench.py --precision fp32 --steps 5 --warmup 2 --no-infer --no-cpu-baseline > $O/b_fp32.json 2> $O/b_fp32.err; rc=$?; echo "bench fp32 rc=$rc"; head -c 600 $O/b_fp32.json; echo
if [ $rc -ne 0 ]; then tail -3 $O/b_fp32.err; exit 1; fi
timeout -k 10 400 python tools/find_fault.py --steps 120 > $O/ff_e.log 2>&1; rc=$?; echo "find_fault two-stream rc=$rc"; tail -1 $O/ff_e.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 600 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_fp32_mode.py tests/test_gpu_conv.py tests/test_gpu_conv_f32.py tests/test_gpu_trainer.py -q -s > $O/t8.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -E "passed|failed|^FAILED|trajectory r" $O/t8.log | tail -8
if [ $rc -ge 124 ]; then echo "stopping: runtime abort"; exit 1; fi
for x in 0 1 0 1; do MI355_WG_XCD=$x timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-infer --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('XCD=$x', d['value'], d['ms_per_step_median'], d['roofline']['conv_ms_per_step'])"; done
for x in 0 1; do MI355_WG_XCD=$x timeout -k 10 200 python tools/bench_head_wgrad.py 50 256 2>&1 | tail -1; done
bash tools/collect_traffic.sh r03 r50 || exit 1
timeout -k 10 300 python tools/profile_plan.py 50 512 16 > $O/prof_conv_per_launch.txt 2>/dev/null; tail -3 $O/prof_conv_per_launch.txt
