#!/bin/bash
# Round profiles in one gpurun call: bench lines, rocprofv3 kernel stats, per-launch tables -> gpurun_out/prof_* (copy the
# ones to keep into profiles/).  Usage (on the GPU box): bash tools/collect_profiles.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out
timeout -k 10 300 python bench.py --steps 40 --warmup 10 > $O/prof_bench_n1.json 2>/dev/null &&
timeout -k 10 300 python bench.py --size 640 --steps 20 --warmup 5 --no-infer --no-cpu-baseline > $O/prof_bench_r50_640.json 2>/dev/null &&
timeout -k 10 300 python bench.py --arch efficientdet-d0 --steps 30 --warmup 5 --no-infer > $O/prof_bench_effdet_d0.json 2>/dev/null &&
timeout -k 10 300 python bench.py --arch efficientdet-d2 --size 768 --batch 8 --steps 30 --warmup 5 --no-infer > $O/prof_bench_effdet_d2.json 2>/dev/null &&
rm -rf $O/prof_s1 $O/prof_s2 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s1 -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-infer > $O/prof_s1.log 2>&1 &&
python tools/summarize_stats.py $O/prof_s1 "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-infer" > $O/prof_rocprofv3_kernel_stats.txt &&
cp $(find $O/prof_s1 -name "*kernel_stats.csv" | head -1) $O/prof_rocprofv3_kernel_stats.csv &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s2 -- python bench.py --arch efficientdet-d0 --steps 20 --warmup 5 --no-cpu-baseline --no-infer > $O/prof_s2.log 2>&1 &&
python tools/summarize_stats.py $O/prof_s2 "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --arch efficientdet-d0 --steps 20 --warmup 5 --no-cpu-baseline --no-infer" > $O/prof_effdet_d0_kernel_stats.txt
find $O/prof_s1 $O/prof_s2 -name "*trace.csv" -delete
timeout -k 10 300 python tools/profile_plan.py 50 512 16 > $O/prof_conv_per_launch.txt 2>/dev/null
timeout -k 10 300 python tools/profile_steps.py efficientdet-d0 512 16 > $O/prof_effdet_d0_per_launch.txt 2>/dev/null
for f in n1 r50_640 effdet_d0 effdet_d2; do head -c 330 $O/prof_bench_$f.json; echo; done
