R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/t_all.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/t_all.log
if [ $rc -ge 124 ]; then echo "stopping: runtime abort"; exit 1; fi
grep -E "trajectory r" $O/t_all.log
bash tools/collect_profiles.sh
timeout -k 10 300 python bench.py --sync-bn --size 640 --steps 10 --warmup 3 > $O/prof_bench_r50_640_syncbn.json 2>/dev/null
timeout -k 10 300 python bench.py --arch ksevendet --depth 18 --steps 20 --warmup 5 > $O/prof_bench_ksevendet_r18_fpn.json 2>/dev/null
timeout -k 10 300 python bench.py --arch ksevendet --depth 50 --steps 20 --warmup 5 > $O/prof_bench_ksevendet_r50_fpn.json 2>/dev/null
timeout -k 10 300 python bench.py --arch ksevendet-bifpn --depth 18 --steps 20 --warmup 5 > $O/prof_bench_ksevendet_r18_bifpn.json 2>/dev/null
for f in r50_640_syncbn ksevendet_r18_fpn ksevendet_r50_fpn ksevendet_r18_bifpn; do python -c "import json; d=json.loads(open('$O/prof_bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
