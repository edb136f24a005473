#!/bin/bash
# Step-time A/B of launch-policy knobs (each configuration = one fresh process: the knobs are read when the library /
# plan starts).  Usage on the GPU box: bash tools/sweep_step.sh > gpurun_out/sweep.log
run() {
  echo "== $*"
  env "$@" python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-infer 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('   value %.1f img/s  mean %.3f ms  median %.3f ms' % (d['value'], d['ms_per_step'], d['ms_per_step_median']))
"
}
run A=0
run MI355_WG_GROUP=0
run MI355_SIDE_STREAMS=2
run MI355_SEG_MB=64
run MI355_SIDE_STREAMS=2 MI355_SEG_MB=64
run MI355_SEG_MB=1000
run A=1
