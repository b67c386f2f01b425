#!/bin/bash
# same-box comparison of the step schedules with the 256 x 256 GEMM configuration on / off: tools/ab_sched.sh [bench args...]
for sched in "DAV_DUMMY=0" "DAV_BATCH=1" "DAV_BATCH=1 DAV_FUSION_STREAM=0"; do
for w in 0 1; do
env $sched DAV_NT256=$w ${AB_EXTRA} timeout 300 python bench.py --no-cpu-baseline --steps 40 "$@" > gpurun_out/abs.json 2> gpurun_out/abs.err
python -c "
import json; d=json.load(open('gpurun_out/abs.json')); print('$sched NT256=$w', d['value'], d['ms_per_step'], d.get('roofline', {}).get('achieved'), d.get('roofline', {}).get('launches_by_config'))"
done
done
