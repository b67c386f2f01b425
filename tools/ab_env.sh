#!/bin/bash
# A/B of one environment switch on the SAME box: tools/ab_env.sh VAR [bench args...] runs bench.py with VAR=0 and VAR=1, twice each
VAR=$1; shift
for i in 1 2; do
for w in 0 1; do
env $VAR=$((w*${AB_SCALE:-1})) timeout 300 python bench.py --no-cpu-baseline --steps 40 "$@" > gpurun_out/ab_$w.json 2> gpurun_out/ab_$w.err
python -c "
import json; d=json.load(open('gpurun_out/ab_$w.json')); print('$VAR=$w', d['value'], d['ms_per_step'], d.get('roofline', {}).get('achieved'), d.get('roofline', {}).get('launches_by_config'))"
done
done
