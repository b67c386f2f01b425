#!/bin/bash
# same-box comparison of arbitrary environment settings: tools/ab_env2.sh "A=1 B=2" "A=0" ... (two rounds, bench --steps 40)
for i in 1 2; do
for S in "$@"; do
env $S timeout 300 python bench.py --no-cpu-baseline --steps 40 > gpurun_out/abe.json 2> gpurun_out/abe.err
python -c "
import json; d=json.load(open('gpurun_out/abe.json')); print('$S |', d['value'], d['ms_per_step'], d.get('roofline', {}).get('achieved'), d.get('roofline', {}).get('launches_by_config'))"
done
done
