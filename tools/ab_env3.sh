#!/bin/bash
# same-box comparison of environment settings, step time only (no roofline replays): tools/ab_env3.sh "A=1 B=2" "A=0" ... (three rounds)
for i in 1 2 3; do
for S in "$@"; do
env $S timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 10 > gpurun_out/abe.json 2> gpurun_out/abe.err
python -c "
import json; d=json.loads(open('gpurun_out/abe.json').read().strip().splitlines()[-1]); print('$S |', d['value'], d['ms_per_step'], d.get('secondary', {}).get('ms_per_step'))"
done
done
