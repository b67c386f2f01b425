#!/bin/bash
# N alternating rounds of bench.py --steps 40 under two or more environment settings on one box: tools/runs_r06/ab_many.sh N "A=1" "A=2" ...
N=$1; shift
for i in $(seq 1 $N); do
for S in "$@"; do
env $S timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 40 > gpurun_out/abm.json 2> gpurun_out/abm.err
python3 -c "
import json; d=json.loads(open('gpurun_out/abm.json').read().strip().splitlines()[-1]); print('$S |', d['ms_per_step'], d['ms_per_step_quantiles_device_events']['p50'])"
done
done | sed -e "s#$GRAFT_REPO_ROOT/##"
