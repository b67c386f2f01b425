#!/bin/bash
# Single-rank cost of the data-parallel machinery (5 graph segments + 1-rank RCCL) on this tree, same box, alternated:
#   bash tools/runs_r06/dp1_tax.sh > gpurun_out/r06_dp1_tax.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
  for w in 0 1; do
    DAV_FORCE_DIST=$w timeout 400 python bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 5 > gpurun_out/dp1_$w.json 2> gpurun_out/dp1_$w.err
    python - <<PY
import json
d = json.load(open('gpurun_out/dp1_$w.json'))
print('DAV_FORCE_DIST=$w  pairs/s', d['value'], ' ms_per_step', d['ms_per_step'], ' dp', d.get('dp'))
PY
  done
done
