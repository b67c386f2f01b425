#!/bin/bash
# The round-5 tree (commit 4af19af: git archive into tools/runs_r06/r05tree + its own library build, see .gitignore) against this tree,
# default settings, same box, alternating: bash tools/runs_r06/r05_vs_r06.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  ( cd tools/runs_r06/r05tree && timeout 400 python bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/r05t.json 2> $GRAFT_REPO_ROOT/gpurun_out/r05t.err )
  python -c "import json; d=json.load(open('gpurun_out/r05t.json')); print('round-5 tree  |', d['value'], d['ms_per_step'])"
  timeout 400 python bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 5 > gpurun_out/r06t.json 2> gpurun_out/r06t.err
  python -c "import json; d=json.load(open('gpurun_out/r06t.json')); print('this tree     |', d['value'], d['ms_per_step'])"
done
