#!/bin/bash
# Cost of the data-parallel step structure on ONE GPU (1-rank RCCL group, DAV_FORCE_DIST=1): bench.py with the step captured
# as 1..8 graph segments (collectives between them) next to the single-graph step without a process group.
run() { # label, env...
  label=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 40 > gpurun_out/dpseg.json 2> gpurun_out/dpseg_$label.err
  python -c "
import json; d=json.load(open('gpurun_out/dpseg.json')); print('$label', d['value'], d['ms_per_step'])" || tail -3 gpurun_out/dpseg_$label.err
}
run no_dist DAV_FORCE_DIST=0
for s in 1 2 3 5 8; do run dist_seg$s DAV_FORCE_DIST=1 DAV_SEGMENTS=$s; done
run no_dist DAV_FORCE_DIST=0
