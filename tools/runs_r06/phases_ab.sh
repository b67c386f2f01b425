#!/bin/bash
# phase boundaries of the captured step under two settings: bash tools/runs_r06/phases_ab.sh "A=1" "A=0"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for S in "$@"; do
  OUT=gpurun_out/ph; rm -rf $OUT; mkdir -p $OUT
  env $S timeout 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python3 bench.py --no-cpu-baseline --no-roofline --steps 12 --warmup 3 > $OUT/bench.json 2> $OUT/prof.err
  T=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
  MS=$(python3 -c "import json; print(json.load(open('$OUT/bench.json'))['ms_per_step'])")
  echo "== $S: step $MS ms (profiled)"
  python3 tools/trace_phases.py $T $MS
  python3 tools/trace_alone.py $T $MS | head -14
  find $OUT -name "*.csv" -delete
done
