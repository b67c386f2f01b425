#!/bin/bash
# weight gradients as a parallel graph branch: parity test, then the A/B (separate processes, alternated)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "wgrad_side or written_first" > gpurun_out/r04_wgrad_side_test.log 2>&1
tail -4 gpurun_out/r04_wgrad_side_test.log
out=gpurun_out/r04_wgrad_side.txt; : > $out
run() { local label="$1"; shift
  local line; line=$(env "$@" timeout 400 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1)
  echo "$label $(echo "$line" | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["ms_per_step"], j["value"])')" >> $out; }
for rep in 1 2 3; do
  run "serial placement (default) " DAV_WGRAD_SIDE=0
  run "side stream                " DAV_WGRAD_SIDE=1
  run "side stream, high priority " DAV_WGRAD_SIDE=1 DAV_WGRAD_SIDE_PRIO=-1
done
cat $out
