#!/bin/bash
# throughput-optimal vs latency-optimal tile choice: fewer, fatter tiles for the mid-size NT launches (separate processes, alternated)
mkdir -p gpurun_out
out=gpurun_out/r04_tile_choice_ab.txt
: > $out
run() { # label, env...
  local label="$1"; shift
  local line
  line=$(env "$@" timeout 400 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1)
  echo "$label $(echo "$line" | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["ms_per_step"], j["value"])')" >> $out
}
for rep in 1 2; do
  run "default            " DAV_X=0
  run "T8=0 (no 128x64)   " DAV_NT_T8=0
  run "T8=250             " DAV_NT_T8=250
  run "TUNE=0             " DAV_NT_TUNE=0
  run "TUNE=0 T8=0        " DAV_NT_TUNE=0 DAV_NT_T8=0
  run "TUNE=0 T8=250      " DAV_NT_TUNE=0 DAV_NT_T8=250
  run "TUNE=0 T8=0 T5=50  " DAV_NT_TUNE=0 DAV_NT_T8=0 DAV_NT_T5=50
done
cat $out
