#!/bin/bash
# fused AdamW: parity case, then the A/B (separate processes, alternated)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/opt_in_schedule_cases.py -x -q -m gpu -p no:cacheprovider -k "fused_adamw" > gpurun_out/r04_fused_adamw_test.log 2>&1
tail -15 gpurun_out/r04_fused_adamw_test.log | cut -c1-220
out=gpurun_out/r04_fused_adamw.txt; : > $out
run() { local label="$1"; shift
  local line; line=$(env "$@" timeout 400 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline 2>gpurun_out/r04_fused_err.txt | tail -1)
  echo "$label $(echo "$line" | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["ms_per_step"], j["value"], j["loss"])' 2>&1 | tail -1)" >> $out; }
for rep in 1 2 3; do
  run "optimizer kernel (default) " DAV_FUSED_ADAMW=0
  run "fused into the wgrad tiles " DAV_FUSED_ADAMW=1
done
cat $out; tail -3 gpurun_out/r04_fused_err.txt
