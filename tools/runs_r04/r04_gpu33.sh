#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/r04_nt_persist4.txt; : > $out
run() { local label="$1"; shift
  local line; line=$(env "$@" timeout 400 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1)
  echo "$label $(echo "$line" | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["ms_per_step"], j["value"], j["loss"])')" >> $out; }
for rep in 1 2 3 4; do
  run "one tile per workgroup " DAV_NT_PERSIST=0
  run "persistent 512         " DAV_NT_PERSIST=512
  run "persistent 448         " DAV_NT_PERSIST=448
done
cat $out
DAV_NT_PERSIST=512 timeout 1200 python -m pytest tests/test_hip_kernels.py tests/test_hip_parity.py -x -q -m gpu -k "kernel_family or fuzz or base-64 or base-4 or large-2 or schedule_independent or graphed_step_equals or end_to_end_vs_oracle or batch64 or video_base_vs" 2>&1 | tail -4
