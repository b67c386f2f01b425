#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/r04_nt_persist2.txt; : > $out
run() { local label="$1"; shift
  local line; line=$(env "$@" timeout 400 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1)
  echo "$label $(echo "$line" | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["ms_per_step"], j["value"], j["loss"])')" >> $out; }
for rep in 1 2 3; do
  run "one tile per workgroup " DAV_NT_PERSIST=0
  run "persistent 256         " DAV_NT_PERSIST=256
  run "persistent 384         " DAV_NT_PERSIST=384
  run "persistent 512         " DAV_NT_PERSIST=512
done
cat $out
