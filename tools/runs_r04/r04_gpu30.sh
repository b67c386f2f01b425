#!/bin/bash
# persistent walk of the tiles (DAV_NT_PERSIST=<workgroups>) against one tile per workgroup, isolated, then in the step
mkdir -p gpurun_out
out=gpurun_out/r04_nt_persist.txt; : > $out
for shape in "6080 2304 768" "6080 3072 768" "6080 768 3072" "5184 2304 768" "14592 1536 512" "22528 2048 512"; do
  for pz in 0 512 256; do
    echo -n "PERSIST=$pz  " >> $out
    DAV_NT_PERSIST=$pz timeout 120 python tools/gemm_one.py $shape 3 44 2>/dev/null | grep -v amdgpu | tr '\n' '|' >> $out; echo >> $out
  done
done
run() { local label="$1"; shift
  local line; line=$(env "$@" timeout 400 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1)
  echo "$label $(echo "$line" | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["ms_per_step"], j["value"], j["loss"])')" >> $out; }
for rep in 1 2; do
  run "step, one tile per workgroup " DAV_NT_PERSIST=0
  run "step, persistent 512         " DAV_NT_PERSIST=512
done
cat $out
