#!/bin/bash
# stability of the final tree: 2000 captured steps (default schedule), then 1000 with each opt-in placement
mkdir -p gpurun_out
out=gpurun_out/r04_long_run.txt; : > $out
echo "== default, 2000 steps" >> $out; timeout 600 python tests/long_run_probe.py base 2000 64 2>/dev/null | grep -v amdgpu >> $out
echo "== DAV_DEFER_ADAMW=1, 1000 steps" >> $out; DAV_DEFER_ADAMW=1 timeout 600 python tests/long_run_probe.py base 1000 64 2>/dev/null | grep -v amdgpu >> $out
echo "== DAV_WGRAD_SIDE=1, 1000 steps" >> $out; DAV_WGRAD_SIDE=1 timeout 600 python tests/long_run_probe.py base 1000 64 2>/dev/null | grep -v amdgpu >> $out
cat $out
