#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "deferred_adamw" > gpurun_out/r04_defer_test.log 2>&1
tail -3 gpurun_out/r04_defer_test.log
: > gpurun_out/r04_defer_ab2.txt
for w in 0 512 256 128 64 32; do
  echo "== DAV_ADAMW_WGS=$w (variants defer,deferB only see the cap beside the forward; plain pays it alone)" >> gpurun_out/r04_defer_ab2.txt
  DAV_ADAMW_WGS=$w timeout 600 python tools/defer_adamw_ab.py --rounds 5 --variants plain,defer,deferB 2>&1 | grep -v "^JSON\|amdgpu.ids" >> gpurun_out/r04_defer_ab2.txt
done
cat gpurun_out/r04_defer_ab2.txt
