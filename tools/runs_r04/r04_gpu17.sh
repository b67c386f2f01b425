#!/bin/bash
# full GPU suite on the final tree + a longer deferred-AdamW A/B
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/r04_gputest.log 2>&1
tail -25 gpurun_out/r04_gputest.log
timeout 900 python tools/defer_adamw_ab.py --rounds 12 > gpurun_out/r04_defer_ab3.txt 2>&1
grep -v "^JSON\|amdgpu.ids" gpurun_out/r04_defer_ab3.txt
