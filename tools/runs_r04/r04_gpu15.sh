#!/bin/bash
# deferred AdamW: parity test, then the A/B
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "deferred_adamw or early_adamw or graphed_step_equals or guards_non_finite" > gpurun_out/r04_defer_test.log 2>&1
tail -5 gpurun_out/r04_defer_test.log
timeout 900 python tools/defer_adamw_ab.py > gpurun_out/r04_defer_ab.txt 2>&1
tail -8 gpurun_out/r04_defer_ab.txt
