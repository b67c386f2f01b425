#!/bin/bash
K="graphed_step_equals or written_first or early_adamw"
for i in 1 2 3 4; do
  echo "== run $i (default GC, no collect)"; DAV_TEST_GC=none timeout 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "$K" 2>&1 | grep -v "^  File\|Extension modules\|^$\|Thread\|no Python" | grep -n "Error\|assert\|passed\|failed\|^E " | head -12
done
