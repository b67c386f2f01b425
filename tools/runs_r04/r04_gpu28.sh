#!/bin/bash
K="graphed_step_equals or written_first or early_adamw or deferred_adamw or wgrad_side or guards_non_finite or segmented_graph"
for i in 1 2 3 4 5; do
  echo "== run $i (default GC, no collect)"; DAV_TEST_GC=none timeout 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "$K" 2>&1 | grep -v "^  File\|Extension modules\|^$\|Thread\|no Python" | grep "Error\|passed\|failed\|^E \|Fatal" | cut -c1-400 | head -12
done
