#!/bin/bash
K="graphed_step_equals or written_first or early_adamw or deferred_adamw or wgrad_side or guards_non_finite or segmented_graph"
for m in none collect off; do
  echo "== DAV_TEST_GC=$m"; DAV_TEST_GC=$m timeout 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "$K" 2>&1 | grep -v "^  File\|Extension modules\|^$\|Thread\|no Python" | tail -3
done
