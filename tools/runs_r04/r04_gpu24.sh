#!/bin/bash
t() { echo "== $1"; timeout 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "$1" 2>&1 | grep -v "^  File\|Extension modules\|^$\|Thread\|no Python" | tail -3; }
t "graphed_step_equals or written_first or early_adamw or deferred_adamw or wgrad_side or guards_non_finite or segmented_graph"
t "graphed_step_equals or written_first or early_adamw or deferred_adamw or guards_non_finite or segmented_graph"
t "graphed_step_equals or written_first or early_adamw or wgrad_side or guards_non_finite or segmented_graph"
t "graphed_step_equals or written_first or deferred_adamw or wgrad_side or guards_non_finite or segmented_graph"
t "early_adamw or wgrad_side or segmented_graph"
t "early_adamw or segmented_graph"
t "written_first or wgrad_side or segmented_graph"
t "deferred_adamw or wgrad_side or segmented_graph"
