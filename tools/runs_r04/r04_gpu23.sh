#!/bin/bash
mkdir -p gpurun_out
echo "== segmented alone"; timeout 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "segmented_graph_step" 2>&1 | tail -3
echo "== wgrad_side then segmented"; timeout 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "wgrad_side or segmented_graph_step" 2>&1 | tail -3
echo "== deferred then segmented"; timeout 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "deferred_adamw or segmented_graph_step" 2>&1 | tail -3
echo "== guards then segmented"; timeout 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "guards_non_finite or segmented_graph_step" 2>&1 | tail -3
