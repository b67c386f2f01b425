#!/bin/bash
K="graphed_step_equals or written_first or early_adamw or deferred_adamw or wgrad_side or guards_non_finite or segmented_graph"
cd $GRAFT_REPO_ROOT
timeout 600 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGSEGV stop print" -ex run -ex "bt 40" -ex "info threads" --args python3 -m pytest tests/test_hip_parity.py -x -q -m gpu -k "$K" > gpurun_out/r04_gdb.txt 2>&1
grep -n "SIGSEGV" -A60 gpurun_out/r04_gdb.txt | head -120
