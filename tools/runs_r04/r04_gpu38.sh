#!/bin/bash
# HIP runtime tunables that touch graph execution / kernel-argument placement / queues: one bench each (then repeat the interesting ones)
mkdir -p gpurun_out
out=gpurun_out/r04_runtime_env.txt; : > $out
run() { local label="$1"; shift
  local line; line=$(env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1)
  echo "$label $(echo "$line" | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["ms_per_step"], j["value"], j["loss"])' 2>&1 | tail -1)" >> $out; }
run "default                          " DAV_X=0
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 " DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 " DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run "HIP_FORCE_DEV_KERNARG=0          " HIP_FORCE_DEV_KERNARG=0
run "HIP_FORCE_DEV_KERNARG=1          " HIP_FORCE_DEV_KERNARG=1
run "GPU_MAX_HW_QUEUES=8              " GPU_MAX_HW_QUEUES=8
run "GPU_MAX_HW_QUEUES=2              " GPU_MAX_HW_QUEUES=2
run "DEBUG_HIP_FORCE_GRAPH_QUEUES=4   " DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run "DEBUG_HIP_FORCE_GRAPH_QUEUES=8   " DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run "DEBUG_HIP_DYNAMIC_QUEUES=0       " DEBUG_HIP_DYNAMIC_QUEUES=0
run "DEBUG_HIP_DYNAMIC_QUEUES=1       " DEBUG_HIP_DYNAMIC_QUEUES=1
run "DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0 " DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0
run "DEBUG_HIP_KERNARG_COPY_OPT=0     " DEBUG_HIP_KERNARG_COPY_OPT=0
run "GPU_STREAMOPS_CP_WAIT=1          " GPU_STREAMOPS_CP_WAIT=1
run "default                          " DAV_X=0
cat $out
