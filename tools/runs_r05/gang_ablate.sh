#!/bin/bash
# timing ablations of the gang kernel's main loop (experiment library with DAV_TN_GANG_DEBUG bit 32 = no fragment reads):
# 2 no epilogue, 6 + no MFMAs (operand stream alone), 38 + no reads (DMA + barriers alone), 34 MFMAs + DMA without reads
L=tools/runs_r05/lib_exp/libdavfusion_hip.so
for d in 0 2 6 38 34; do
  echo "== DAV_TN_GANG_DEBUG=$d"
  DAV_BENCH_LIB=$L TN_BENCH_CHECK=0 DAV_TN_GANG_DEBUG=$d python tools/tn_gang_bench.py enc dec big 2>&1 | grep -v amdgpu.ids | grep "ONE launch\|decoders: gang\|single problem"
done
