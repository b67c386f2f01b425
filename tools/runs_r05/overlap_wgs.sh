#!/bin/bash
# the gang launch on fewer than 256 workgroups (experiment library: DAV_TN_GANG_WGS), AdamW beside it on a second stream: do the CUs left free take the optimizer pass?
for w in 256 240 224 208 192; do
  echo "== gang workgroups $w"
  DAV_BENCH_LIB=tools/runs_r05/lib_exp/libdavfusion_hip.so DAV_TN_GANG_WGS=$w python tools/runs_r05/overlap_probe.py 2>&1 | grep -v amdgpu.ids | tail -1
done
