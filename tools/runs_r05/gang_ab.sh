#!/bin/bash
# same-box A/B of the gang weight-gradient kernel: the tree's library against tools/runs_r05/lib_base (the build before a change)
mkdir -p gpurun_out/r05
python -m pytest tests/test_hip_kernels.py -q -k "gemm_tn_gang" 2>&1 | tail -3
for i in 1 2; do
  echo "--- tree"; python tools/tn_gang_bench.py enc all dec 2>&1 | grep -v amdgpu.ids | grep "gang"
  echo "--- base"; DAV_BENCH_LIB=tools/runs_r05/lib_base/libdavfusion_hip.so python tools/tn_gang_bench.py enc all dec 2>&1 | grep -v amdgpu.ids | grep "gang"
done
