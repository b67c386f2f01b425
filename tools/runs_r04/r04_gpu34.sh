#!/bin/bash
# L2 hit rate of the grouped weight-gradient kernel (isolated launches of the step's mix), default XCD runs vs every-8th-tile order
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tnl2
out=gpurun_out/r04_tn_l2_hit.txt; : > $out
for xcd in 1 0; do
  rm -rf gpurun_out/tnl2/x$xcd
  DAV_TN_XCD=$xcd timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/tnl2/x$xcd -o t -- python3 tools/tn_group_bench.py 0 1 5 > gpurun_out/tnl2/bench_$xcd.txt 2> gpurun_out/tnl2/err_$xcd.txt
  f=$(find gpurun_out/tnl2/x$xcd -name "*counter_collection.csv" | head -1)
  echo "== DAV_TN_XCD=$xcd" >> $out
  grep -v amdgpu gpurun_out/tnl2/bench_$xcd.txt | tail -4 >> $out
  python3 tools/pmc_summary.py $f TCC_HIT_sum gemm_tn_grouped >> $out
  python3 tools/pmc_summary.py $f TCC_MISS_sum gemm_tn_grouped >> $out
done
find gpurun_out/tnl2 -name "*.csv" -delete
cat $out
