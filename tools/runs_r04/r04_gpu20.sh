#!/bin/bash
# which operand's row stride matters for the K = 768 GEMMs (and does K = 2304 / 3072 / 512 care)?
mkdir -p gpurun_out
out=gpurun_out/r04_gemm_pad.txt; : > $out
for shape in "5184 2304 768" "6080 3072 768" "6080 768 768" "6080 768 3072" "5184 768 2304" "14592 1536 512"; do
  for w in a b c ab abc; do
    PAD_WHICH=$w timeout 120 python tools/gemm_pad.py $shape 0 64 2>/dev/null | grep -v amdgpu >> $out
  done
  echo >> $out
done
cat $out
