#!/bin/bash
# 4-wave 128x128 tile configurations (64x64 wave tiles: -33 % LDS bytes per MFMA) against the 8-wave default, EXPERIMENTAL build:
#   bash tools/exp_4wave.sh path/to/lib_exp.so
L=deepavfusion_amd/libdavfusion_hip.so; cp $L /tmp/lib_keep.so; cp $1 $L
for shape in "4032 3072 768" "6080 2304 768" "22528 512 2048" "22528 2048 512" "22528 1536 512" "14592 512 2048" "4096 4096 4096"; do
  python tools/gemm_one.py $shape 3 1 2 12 15 16 17 44 2>/dev/null | grep cfg
done
cp /tmp/lib_keep.so $L
