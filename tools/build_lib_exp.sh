#!/bin/bash
# Experiment build of the library (every measured-and-rejected GEMM tile configuration + the timing ablations of DAV_TN_GANG_DEBUG) into
# tools/runs_r05/lib_exp/ — what tools/runs_r05/{gang_ablate,overlap_wgs,dec_serial_ab}.sh and tools/ab_lib_env.sh swap in.  The product
# library in the tree is not touched.  (tools/runs_r05/nt_table_dec52.json, the tuned table with configuration 52 on the audio decoder's
# shapes, was a one-off edit of deepavfusion_amd/tuning/nt_gfx950.json — entries with M = 22528 set to 52 — and is not kept.)
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
T=$(mktemp -d)
mkdir -p $T/deepavfusion_amd $T/include $ROOT/tools/runs_r05/lib_exp
cp -r $ROOT/deepavfusion_amd/csrc $T/deepavfusion_amd/csrc && cp $ROOT/include/dav_kernels.h $T/include/
rm -f $T/deepavfusion_amd/csrc/*.o
make -C $T/deepavfusion_amd/csrc -j8 EXPERIMENTAL=1
cp $T/deepavfusion_amd/libdavfusion_hip.so $ROOT/tools/runs_r05/lib_exp/libdavfusion_hip.so
rm -rf $T
ls -la $ROOT/tools/runs_r05/lib_exp/libdavfusion_hip.so
