#!/bin/bash
# tools/ab_lib_env.sh <library.so> "ENV=.." "ENV=.." ...: bench.py step time with another build of the library swapped in (on the
# GPU box's scratch copy of the tree), three alternating rounds
cp deepavfusion_amd/libdavfusion_hip.so /tmp/lib_product.so
cp "$1" deepavfusion_amd/libdavfusion_hip.so; shift
bash tools/ab_env3.sh "$@"
cp /tmp/lib_product.so deepavfusion_amd/libdavfusion_hip.so
