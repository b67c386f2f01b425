#!/bin/bash
# tools/ab_lib_env.sh <library.so> "ENV=.." "ENV=.." ...: bench.py step time with another build of the library swapped in (on the
# GPU box's scratch copy of the tree), three alternating rounds
# (the product library is put back whatever happens: an interrupted run must not leave an experiment build in the tree)
set -e
cp deepavfusion_amd/libdavfusion_hip.so /tmp/lib_product_$$.so
trap 'cp /tmp/lib_product_$$.so deepavfusion_amd/libdavfusion_hip.so; rm -f /tmp/lib_product_$$.so' EXIT
cp "$1" deepavfusion_amd/libdavfusion_hip.so; shift
bash tools/ab_env3.sh "$@"
