#!/bin/bash
# same-box comparison of two builds of the library: tools/ab_lib.sh old.so new.so  (alternating, bench --steps 40; restores new.so)
# (the product library is put back whatever happens: an interrupted run must not leave an experiment build in the tree)
set -e
cp deepavfusion_amd/libdavfusion_hip.so /tmp/lib_product_$$.so
trap 'cp /tmp/lib_product_$$.so deepavfusion_amd/libdavfusion_hip.so; rm -f /tmp/lib_product_$$.so' EXIT
L=deepavfusion_amd/libdavfusion_hip.so
for i in 1 2 3; do
for S in "$1" "$2"; do
cp $S $L
timeout 300 python bench.py --no-cpu-baseline --steps 40 > gpurun_out/abl.json 2> gpurun_out/abl.err
python -c "
import json; d=json.load(open('gpurun_out/abl.json')); print('$S |', d['value'], d['ms_per_step'])"
done
done
cp "$2" $L
