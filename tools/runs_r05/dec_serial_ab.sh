#!/bin/bash
# both decoders on one stream (DAV_DEC_SERIAL=1, experiment switch) with and without 192-row tiles on the audio decoder's shapes
# (experiment library: configuration 52 dispatchable from the tuned table), against the default schedule; same box, three rounds
# (the product library is put back whatever happens: an interrupted run must not leave an experiment build in the tree)
set -e
cp deepavfusion_amd/libdavfusion_hip.so /tmp/lib_product_$$.so
trap 'cp /tmp/lib_product_$$.so deepavfusion_amd/libdavfusion_hip.so; rm -f /tmp/lib_product_$$.so' EXIT
cp tools/runs_r05/lib_exp/libdavfusion_hip.so deepavfusion_amd/libdavfusion_hip.so
bash tools/ab_env3.sh "DAV_DEC_SERIAL=0" "DAV_DEC_SERIAL=1" "DAV_DEC_SERIAL=1 DAV_NT_TUNE_FILE=tools/runs_r05/nt_table_dec52.json" "DAV_DEC_SERIAL=0 DAV_NT_TUNE_FILE=tools/runs_r05/nt_table_dec52.json"
