#!/bin/bash
# both decoders on one stream (DAV_DEC_SERIAL=1, experiment switch) with and without 192-row tiles on the audio decoder's shapes
# (experiment library: configuration 52 dispatchable from the tuned table), against the default schedule; same box, three rounds
cp deepavfusion_amd/libdavfusion_hip.so /tmp/lib_product.so
cp tools/runs_r05/lib_exp/libdavfusion_hip.so deepavfusion_amd/libdavfusion_hip.so
bash tools/ab_env3.sh "DAV_DEC_SERIAL=0" "DAV_DEC_SERIAL=1" "DAV_DEC_SERIAL=1 DAV_NT_TUNE_FILE=tools/runs_r05/nt_table_dec52.json" "DAV_DEC_SERIAL=0 DAV_NT_TUNE_FILE=tools/runs_r05/nt_table_dec52.json"
cp /tmp/lib_product.so deepavfusion_amd/libdavfusion_hip.so
