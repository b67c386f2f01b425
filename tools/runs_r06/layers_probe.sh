#!/bin/bash
# how often a layer of the replayed step runs its three streams one after the other (tools/trace_layers.py), and the unprofiled step-time distribution
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/ly -o ly -- python3 bench.py --no-cpu-baseline --no-roofline --steps 24 --warmup 3 > $OUT/ly.json 2> $OUT/ly.err
T=$(find $OUT/ly -name "*kernel_trace.csv" | head -1)
python3 tools/trace_layers.py $T > $OUT/r06_layers_base.txt 2>&1
rm -rf $OUT/ly
timeout 300 python3 bench.py --no-cpu-baseline --no-roofline --steps 200 --warmup 5 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unprofiled, 200 replays:', d['ms_per_step'], d['ms_per_step_quantiles_device_events'])" > $OUT/r06_step_hist.txt 2>&1
cat $OUT/r06_layers_base.txt $OUT/r06_step_hist.txt
