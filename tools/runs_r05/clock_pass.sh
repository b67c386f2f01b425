#!/bin/bash
# effective shader clock per kernel family over eager steps (PMC serialises the kernels: the clock each kernel gets ALONE on the GPU)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05/clk; rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT -o c -- python3 bench.py --no-graph --no-roofline --no-cpu-baseline --steps 3 --warmup 2 > $OUT/bench.json 2> $OUT/err.txt
python3 tools/clock_in_step.py $OUT > gpurun_out/r05/clock_per_kernel.txt 2>&1
cat gpurun_out/r05/clock_per_kernel.txt
find $OUT -name "*.csv" -delete
