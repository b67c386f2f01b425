#!/bin/bash
# kernel trace of the captured step -> timeline summaries (busy / concurrency / who runs alone): bash tools/timeline_step.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/tl; mkdir -p $OUT
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python3 bench.py --no-cpu-baseline --no-roofline --steps 12 --warmup 3 > $OUT/bench.json 2> $OUT/prof.err
T=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
MS=$(python3 -c "import json; print(json.load(open('$OUT/bench.json'))['ms_per_step'])")
echo "step $MS ms (profiled)"
python3 tools/trace_timeline.py $T $MS > $OUT/timeline.txt 2>&1; head -40 $OUT/timeline.txt
python3 tools/trace_alone.py $T $MS > $OUT/alone.txt 2>&1; head -40 $OUT/alone.txt
find $OUT -name "*.csv" -delete
