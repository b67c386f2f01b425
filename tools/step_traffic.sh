#!/bin/bash
# Memory-side traffic of the whole training step by kernel family (two PMC passes): bash tools/step_traffic.sh [bench.py args]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/traffic; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/$c -o p -- python3 bench.py --no-graph --no-roofline --no-cpu-baseline --steps 3 --warmup 1 "$@" > $OUT/bench_$c.json 2> $OUT/$c.err
done
python3 tools/step_traffic.py $(find $OUT/FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $OUT/WRITE_SIZE -name "*counter_collection.csv" | head -1) > $OUT/step_traffic.txt 2>&1
find $OUT -name "*.csv" -delete
cat $OUT/step_traffic.txt
