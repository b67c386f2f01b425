#!/bin/bash
# the decoders' branches in the captured step: A/B of DAV_DEC_SCHED settings on one box + per-queue trace of each (tools/trace_streams.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06; mkdir -p $OUT
bash tools/ab_env2.sh "$@" > $OUT/dec_sched_ab.txt 2>&1
cat $OUT/dec_sched_ab.txt
i=0
for S in "$@"; do
  i=$((i+1))
  export $S
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/ds$i -o st -- python3 bench.py --no-cpu-baseline --no-roofline --steps 6 --warmup 2 > $OUT/ds$i.json 2> $OUT/ds$i.err
  T=$(find $OUT/ds$i -name "*kernel_trace.csv" | head -1)
  echo "== $S"; python3 tools/trace_streams.py $T | head -16
  rm -rf $OUT/ds$i
done > $OUT/dec_sched_streams.txt 2>&1
cat $OUT/dec_sched_streams.txt
