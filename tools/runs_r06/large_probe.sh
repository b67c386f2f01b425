#!/bin/bash
# ViT-L (BASELINE configs[3], B = 32): where the step goes (kernel trace -> phases / concurrency / families) and the launch-batch policy A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06; mkdir -p $OUT
for pol in "" "DAV_BATCH=1" "DAV_BATCH=0"; do
  echo "== $pol"; env $pol timeout 300 python bench.py --config large --no-cpu-baseline --no-roofline --steps 30 --warmup 5 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done > $OUT/r06_large_policy.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lg -o lg -- python3 bench.py --config large --no-cpu-baseline --no-roofline --steps 20 --warmup 3 > $OUT/r06_large_profiled.json 2> $OUT/lg.err
cp $(find $OUT/lg -name "*kernel_stats.csv" | head -1) $OUT/r06_instep_kernel_stats_large.csv
python3 tools/instep_families.py $OUT/r06_instep_kernel_stats_large.csv > $OUT/r06_instep_family_ms_large.txt 2>&1
T=$(find $OUT/lg -name "*kernel_trace.csv" | head -1)
MS=$(python3 -c "import json; print(json.load(open('$OUT/r06_large_profiled.json'))['ms_per_step'])")
( echo "# ViT-L B = 32: phases of the last replayed step (profiled step $MS ms)"; python3 tools/trace_phases.py $T $MS; python3 tools/trace_timeline.py $T $MS | head -4; python3 tools/trace_alone.py $T $MS | head -24 ) > $OUT/r06_timeline_large.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
cat $OUT/r06_large_policy.txt; cat $OUT/r06_timeline_large.txt
