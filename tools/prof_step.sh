cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof2; mkdir -p $OUT
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 bench.py --no-cpu-baseline --no-roofline > $OUT/bench.json 2> $OUT/prof.err
cp $(find $OUT/bench -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
cat $OUT/bench.json | head -c 300
