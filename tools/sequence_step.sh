cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/seq; mkdir -p $OUT
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python3 bench.py --no-cpu-baseline --no-roofline --steps 12 --warmup 3 > $OUT/bench.json 2> $OUT/prof.err
T=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
MS=$(python3 -c "import json; print(json.load(open('$OUT/bench.json'))['ms_per_step'])")
python3 tools/trace_sequence.py $T $MS > $OUT/sequence.txt 2>&1
find $OUT -name "*.csv" -delete
tail -2 $OUT/sequence.txt
