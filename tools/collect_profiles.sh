#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box into gpurun_out/profiles_rNN/ (copy the summaries into profiles/).
# Usage: bash tools/collect_profiles.sh r02
R=${1:-r02}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/profiles_$R
mkdir -p $OUT
# 1. default bench under --kernel-trace --stats
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 bench.py --no-cpu-baseline > $OUT/bench_profiled.json 2> $OUT/bench.err
cp $(find $OUT/bench -name "*kernel_stats.csv" | head -1) $OUT/${R}_bench_kernel_stats.csv
# 2. dominant-kernel replay only
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/roof -o roof -- python3 bench.py --roofline-only > $OUT/${R}_roofline_bench.json 2> $OUT/roof.err
cp $(find $OUT/roof -name "*kernel_stats.csv" | head -1) $OUT/${R}_roofline_kernel_stats.csv
# 3. HBM traffic of the dominant kernel: separate PMC passes (TCC counters), each bounded
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/pmc_$c -o p -- python3 bench.py --roofline-only > /dev/null 2> $OUT/pmc_$c.err
  f=$(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 tools/pmc_families.py $f > $OUT/${R}_pmc_$c.txt 2>&1
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
ls $OUT
