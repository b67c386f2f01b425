#!/bin/bash
# In-situ kernel times of the step under two settings of one switch: tools/prof_ab.sh "ENV=a" "ENV=b" [bench args]
# (rocprofv3 --kernel-trace --stats of bench.py --no-roofline; per-step summary by kernel name)
A=$1; B=$2; shift; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for tag in A B; do
  if [ $tag = A ]; then S=$A; else S=$B; fi
  rm -rf gpurun_out/pab_$tag
  env $S timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pab_$tag -o p -- python3 bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 "$@" > gpurun_out/pab_$tag.json 2> gpurun_out/pab_$tag.err
  f=$(find gpurun_out/pab_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $S"; python3 tools/prof_summary.py $f 26 14
  find gpurun_out/pab_$tag -name "*kernel_trace.csv" -delete
done
