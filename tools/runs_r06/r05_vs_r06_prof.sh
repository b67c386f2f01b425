#!/bin/bash
# in-situ kernel time by family, round-5 tree vs this tree (rocprofv3 --kernel-trace --stats of bench.py --no-roofline), same box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
for tag in r05 r06; do
  rm -rf $R/gpurun_out/pf_$tag; mkdir -p $R/gpurun_out/pf_$tag
  if [ $tag = r05 ]; then cd $R/tools/runs_r06/r05tree; else cd $R; fi
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pf_$tag -o p -- python3 bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 > $R/gpurun_out/pf_$tag.json 2> $R/gpurun_out/pf_$tag.err
  cd $R
  f=$(find gpurun_out/pf_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag"; python3 tools/instep_families.py $f | head -48
  T=$(find gpurun_out/pf_$tag -name "*kernel_trace.csv" | head -1)
  MS=$(python3 -c "import json; print(json.load(open('gpurun_out/pf_$tag.json'))['ms_per_step'])")
  python3 tools/trace_phases.py $T $MS
  find gpurun_out/pf_$tag -name "*kernel_trace.csv" -delete
done
