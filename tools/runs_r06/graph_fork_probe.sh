#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06; mkdir -p $OUT
for v in plain join8 join2 ext; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/gf_$v -o gf -- python3 tools/runs_r06/graph_fork_probe.py $v > $OUT/gf_$v.log 2>&1
  T=$(find $OUT/gf_$v -name "*kernel_trace.csv" | head -1)
  [ -n "$T" ] && python3 tools/runs_r06/graph_fork_probe_report.py $T $v || tail -3 $OUT/gf_$v.log
  rm -rf $OUT/gf_$v
done > $OUT/r06_graph_fork_probe.txt 2>&1
cat $OUT/r06_graph_fork_probe.txt
