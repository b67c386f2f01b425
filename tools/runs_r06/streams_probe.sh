#!/bin/bash
# which stream a layer waits for: kernel trace of the default step, per hardware queue (tools/trace_streams.py); base and ViT-L
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06; mkdir -p $OUT
for cfg in ${1:-base large}; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/st_$cfg -o st -- python3 bench.py --config $cfg --no-cpu-baseline --no-roofline --steps 6 --warmup 2 > $OUT/st_$cfg.json 2> $OUT/st_$cfg.err
  T=$(find $OUT/st_$cfg -name "*kernel_trace.csv" | head -1)
  head -1 $T > $OUT/trace_header_$cfg.txt
  python3 tools/trace_streams.py $T > $OUT/r06_streams_$cfg.txt 2>&1
  python3 - "$T" "$OUT/trace_last_step_$cfg.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ad = sorted(int(r['End_Timestamp']) for r in rows if 'adamw_flat' in r['Kernel_Name'])
keep = [r for r in rows if ad[-2] <= int(r['Start_Timestamp']) <= ad[-1]]
w = csv.DictWriter(open(sys.argv[2], 'w'), fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(keep)
PY
  rm -rf $OUT/st_$cfg
done
cat $OUT/r06_streams_base.txt; cat $OUT/r06_streams_large.txt
