#!/bin/bash
# final-tree bench line (default flags) + the serial schedule for the record (sum of the kernels' in-sequence times)
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err
tail -c 1500 gpurun_out/r04_bench_final.json
for s in "DAV_STREAMS=0 DAV_BATCH=0" "DAV_X=0"; do
  echo "== $s" >> gpurun_out/r04_serial_schedule.txt
  env $s timeout 400 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["ms_per_step"], j["value"])' >> gpurun_out/r04_serial_schedule.txt
done
cat gpurun_out/r04_serial_schedule.txt
