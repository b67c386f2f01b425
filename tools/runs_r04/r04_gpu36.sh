#!/bin/bash
# final-tree sanity over the other configurations and the data-parallel launch form
mkdir -p gpurun_out
out=gpurun_out/r04_final_sanity.txt; : > $out
p() { python -c 'import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j.get("value"), j.get("ms_per_step"), j.get("loss"))'; }
echo -n "base      " >> $out; timeout 300 python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline 2>/dev/null | p >> $out
echo -n "base_as   " >> $out; timeout 300 python bench.py --config base_as --steps 20 --warmup 5 --no-roofline --no-cpu-baseline 2>/dev/null | p >> $out
echo -n "large     " >> $out; timeout 300 python bench.py --config large --steps 20 --warmup 5 --no-roofline --no-cpu-baseline 2>/dev/null | p >> $out
echo -n "video     " >> $out; timeout 300 python tools/video_bench.py 2>/dev/null | p >> $out
echo -n "torchrun 1 rank, forced DP " >> $out; DAV_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | p >> $out
cat $out
