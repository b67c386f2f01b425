#!/bin/bash
# the driver's multi-GPU launch form on the one GPU there is: torchrun, 1 rank, with and without the forced data-parallel machinery
mkdir -p gpurun_out
out=gpurun_out/r04_torchrun_form.txt; : > $out
for fd in 0 1; do
  echo "== DAV_FORCE_DIST=$fd" >> $out
  DAV_FORCE_DIST=$fd timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2953$fd bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2> gpurun_out/r04_torchrun_$fd.err | tail -1 | cut -c1-900 >> $out
  echo "rc=$?" >> $out
done
cat $out
