#!/bin/bash
# Regenerates the NT tile-configuration table for the bench configurations on this box (run on the MI355X):
#   for each configuration: dump the step's launch mix under the built-in rules, sweep every group, merge the winners.
# Output: gpurun_out/tuning/nt_gfx950.json (+ the sweep logs); copy it to deepavfusion_amd/tuning/ to ship it.
set -u
OUT=gpurun_out/tuning
mkdir -p $OUT
rm -f $OUT/nt_gfx950.json
for cfg in "$@"; do
  DAV_NT_TUNE=0 DAV_DUMP_MIX=$OUT/mix_$cfg.json timeout 400 python bench.py --config $cfg --no-cpu-baseline --no-roofline --steps 2 --warmup 1 > /dev/null 2> $OUT/dump_$cfg.err
  timeout 1500 python tools/mix_sweep.py $OUT/mix_$cfg.json --write $OUT/nt_gfx950.json > $OUT/sweep_$cfg.txt 2>&1
  tail -3 $OUT/sweep_$cfg.txt
done
