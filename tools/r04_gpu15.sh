mkdir -p gpurun_out/r04
rm -f gpurun_out/r04/placement_sweep.txt
for rep in 1 2; do
for sh in "" 1 3 17 64.5 257 1025.25 4099; do
  DAV_BENCH_SHIFT_MB=$sh python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('shift_MB=[$sh]', d['ms_per_step'], d['median_ms_per_step_device_events'], 'm75', d['secondary']['ms_per_step'])" >> gpurun_out/r04/placement_sweep.txt
done
done
cat gpurun_out/r04/placement_sweep.txt
