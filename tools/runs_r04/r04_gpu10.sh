mkdir -p gpurun_out/r04
rm -f gpurun_out/r04/nt256_plain_ab.txt
for i in 1 2 3; do
  for v in "DAV_NT256_PLAIN=1" "DAV_NT256_PLAIN=0"; do
    env $v python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$v', d['ms_per_step'], d['median_ms_per_step_device_events'], d['loss'], 'roofline', r['frac'], r['launches_by_config'], 'in_step', r['in_step']['frac'], 'm75', d['secondary']['ms_per_step'])" >> gpurun_out/r04/nt256_plain_ab.txt
  done
done
cat gpurun_out/r04/nt256_plain_ab.txt
for c in base_as large base_m75; do for v in "DAV_NT256_PLAIN=1" "DAV_NT256_PLAIN=0"; do env $v python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$c $v', d['ms_per_step'], d['median_ms_per_step_device_events'])" >> gpurun_out/r04/nt256_plain_ab.txt; done; done
tail -4 gpurun_out/r04/nt256_plain_ab.txt
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "baseline_config_shapes and (base-64 or base-4 or large-2)" 2>&1 | tail -3
