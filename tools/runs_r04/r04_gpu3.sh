set -x
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "fusion_tails or fusion_block_module or (baseline_config_shapes and base-4)" > gpurun_out/r04/tails_test.txt 2>&1
tail -25 gpurun_out/r04/tails_test.txt
for i in 1 2 3; do
  for v in 1 0; do
    DAV_FUSION_TAIL=$v python bench.py --steps 40 --warmup 5 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('DAV_FUSION_TAIL=$v', d['ms_per_step'], d['median_ms_per_step_device_events'], d['loss'])" >> gpurun_out/r04/tails_ab.txt
  done
done
cat gpurun_out/r04/tails_ab.txt
