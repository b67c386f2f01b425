set -x
mkdir -p gpurun_out/r04
python tools/fusion_tail_bench.py > gpurun_out/r04/tail_bench.txt 2>&1
cat gpurun_out/r04/tail_bench.txt
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "fusion_tails or (baseline_config_shapes and base-4)" > gpurun_out/r04/tails_test2.txt 2>&1
tail -5 gpurun_out/r04/tails_test2.txt
rm -f gpurun_out/r04/tails_ab2.txt
for i in 1 2; do
  for v in "DAV_FUSION_TAIL=1" "DAV_FUSION_TAIL=0" "DAV_FUSION_TAIL=1 DAV_FUSION_PRIO=-1"; do
    env $v python bench.py --steps 40 --warmup 5 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$v', d['ms_per_step'], d['median_ms_per_step_device_events'], d['loss'])" >> gpurun_out/r04/tails_ab2.txt
  done
done
cat gpurun_out/r04/tails_ab2.txt
