set -x
mkdir -p gpurun_out/r04
rm -f gpurun_out/r04/prio_ab.txt
for i in 1 2; do
  for v in "DAV_FUSION_PRIO=0" "DAV_FUSION_PRIO=-1" "DAV_FUSION_PRIO=1"; do
    env $v python bench.py --steps 40 --warmup 5 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$v', d['ms_per_step'], d['median_ms_per_step_device_events'], d['loss'])" >> gpurun_out/r04/prio_ab.txt
  done
done
cat gpurun_out/r04/prio_ab.txt
bash tools/collect_r04.sh quick
python tools/prof_summary.py gpurun_out/r04/r04_instep_kernel_stats.csv 23 40
