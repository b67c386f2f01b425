mkdir -p gpurun_out/r04
DAV_TN_TILE=257 python - <<'PY' > gpurun_out/r04/tn256x128_check.txt 2>&1
import sys; sys.path.insert(0, 'tests')
import gpu_selfcheck as sc
sc.gemm_tn()
bad = [r for r in sc.RESULTS if not r[3]]
print(len(sc.RESULTS), 'checks', len(bad), 'bad', bad[:5])
PY
tail -3 gpurun_out/r04/tn256x128_check.txt
for v in 128 257; do echo "== DAV_TN_TILE=$v"; DAV_TN_TILE=$v python tools/tn_group_bench.py 2>&1 | tail -6; done > gpurun_out/r04/tn256x128_bench.txt
cat gpurun_out/r04/tn256x128_bench.txt
rm -f gpurun_out/r04/tn_tile_ab.txt
for i in 1 2; do
  for v in "DAV_TN_TILE=257" "DAV_TN_TILE=128"; do
    env $v python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$v', d['ms_per_step'], d['median_ms_per_step_device_events'], d['loss'], 'wgrad', d['roofline_wgrad']['frac'], d['roofline_wgrad']['avg_launch_us'], 'm75', d['secondary']['ms_per_step'])" >> gpurun_out/r04/tn_tile_ab.txt
  done
done
cat gpurun_out/r04/tn_tile_ab.txt
