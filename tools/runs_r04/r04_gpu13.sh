mkdir -p gpurun_out/r04
python - <<'PY' > gpurun_out/r04/attn_fused_check.txt 2>&1
import sys; sys.path.insert(0, 'tests')
import gpu_selfcheck as sc
sc.attention()
bad = [r for r in sc.RESULTS if not r[3]]
print(len(sc.RESULTS), 'checks', len(bad), 'bad', bad[:8])
PY
tail -4 gpurun_out/r04/attn_fused_check.txt
rm -f gpurun_out/r04/attn_fused_ab.txt
for i in 1 2; do
  for v in "DAV_ATTN_FUSED_BWD=1" "DAV_ATTN_FUSED_BWD=0"; do
    env $v python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); a=[(e['Nq'],e['fwd_us'],e['bwd_us']) for e in d['roofline_attention']['shapes'] if e['dqk']==32]; print('$v', d['ms_per_step'], d['median_ms_per_step_device_events'], d['loss'], a, 'm75', d['secondary']['ms_per_step'])" >> gpurun_out/r04/attn_fused_ab.txt
  done
done
cat gpurun_out/r04/attn_fused_ab.txt
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "baseline_config_shapes and (base-4 or base_m75)" 2>&1 | tail -3
