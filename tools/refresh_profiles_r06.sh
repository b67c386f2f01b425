#!/bin/bash
# after `gpurun -- bash tools/collect_r06.sh [quick]` merged gpurun_out/r06/: copy what is judged into profiles/ and regenerate the traffic constants
# (run in the build container, from the repository root; then run bench.py once more on the GPU so that profiles/r06_bench.json carries them)
set -e
export DAV_MEASURED_ON="$(cat gpurun_out/r06/measured_on.txt)"
for f in r06_bench_base_as.json r06_bench_large.json r06_bench_video.json r06_bench_profiled.json r06_instep_family_ms.txt r06_instep_kernel_stats.csv r06_timeline.txt r06_pmc_FETCH_SIZE.txt r06_pmc_WRITE_SIZE.txt r06_pmc_tn_gang.txt r06_roofline_bench.json r06_roofline_kernel_stats.csv r06_step_traffic.txt r06_tn_gang_bench.txt r06_ln_micro.txt r06_ln_fuse_ab.txt r06_ln_phases.txt r06_ln_fwd_only.txt r06_bench_dp1.json r06_dp1_tax.txt r06_timeline_large.txt r06_instep_family_ms_large.txt r06_large_policy.txt r06_dropout_selfcheck.txt; do cp gpurun_out/r06/$f profiles/$f; done
[ -f gpurun_out/r06/r06_gputest.log ] && cp gpurun_out/r06/r06_gputest.log profiles/r06_gputest.log
cp gpurun_out/r06/step_traffic.json profiles/step_traffic.json
cp gpurun_out/r06/wgrad_traffic.json profiles/wgrad_traffic.json
python tools/traffic_json.py profiles base_b64 r06
python -m pytest tests/test_cabi_and_host.py -q -k traffic_constants | tail -1
