#!/bin/bash
# after `gpurun -- bash tools/collect_r05.sh quick` merged gpurun_out/r05/: copy what is judged into profiles/ and regenerate the traffic constants
# (run in the build container, from the repository root; then run bench.py once more on the GPU so that profiles/r05_bench.json carries them)
set -e
for f in r05_bench_base_as.json r05_bench_large.json r05_bench_video.json r05_bench_profiled.json r05_instep_family_ms.txt r05_instep_kernel_stats.csv r05_pmc_FETCH_SIZE.txt r05_pmc_WRITE_SIZE.txt r05_pmc_sq_counters_cfg3.txt r05_pmc_tn_gang.txt r05_roofline_bench.json r05_roofline_kernel_stats.csv r05_step_traffic.txt r05_tn_gang_bench.txt; do cp gpurun_out/r05/$f profiles/$f; done
for c in base_as large video; do cp gpurun_out/r05/instep_family_ms_$c.txt profiles/r05_instep_family_ms_$c.txt; done
cp gpurun_out/r05/step_traffic.json profiles/step_traffic.json
python tools/traffic_json.py profiles base_b64 r05
python -m pytest tests/test_cabi_and_host.py -q -k traffic_constants | tail -1
