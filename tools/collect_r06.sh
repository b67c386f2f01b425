#!/bin/bash
# Round-6 evidence, collected on the GPU box into gpurun_out/r06/ (then: bash tools/refresh_profiles_r06.sh in the build container):
#   bash tools/collect_r06.sh [quick]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06; mkdir -p $OUT
export DAV_MEASURED_ON="$(date +%F), $(python3 -c 'import torch; print(torch.cuda.get_device_name(0))' 2>/dev/null) (gfx950), host $(hostname | cut -c1-12)"
echo "$DAV_MEASURED_ON" > $OUT/measured_on.txt
# 1. the GPU suite as the driver runs it
if [ "$1" != "quick" ]; then timeout 2400 python -m pytest tests -x -q -m gpu --durations=12 > $OUT/r06_gputest.log 2>&1; tail -3 $OUT/r06_gputest.log; fi
# 2. bench lines: BASELINE configs[1] (default), configs[2] shapes, configs[3] ViT-L, configs[4] video
timeout 900 python bench.py > $OUT/r06_bench.json 2> $OUT/bench.err
timeout 900 python bench.py --config base_as --no-cpu-baseline > $OUT/r06_bench_base_as.json 2> $OUT/bench_as.err
timeout 900 python bench.py --config large --no-cpu-baseline > $OUT/r06_bench_large.json 2> $OUT/bench_l.err
timeout 600 python tools/video_bench.py > $OUT/r06_bench_video.json 2> $OUT/bench_v.err
# 3. rocprofv3 summaries: the captured step in situ and the roofline replay
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 3 > $OUT/r06_bench_profiled.json 2> $OUT/prof.err
cp $(find $OUT/bench -name "*kernel_stats.csv" | head -1) $OUT/r06_instep_kernel_stats.csv
python3 tools/instep_families.py $OUT/r06_instep_kernel_stats.csv > $OUT/r06_instep_family_ms.txt 2>&1
T=$(find $OUT/bench -name "*kernel_trace.csv" | head -1)
MS=$(python3 -c "import json; print(json.load(open('$OUT/r06_bench_profiled.json'))['ms_per_step'])")
( echo "# phases of the last replayed step (rocprofv3 --kernel-trace of bench.py --no-roofline --steps 20; profiled step $MS ms)"; python3 tools/trace_phases.py $T $MS; python3 tools/trace_timeline.py $T $MS | head -4; python3 tools/trace_alone.py $T $MS | head -16 ) > $OUT/r06_timeline.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/roof -o roof -- python3 bench.py --roofline-only > $OUT/r06_roofline_bench.json 2> $OUT/roof.err
cp $(find $OUT/roof -name "*kernel_stats.csv" | head -1) $OUT/r06_roofline_kernel_stats.csv
# 4. HBM traffic of the dominant kernel (separate PMC passes over the roofline replay) and of the whole step
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/pmc_$c -o p -- python3 bench.py --roofline-only > /dev/null 2> $OUT/pmc_$c.err
  f=$(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 tools/pmc_families.py $f > $OUT/r06_pmc_$c.txt 2>&1
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/step_$c -o p -- python3 bench.py --no-graph --no-roofline --no-cpu-baseline --steps 3 --warmup 1 > $OUT/step_$c.json 2> $OUT/step_$c.err
done
python3 tools/step_traffic.py $(find $OUT/step_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $OUT/step_WRITE_SIZE -name "*counter_collection.csv" | head -1) $OUT/step_traffic.json base_b64 "profiles/r06_step_traffic.txt (tools/collect_r06.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py --no-graph)" > $OUT/r06_step_traffic.txt 2>&1
# 4b. the step's two gang weight-gradient launches: fabric fetch / write / L2 hit rate per launch -> wgrad_traffic.json
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  TN_BENCH_CHECK=0 timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $OUT/wg$i -o p -- python3 tools/tn_gang_bench.py pmc_step > $OUT/wg$i.log 2>&1
done
cp profiles/wgrad_traffic.json $OUT/wgrad_traffic.json 2>/dev/null
python3 tools/wgrad_traffic.py $(find $OUT/wg1 -name "*counter_collection.csv" | head -1) $(find $OUT/wg2 -name "*counter_collection.csv" | head -1) $(find $OUT/wg3 -name "*counter_collection.csv" | head -1) $OUT/wg1.log base_b64 "profiles/r06_pmc_tn_gang.txt (tools/collect_r06.sh: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum over tools/tn_gang_bench.py pmc_step)" > $OUT/r06_pmc_tn_gang.txt 2>&1
cp profiles/wgrad_traffic.json $OUT/wgrad_traffic.json
TN_BENCH_CHECK=0 timeout 300 python tools/tn_gang_bench.py dec enc all > $OUT/r06_tn_gang_bench.txt 2>&1
# 5. LayerNorm folding: micro benches + same-box A/B of the step + phases
( python3 tools/ln_gemm_bench.py; python3 tools/ln_twin_bench.py ) > $OUT/r06_ln_micro.txt 2>&1
bash tools/ab_env2.sh "DAV_LN_FUSE=0" "DAV_LN_FUSE=1" > $OUT/r06_ln_fuse_ab.txt 2>&1
bash tools/runs_r06/phases_ab.sh DAV_LN_FUSE=0 DAV_LN_FUSE=1 > $OUT/r06_ln_phases.txt 2>&1
python3 tools/runs_r06/fwd_only.py > $OUT/r06_ln_fwd_only.txt 2>&1
# 6. single-rank cost of the data-parallel machinery
DAV_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 40 > $OUT/r06_bench_dp1.json 2> $OUT/dp1.err
bash tools/runs_r06/dp1_tax.sh > $OUT/r06_dp1_tax.txt 2>&1
# 7. ViT-L (configs[3]): phases / concurrency / families of its step, launch-batch policy A/B
bash tools/runs_r06/large_probe.sh > /dev/null 2>&1
# 8. dropout kernels (ABI 9) against torch with the same masks
timeout 300 python tests/gpu_selfcheck.py dropout > $OUT/r06_dropout_selfcheck.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
ls $OUT
