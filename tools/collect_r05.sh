#!/bin/bash
# Round-5 evidence, collected on the GPU box into gpurun_out/r05/ (copy what is to be judged into profiles/):
#   bash tools/collect_r05.sh [quick]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05; mkdir -p $OUT
# 1. the GPU suite as the driver runs it
if [ "$1" != "quick" ]; then timeout 2400 python -m pytest tests -x -q -m gpu --durations=12 > $OUT/r05_gputest.log 2>&1; tail -3 $OUT/r05_gputest.log; fi
# 2. bench lines: BASELINE configs[1] (default), configs[2] shapes, configs[3] ViT-L, configs[4] video
timeout 900 python bench.py > $OUT/r05_bench.json 2> $OUT/bench.err
timeout 900 python bench.py --config base_as --no-cpu-baseline > $OUT/r05_bench_base_as.json 2> $OUT/bench_as.err
timeout 900 python bench.py --config large --no-cpu-baseline > $OUT/r05_bench_large.json 2> $OUT/bench_l.err
timeout 600 python tools/video_bench.py > $OUT/r05_bench_video.json 2> $OUT/bench_v.err
# 3. rocprofv3 summaries: the captured step in situ (graph replays; per-kernel averages = in-step durations) and the roofline replay
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 3 > $OUT/r05_bench_profiled.json 2> $OUT/prof.err
cp $(find $OUT/bench -name "*kernel_stats.csv" | head -1) $OUT/r05_instep_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/roof -o roof -- python3 bench.py --roofline-only > $OUT/r05_roofline_bench.json 2> $OUT/roof.err
cp $(find $OUT/roof -name "*kernel_stats.csv" | head -1) $OUT/r05_roofline_kernel_stats.csv
# 4. HBM traffic of the dominant kernel (separate PMC passes over the roofline replay) and of the whole step
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/pmc_$c -o p -- python3 bench.py --roofline-only > /dev/null 2> $OUT/pmc_$c.err
  f=$(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 tools/pmc_families.py $f > $OUT/r05_pmc_$c.txt 2>&1
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/step_$c -o p -- python3 bench.py --no-graph --no-roofline --no-cpu-baseline --steps 3 --warmup 1 > $OUT/step_$c.json 2> $OUT/step_$c.err
done
python3 tools/step_traffic.py $(find $OUT/step_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $OUT/step_WRITE_SIZE -name "*counter_collection.csv" | head -1) $OUT/step_traffic.json base_b64 > $OUT/r05_step_traffic.txt 2>&1
# 4b. the gang-scheduled weight-gradient launch against the 128 x 128 grouped kernel: L2 hit rate, fabric fetch, SQ counters
bash tools/runs_r05/pmc_tn_gang.sh > $OUT/r05_pmc_tn_gang.txt 2>&1
TN_BENCH_CHECK=0 timeout 300 python tools/tn_gang_bench.py dec enc all > $OUT/r05_tn_gang_bench.txt 2>&1
# 4c. in-step kernel time by family (the profiled bench of step 3) and for the other published configurations
python3 tools/instep_families.py $OUT/r05_instep_kernel_stats.csv > $OUT/r05_instep_family_ms.txt 2>&1
bash tools/runs_r05/family_profiles.sh > /dev/null 2>&1
# 5. SQ counters of the DEFAULT instantiation of the dominant GEMM (configuration 3 = <128,128,4,2,2>)
for cfg in 3; do
  i=0; : > $OUT/r05_pmc_sq_counters_cfg$cfg.txt
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU" "GRBM_GUI_ACTIVE TA_TA_BUSY_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $OUT/sq$cfg/p$i -o p$i -- python3 tools/gemm_one.py 11264 2304 768 $cfg > $OUT/sq$cfg.log 2>&1
    f=$(find $OUT/sq$cfg/p$i -name "*counter_collection.csv" | head -1)
    echo "== pass $i: $set   (tools/gemm_one.py 11264 2304 768 $cfg)" >> $OUT/r05_pmc_sq_counters_cfg$cfg.txt
    [ -n "$f" ] && python3 tools/pmc_families.py $f >> $OUT/r05_pmc_sq_counters_cfg$cfg.txt 2>&1
  done
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
ls $OUT
