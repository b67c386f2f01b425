# in-step kernel-family tables of the other published configurations (rocprofv3 kernel trace of the benches)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_as -o bench -- python3 bench.py --config base_as --no-cpu-baseline --no-roofline --steps 20 --warmup 3 > $OUT/bench_base_as_profiled.json 2> $OUT/p_as.err
python3 tools/instep_families.py $(find $OUT/p_as -name "*kernel_stats.csv" | head -1) > $OUT/instep_family_ms_base_as.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_v -o bench -- python3 tools/video_bench.py > $OUT/bench_video_profiled.json 2> $OUT/p_v.err
python3 tools/instep_families.py $(find $OUT/p_v -name "*kernel_stats.csv" | head -1) > $OUT/instep_family_ms_video.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_l -o bench -- python3 bench.py --config large --no-cpu-baseline --no-roofline --steps 20 --warmup 3 > $OUT/bench_large_profiled.json 2> $OUT/p_l.err
python3 tools/instep_families.py $(find $OUT/p_l -name "*kernel_stats.csv" | head -1) > $OUT/instep_family_ms_large.txt
find $OUT/p_as $OUT/p_v $OUT/p_l -name "*.csv" -delete
for f in base_as video large; do echo "== $f"; head -32 $OUT/instep_family_ms_$f.txt; done
