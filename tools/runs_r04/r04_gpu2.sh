set -x
mkdir -p gpurun_out/r04
timeout 900 python tools/instep_knockout.py --rounds 6 --reps 10 > gpurun_out/r04/knockout_streams.txt 2>&1
python bench.py --steps 30 --warmup 5 > gpurun_out/r04/bench_base1.json 2> gpurun_out/r04/bench_base1.err
timeout 900 python -m pytest tests/test_dp_rccl_gpu.py tests/test_audio_frontend_gpu.py -x -q -m gpu --durations=8 > gpurun_out/r04/dp_tests.txt 2>&1
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04/blas_prof -o blas -- python3 $GRAFT_REPO_ROOT/tools/blas_ref.py > $GRAFT_REPO_ROOT/gpurun_out/r04/blas_ref.txt 2>&1
cd $GRAFT_REPO_ROOT
tail -12 gpurun_out/r04/knockout_streams.txt gpurun_out/r04/dp_tests.txt gpurun_out/r04/blas_ref.txt
