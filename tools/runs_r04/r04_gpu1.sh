set -x
mkdir -p gpurun_out/r04
cd $GRAFT_REPO_ROOT
python bench.py --steps 30 --warmup 5 > gpurun_out/r04/bench_base0.json 2> gpurun_out/r04/bench_base0.err
timeout 900 python tools/instep_gemm_bound.py --rounds 6 --reps 10 > gpurun_out/r04/gemm_bound_streams.txt 2>&1
DAV_STREAMS=0 DAV_BATCH=0 timeout 900 python tools/instep_gemm_bound.py --rounds 5 --reps 10 > gpurun_out/r04/gemm_bound_serial.txt 2>&1
DAV_STOCK_OUT=gpurun_out/r04/torch_rocm_step.json timeout 600 python tools/torch_rocm_step.py --steps 10 --warmup 3 > gpurun_out/r04/stock1.txt 2>&1
DAV_STOCK_OUT=gpurun_out/r04/torch_rocm_step.json timeout 600 python tools/torch_rocm_step.py --steps 10 --warmup 3 --no-sdpa > gpurun_out/r04/stock2.txt 2>&1
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu --durations=15 -k "baseline_config or full_size" > gpurun_out/r04/parity_new.txt 2>&1
tail -5 gpurun_out/r04/*.txt
