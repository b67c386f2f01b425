mkdir -p gpurun_out/r04
timeout 1200 python tools/instep_gemm_bound.py --rounds 6 --reps 10 --modes base,cfg60,cfg60:fwd,cfg60:bkn,cfg60:enc,cfg60:dec,cfg60all:fwd > gpurun_out/r04/gemm_bound_cfg60b.txt 2>&1
cat gpurun_out/r04/gemm_bound_cfg60b.txt | grep -v JSON
