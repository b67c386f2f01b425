mkdir -p gpurun_out/r04
timeout 1200 python tools/instep_gemm_bound.py --rounds 6 --reps 10 --modes base,baseB,cfg60,baseC,sub_blas,baseD > gpurun_out/r04/gemm_bound_noise.txt 2>&1
cat gpurun_out/r04/gemm_bound_noise.txt | grep -v JSON
