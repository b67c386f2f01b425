set -x
mkdir -p gpurun_out/r04
timeout 900 python tools/instep_gemm_bound.py --rounds 6 --reps 10 --modes base,sub_blas,cfg60,cfg60all,cfg44,cfg46 > gpurun_out/r04/gemm_bound_cfg60.txt 2>&1
cat gpurun_out/r04/gemm_bound_cfg60.txt | grep -v JSON
