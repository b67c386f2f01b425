for shape in "3136 768 768 5" "4032 768 3072 5" "2048 768 768 5" "6080 768 2304 8" "512 768 768 5"; do python tools/gemm_one.py $shape 2>/dev/null | tail -1; done
