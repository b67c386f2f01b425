for shape in "3136 768 768 5" "4032 768 768 5" "4032 768 3072 5" "2048 768 768 5" "6080 768 2304 8" "4032 3072 768 3" "22528 512 2048 3" "22528 2048 512 44"; do
  for d in 0 1 2 4; do echo -n "DEBUG=$d  "; DAV_NT_DEBUG=$d python tools/gemm_one.py $shape 2>/dev/null | tail -1; done
done
