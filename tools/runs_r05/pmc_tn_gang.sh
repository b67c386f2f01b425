# rocprofv3 counter passes over the 12 encoder layers' weight gradients: 128 x 128 grouped (12 launches) vs gang (one launch)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r05/pmc_tn_gang_tmp; mkdir -p $out; rm -f $out/summary.txt
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  TN_BENCH_CHECK=0 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $out/p$i -o p$i -- python3 tools/tn_gang_bench.py pmc > $out/p$i.log 2>&1
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $set" >> $out/summary.txt
  for c in $set; do
    python3 tools/pmc_summary.py $f $c gemm_tn_grouped >> $out/summary.txt 2>&1
    python3 tools/pmc_summary.py $f $c gemm_tn_gang_kernel >> $out/summary.txt 2>&1
  done
  find $out/p$i -name "*.csv" -delete
done
cat $out/summary.txt
