export TMPDIR=/tmp
rocprofv3 --list-avail > gpurun_out/avail.txt 2>&1
for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  SPLIT=1 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc_$name -o p --output-format csv -- python3 tools/profile_gemm.py 4096 5 > gpurun_out/pmc_$name.log 2>&1 || echo "FAILED $set"
done
ls gpurun_out | head -30
