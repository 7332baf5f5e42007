#!/bin/bash
# PMC passes over the GEMM-only driver (one counter set per run, as MI355X_MICROARCH.md prescribes).
# Usage (on the GPU box): [FMT=3] bash tools/pmc_gemm.sh  ->  gpurun_out/pmc_<set>/p_counter_collection.csv
# (FMT=2 / 3: the row-scaled fp16 kernel, 3 + 3 / 3 + 1 products; default: the bf16x3 kernel)
export TMPDIR=/tmp
export FMT=${FMT:-0}
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  echo "== $set"
  SPLIT=1 timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc_$name -o p --output-format csv -- python3 tools/profile_gemm.py 4096 5 > gpurun_out/pmc_$name.log 2>&1 || echo "FAILED $set (see gpurun_out/pmc_$name.log)"
done
