#!/bin/bash
# One GPU call: phase stamps of the GEMM, A/B of lab variants against the product build, SQ counter passes.
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
OUT=gpurun_out/r02/gemm_lab.txt
: > $OUT
echo "== stamps 1200x1200" >> $OUT
LBBNN_LIB_PATH=tools/lab/liblbbnn_gstamps.so timeout -k 10 120 python3 tools/gemm_stamps.py 1200 1200 >> $OUT 2>&1 || exit 1
echo "== stamps 784x1200" >> $OUT
LBBNN_LIB_PATH=tools/lab/liblbbnn_gstamps.so timeout -k 10 120 python3 tools/gemm_stamps.py 784 1200 >> $OUT 2>&1 || exit 1
for rep in 1 2; do
  for v in product ${LAB_VARIANTS:-noslp}; do
    echo "== $v (rep $rep)" >> $OUT
    if [ $v = product ]; then
      SPLIT=1 timeout -k 10 120 python3 tools/profile_gemm.py 4096 200 >> $OUT 2>&1 || exit 1
    else
      LBBNN_LIB_PATH=tools/lab/liblbbnn_$v.so SPLIT=1 timeout -k 10 120 python3 tools/profile_gemm.py 4096 200 >> $OUT 2>&1 || exit 1
    fi
  done
done
if [ -z "$NO_PMC" ]; then
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  echo "== pmc $set" >> $OUT
  SPLIT=1 timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/r02/pmc_sq_$name -o p --output-format csv -- python3 tools/profile_gemm.py 4096 5 > gpurun_out/r02/pmc_sq_$name.log 2>&1 || echo "FAILED $set" >> $OUT
done
python3 tools/pmc_summary.py gpurun_out/r02/pmc_sq_*/*/p_counter_collection.csv gpurun_out/r02/pmc_sq_*/p_counter_collection.csv >> $OUT 2>&1
fi
grep -v amdgpu $OUT | tail -150
