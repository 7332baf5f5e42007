mkdir -p gpurun_out/r02
timeout -k 10 200 python3 tools/ring_check.py 2>&1 | grep -v amdgpu | grep "RESULT\|DIFF" 
for r in 0 2; do echo "== ring $r"; LBBNN_GEMM_RING=$r timeout -k 10 200 python3 tools/gemm_ksweep.py 2>&1 | grep -v amdgpu | grep "fit\|784\|1200" || exit 1; done
echo "== ring 2 stamps"; LBBNN_GEMM_RING=2 LBBNN_LIB_PATH=tools/lab/liblbbnn_gstamps.so timeout -k 10 120 python3 tools/gemm_stamps.py 1200 1200 2>&1 | grep -v amdgpu | head -10
