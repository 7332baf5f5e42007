export TMPDIR=/tmp
for v in k1nostore k1nomath; do
  LBBNN_LIB_PATH=$PWD/tools/lab/liblbbnn_$v.so timeout -k 10 150 rocprofv3 --kernel-trace --stats -d gpurun_out/k1_$v -o f --output-format csv -- python3 bench.py --no-cpu-baseline --steps 50 > /dev/null 2>&1
  echo "== $v"; grep weight_pass_kernel gpurun_out/k1_$v/f_kernel_stats.csv | cut -d, -f2-5
done
