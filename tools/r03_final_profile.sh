#!/bin/bash
# End-of-round evidence (on the GPU box, from the repo root): the driver's bench command, the default bench.py under
# rocprofv3 --kernel-trace --stats, the PMC traffic pass over K1.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r03f
cd $R && python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03f/bench_driver_cmd.json 2> gpurun_out/r03f/bench_driver_cmd.err; echo "bench rc $?"
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03f/prof_bench -o b --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r03f/bench_default_under_rocprof.json 2> $R/gpurun_out/r03f/bench_under_rocprof.err); echo "rocprof rc $?"
cp $(find $R/gpurun_out/r03f/prof_bench -name "b_kernel_stats.csv" | head -1) $R/gpurun_out/r03f/kernel_stats_bench_default.csv
head -12 $R/gpurun_out/r03f/kernel_stats_bench_default.csv | cut -c1-160
cd $R && bash tools/pmc_k1.sh > gpurun_out/r03f/pmc_k1.txt 2>&1; tail -8 gpurun_out/r03f/pmc_k1.txt | cut -c1-200
