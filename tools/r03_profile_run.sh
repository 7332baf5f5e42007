#!/bin/bash
# Round-3 evidence run (on the GPU box, from the repo root): the driver's bench command, the same command under
# rocprofv3 --kernel-trace --stats, and the PMC passes of the dominant GEMM (one counter set per run).
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r03
cd $R && python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03/bench_driver_cmd.json 2> gpurun_out/r03/bench_driver_cmd.err; echo "bench rc $?"
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03/prof_bench -o b --output-format csv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r03/bench_under_rocprof.json 2> $R/gpurun_out/r03/bench_under_rocprof.err); echo "rocprof rc $?"
cd $R && FMT=3 bash tools/pmc_gemm.sh > gpurun_out/r03/pmc_run.log 2>&1; echo "pmc rc $?"
python3 tools/pmc_summary.py gpurun_out/pmc_*/p_counter_collection.csv --match lrt_gemm_f16s > gpurun_out/r03/pmc_gemm_f16f.txt 2>&1
cat gpurun_out/r03/pmc_gemm_f16f.txt | cut -c1-250
