#!/bin/bash
# PMC passes over the weight pass alone (one counter set per run): bench.py's forward, counters filtered to the K1 kernels.
export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  for rows in 1 0; do
    LBBNN_K1_ROWS=$rows timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmck1_${set}_rows$rows -o p --output-format csv -- python3 bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 10 --warmup 2 > gpurun_out/pmck1_${set}_rows$rows.log 2>&1 || echo "FAILED $set"
  done
done
python3 tools/pmc_summary.py gpurun_out/pmck1_*/p_counter_collection.csv --match weight_
