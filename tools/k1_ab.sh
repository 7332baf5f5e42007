#!/bin/bash
# A/B of the weight pass on the GPU box: rocprofv3 kernel stats of bench.py for (rows kernel + in-kernel flows) |
# (rows kernel, K3 launch) | (round-1 kernel, K3 launch).  Usage: bash tools/k1_ab.sh   (outputs under gpurun_out/r02/k1ab_*)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "inflow:1:1" "rows:0:1" "old:0:0"; do
  IFS=: read name inflow rows <<< "$cfg"
  export LBBNN_K1_INFLOW=$inflow LBBNN_K1_ROWS=$rows
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02/k1ab_$name -o k1 -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 100 > $R/gpurun_out/r02/k1ab_$name.log 2>&1
  head -6 $R/gpurun_out/r02/k1ab_$name/k1_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
done
