#!/bin/bash
# A/B of two builds of the library on the headline forward, alternating processes on one box:
#   bash tools/ab_run.sh <base.so> [rounds]      (the tree's own csrc/liblbbnn_hip.so is the candidate)
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
BASE=$1; N=${2:-3}
mkdir -p gpurun_out/ab
for i in $(seq 1 $N); do
  echo "--- round $i: base";      LBBNN_LIB_PATH=$BASE PRECS=${PRECS:-fp16x3f} ROUNDS=5 python3 tools/precision_time.py
  echo "--- round $i: candidate"; PRECS=${PRECS:-fp16x3f} ROUNDS=5 python3 tools/precision_time.py
done
