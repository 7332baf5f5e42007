#!/bin/bash
# Round-3 evidence of the training step (forward + backward + Adam under HIP-graph replay), headline net B = 4096:
# wall time per precision / flow family, and the rocprofv3 kernel summary of the planar and RNVP steps.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r03
cd $R
for prec in bf16x3 fp16x3f; do for flow in Planar RNVP; do
  echo "== PREC=$prec FLOW=$flow"; PREC=$prec FLOW=$flow python3 tools/train_step_time.py graph 2>&1 | grep -v amdgpu.ids
done; done
export TMPDIR=/tmp
for flow in Planar RNVP; do
  (cd /tmp && PREC=${TPREC:-fp16x3f} FLOW=$flow rocprofv3 --kernel-trace -d $R/gpurun_out/r03/prof_train_$flow -o t -- python3 $R/tools/train_step_time.py graph > /dev/null 2>&1)
  python3 tools/trace_summary.py $(ls gpurun_out/r03/prof_train_$flow/*.db gpurun_out/r03/prof_train_$flow/*/*.db 2>/dev/null | head -1) 106 40 > gpurun_out/r03/train_step_graph_${flow}_kernel_summary.txt 2>&1
  head -45 gpurun_out/r03/train_step_graph_${flow}_kernel_summary.txt | cut -c1-190
done
