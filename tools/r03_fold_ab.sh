R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; PRECS=fp16x3f python tools/precision_time.py; LBBNN_HEAD_FOLD=0 PRECS=fp16x3f python tools/precision_time.py
export TMPDIR=/tmp
(cd /tmp && PRECS=fp16x3f ROUNDS=2 rocprofv3 --kernel-trace -d $R/gpurun_out/r03/prof_fold -o t -- python3 $R/tools/precision_time.py > /dev/null 2>&1)
python3 tools/trace_summary.py $(ls gpurun_out/r03/prof_fold/*.db gpurun_out/r03/prof_fold/*/*.db 2>/dev/null | head -1) 1 8 | cut -c1-200
(cd /tmp && LBBNN_HEAD_FOLD=0 PRECS=fp16x3f ROUNDS=2 rocprofv3 --kernel-trace -d $R/gpurun_out/r03/prof_nofold -o t -- python3 $R/tools/precision_time.py > /dev/null 2>&1)
python3 tools/trace_summary.py $(ls gpurun_out/r03/prof_nofold/*.db gpurun_out/r03/prof_nofold/*/*.db 2>/dev/null | head -1) 1 8 | cut -c1-200
