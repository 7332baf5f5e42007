#!/bin/bash
# Every random-case tool of tools/ with fresh seeds (argument: first seed), one line of result per run.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
S=${1:-100}
run() { echo "== $*"; timeout -k 10 280 python3 "$@" 2>&1 | grep -v "amdgpu.ids\|Warn\|detach()\|errs, e32\|run_backward\|^ok " | tail -2 | cut -c1-400; }
run tools/net_train_fuzz.py $S 40
run tools/net_train_fuzz.py $((S+1)) 40
run tools/graph_step_fuzz.py $S 30
run tools/dp_step_fuzz.py $S 30
run tools/forward_replay_fuzz.py $S 40
run tools/forward_replay_fuzz.py $((S+1)) 40
run tools/ensemble_fuzz.py $S 40
run tools/base_vd_fuzz.py $S 60
run tools/layer_fuzz.py $S 60
run tools/net16_fuzz.py $S 40
run tools/gemm16_fuzz.py $S 150
