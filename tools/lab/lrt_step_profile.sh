#!/bin/bash
# Kernel list of the captured LRT training step (rocprofv3 --kernel-trace --stats over tools/lrt_train_step_time.py).
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r03f
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03f/prof_lrt -o b --output-format csv -- python3 $R/tools/lrt_train_step_time.py > $R/gpurun_out/r03f/lrt_prof.log 2>&1)
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/r03f/prof_lrt/**/b_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print("%-100s calls %6s avg %8.2f us  %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
grep "ms per" $R/gpurun_out/r03f/lrt_prof.log
