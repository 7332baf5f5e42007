#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r03f
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03f/prof_base -o b --output-format csv -- python3 $R/tools/lab/base_step_profile.py > $R/gpurun_out/r03f/base_prof.log 2>&1)
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/r03f/prof_base/**/b_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
nat = sum(float(r["TotalDurationNs"]) for r in rows if "at::native" in r["Name"] or "rocclr" in r["Name"])
calls = sum(int(r["Calls"]) for r in rows); ncalls = sum(int(r["Calls"]) for r in rows if "at::native" in r["Name"] or "rocclr" in r["Name"])
print("kernel time %.1f us per step over 30 steps, %.0f launches per step; torch-native: %.1f us, %.0f launches" % (tot / 30e3, calls / 30, nat / 30e3, ncalls / 30))
for r in rows[:22]:
    print("%-110s calls %6s avg %8.2f us  %5.1f%%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
grep "ms" $R/gpurun_out/r03f/base_prof.log
