#!/bin/bash
# Kernel durations of the headline forward with the first layer on fp32 x (default) and on planes formatted by extra
# workgroups of the flow launch (LBBNN_F16_FIRST=planes), under rocprofv3 --kernel-trace --stats.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r03f
for mode in f32 planes; do
  (cd /tmp && LBBNN_F16_FIRST=$mode timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03f/prof_first_$mode -o b --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-reduced --no-kernel-events --no-train --steps 200 --warmup 20 > $R/gpurun_out/r03f/first_$mode.json 2> $R/gpurun_out/r03f/first_$mode.err)
  echo "== LBBNN_F16_FIRST=$mode"
  python3 - <<PY
import csv, glob, json
f = glob.glob("$R/gpurun_out/r03f/prof_first_$mode/**/b_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 1.0:
        print("%-110s calls %5s avg %8.2f us" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3))
d = json.loads(open("$R/gpurun_out/r03f/first_$mode.json").read().strip().split("\n")[-1])
print("bench under rocprof: %.4f ms per step" % d["ms_per_step"])
PY
done
