#!/usr/bin/env python3
"""Per-kernel, per-launch averages of the counters in rocprofv3 p_counter_collection.csv files.
Usage: python tools/pmc_summary.py gpurun_out/pmc_*/p_counter_collection.csv [--match substr]"""
import collections, csv, sys
match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else "lrt_gemm"
files = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] != "--match"]
for f in files:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if match not in k:
            continue
        key = (k[:70], r["Grid_Size"])
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[key].add(r["Dispatch_Id"])
    for key, v in agg.items():
        n = len(disp[key])
        print(f.split("/")[-2], key[0], "grid", key[1], "launches", n, {c: x / n for c, x in v.items()})
