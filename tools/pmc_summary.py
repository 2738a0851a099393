#!/usr/bin/env python3
"""Folds the rocprofv3 outputs of tools/pmc_profile.sh into one small CSV: per-kernel time from the
kernel trace, and each PMC counter summed over XCDs, averaged per launch of the trace kernel (the
stats-enabled census launch is a different instantiation and is listed separately)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def main(root):
    w = csv.writer(sys.stdout)
    w.writerow(["# note: rocprofv3 reports vgpr = 64 for kernels whose code-object notes say 128 -- trust raytracer_amd/kernel_resources.json "
                "(read from the notes); the scratch size is right, and is what bench.py matches a profile to a build by"])
    for path in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
        w.writerow(["# kernel stats (rocprofv3 --kernel-trace --stats)"])
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct"])
        for r in csv.DictReader(open(path)):
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
    sums = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    meta = {}
    for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if "trace_" not in k and "sorted_" not in k:
                continue
            sums[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
            meta[k] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"),
                       r.get("Scratch_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))
    for k in sorted(sums):
        w.writerow([])
        w.writerow(["# " + k])
        w.writerow(["# vgpr,agpr,sgpr,lds_block,scratch,grid,wg = " + ",".join(str(v) for v in meta[k])])
        w.writerow(["counter", "per_launch_summed_over_xcds", "launches", "all_launches"])
        for c in sorted(sums[k]):
            n = len(launches[(k, c)])
            w.writerow([c, "%.0f" % (sums[k][c] / max(1, n)), n, "%.0f" % sums[k][c]])


if __name__ == "__main__":
    main(sys.argv[1])
