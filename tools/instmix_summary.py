#!/usr/bin/env python3
"""Folds the passes of tools/pmc_instmix.sh: (1) calibration -- each counter divided by the known number of
wave-instructions of tools/instmix_calib's single-kind kernels; (2) the trace kernels' counters per launch, summed over
XCDs (the timed instantiation and the stats-enabled census launch are listed separately)."""
import csv
import glob
import os
import sys
from collections import defaultdict

CALIB_INSTS = 256 * 4 * 256 * 8


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


META = {}


def fold(root, sub, want):
    sums, launches = defaultdict(lambda: defaultdict(float)), defaultdict(set)
    for path in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if want not in k:
                continue
            sums[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
            META[k] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"),
                       r.get("Scratch_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))
    return {k: {c: v / max(1, len(launches[(k, c)])) for c, v in cs.items()} for k, cs in sums.items()}


def main(root):
    w = csv.writer(sys.stdout)
    w.writerow(["# note: rocprofv3 reports vgpr = 64 for kernels whose code-object notes say 128 -- trust raytracer_amd/kernel_resources.json "
                "(read from the notes); the scratch size is right, and is what bench.py matches a profile to a build by"])
    calib = {}
    for sub in ("calib_a", "calib_b"):
        for k, cs in fold(root, sub, "calib_").items():
            calib.setdefault(k, {}).update(cs)
    if calib:
        w.writerow(["# calibration: counter value / %d wave-instructions of the one kind each kernel issues (tools/instmix_calib.hip); "
                    "a few dozen set-up instructions per wave explain fractions below 0.01" % CALIB_INSTS])
        names = sorted({c for cs in calib.values() for c in cs})
        w.writerow(["kernel"] + [n.replace("SQ_INSTS_VALU_", "").replace("SQ_INSTS_", "") for n in names])
        for k in calib:
            w.writerow([k.replace("calib_", "")] + ["%.2f" % (calib[k].get(n, 0.0) / CALIB_INSTS) for n in names])
    mix = {}
    for sub in ("mix_a", "mix_b", "mix_c", "mix_d"):
        for k, cs in fold(root, sub, "trace_").items():
            mix.setdefault(k, {}).update(cs)
    for k in sorted(mix):
        w.writerow([])
        w.writerow(["# " + k])
        w.writerow(["# vgpr,agpr,sgpr,lds_block,scratch,grid,wg = " + ",".join(str(v) for v in META.get(k, ()))])
        w.writerow(["counter", "per_launch_summed_over_xcds"])
        for c in sorted(mix[k]):
            w.writerow([c, "%.0f" % mix[k][c]])


if __name__ == "__main__":
    main(sys.argv[1])
