import csv, sys, glob, os
fs = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(fs[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last frame: from the last LEVEL0 non-stats launch
idx = max(i for i, r in enumerate(rows) if "sorted_level_kernel<false, 2, true, false>" in r["Kernel_Name"] or "sorted_level_kernel<false, 1, true, false>" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-62s start %8.3f end %8.3f dur %7.3f" % (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:62], s / 1e6, e / 1e6, (e - s) / 1e6))
