#!/bin/bash
# Usage (GPU box, repo root): tools/kernel_trace.sh <tag> [bench.py args...]   -> per-kernel stats of one bench.py run (rocprofv3 --kernel-trace --stats)
tag=$1; shift
root=$(pwd); out=$root/gpurun_out/kt_$tag; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$root/bench.py" --no-cpu --steps 3 --warmup 1 "$@" > "$out/trace.log" 2>&1 || exit 1
cd "$root"; f=$(find "$out" -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    print("%-90s calls %5s  total %10.3f ms  avg %9.3f us  %5s %%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
