#!/bin/bash
# One SQ counter pass per library variant (same-box): VALU/SALU instruction counts and busy cycles of the timed trace kernel.
# usage (GPU box, repo root): tools/pmc_variants.sh "<lib1> <lib2> ..." [bench.py args]     ("-" = the default build)
libs=$1; shift
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
for lib in $libs; do
  tag=$(basename $lib .so); [ "$lib" = "-" ] && tag=default
  out=$root/gpurun_out/pmcvar/$tag
  rm -rf "$out"; mkdir -p "$out"
  if [ "$lib" != "-" ]; then export RT_AMD_LIB=$root/$lib; else unset RT_AMD_LIB; fi
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d "$out" -- python3 "$root/bench.py" --no-cpu --steps 3 --warmup 1 "$@" > "$out/log.txt" 2>&1 || { echo "$tag: failed"; continue; }
  python3 - "$out" "$tag" <<'P'
import csv, glob, sys, collections
s = collections.defaultdict(float); n = collections.defaultdict(set)
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "trace_sweep_kernel" not in k or "false, false, false" not in k.replace("(anonymous namespace)::", ""):
            continue
        s[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
print("%-16s" % sys.argv[2], " ".join("%s=%.4g" % (c.replace("SQ_", ""), s[c] / max(1, len(n[c]))) for c in sorted(s)))
P
done
