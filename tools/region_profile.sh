#!/bin/bash
# Where do the trace kernel's wave-cycles go?  Needs a profiling build of the grouped sweep:
#   cd raytracer_amd/csrc && hipcc <flags of the Makefile> -DRT_REGION_TIMERS -c rt_sweep_grouped.hip -o build_rt/rt_sweep_grouped.o   (and rt_sweep_grouped3.hip likewise)
#   and a library linked from it: raytracer_amd/librt_amd_regions.so
# usage (GPU box): bash tools/region_profile.sh [bench.py options] > gpurun_out/regions.txt
set -e
LIB=${RT_REGION_LIB:-$PWD/raytracer_amd/librt_amd_regions.so}
RT_AMD_LIB=$LIB RT_AMD_DEBUG=1 RT_AMD_REGION_DUMP=1 timeout -k 10 200 python bench.py --no-cpu --steps 1 --warmup 0 "$@" 2>&1 >/dev/null | grep "rt region cycles" | grep -v "cycles: 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0" | tail -1 | python3 -c '
import sys
names = ["ray generation", "primary: super-group bounds", "primary: group bounds", "primary: pair records (filter + exact tests)", "primary: planes, finish",
         "hit point, normal, material, reflect", "shadow-ray set-up", "segment: super-group bounds", "segment: group bounds",
         "segment: pair records (filter + exact tests)", "planes + lit decisions", "records, next segment", "shading (backward pass)", "write-out", "sweeps without beams, only the reflection class wide (bounds + pairs)", "other sweeps without beams, far lanes (bounds + pairs)"]
v = [int(x) for x in sys.stdin.read().split(":")[1].split()]
t = sum(v)
print("wave-cycles by region of the grouped kernel as timed (last launch) (s_memtime, summed over waves; %s)" % " ".join(sys.argv[1:]))
for n, c in zip(names, v):
    if c: print("%-48s %6.2f %%  %14d" % (n, 100.0 * c / t, c))
' "$@"
