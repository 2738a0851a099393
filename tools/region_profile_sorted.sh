#!/bin/bash
# Where do the sorted sweep's wave-cycles go (all launches of a frame together)?  Needs a profiling build:
#   SORTED=1 tools/build_variant.sh regions -DRT_REGION_TIMERS        -> raytracer_amd/librt_amd_regions.so
# usage (GPU box): bash tools/region_profile_sorted.sh [bench.py options] > gpurun_out/regions_sorted.txt
set -e
LIB=${RT_REGION_LIB:-$PWD/raytracer_amd/librt_amd_regions.so}
RT_AMD_LIB=$LIB RT_AMD_DEBUG=1 RT_AMD_REGION_DUMP=1 timeout -k 10 200 python bench.py --no-cpu --steps 1 --warmup 0 "$@" 2>&1 >/dev/null | grep "rt region cycles" | grep -v "cycles: 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0" | tail -1 | python3 -c '
import sys
names = ["entries, stats, waiting at the end", "beam sweeps of one ray (primary, reflection): super-group bounds", "... group bounds", "... pair records (filter + exact tests)", "-",
         "hit point, normal, material, reflect, far hand-over", "ray set-up", "beam sweeps of several rays (shadow): super-group bounds", "... group bounds",
         "... pair records (filter + exact tests)", "planes + lit decisions", "records, slots, cells, queue entries", "-", "sweeps of several rays without beams", "sweeps of one ray without beams (nearest-first, culled)", "far segments and lanes (bounds + dense sweep)"]
v = [int(x) for x in sys.stdin.read().split(":")[1].split()]
t = sum(v)
print("wave-cycles by region of the sorted sweep as timed, all launches of the last frame (s_memtime, summed over waves; %s)" % " ".join(sys.argv[1:]))
for n, c in zip(names, v):
    if c: print("%-72s %6.2f %%  %14d" % (n, 100.0 * c / t, c))
' "$@"
