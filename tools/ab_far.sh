#!/bin/bash
# A/B of RT_AMD_FAR_FRACTION (which lanes go through dense_sweep) on the bench workloads.
# usage (GPU box): bash tools/ab_far.sh "0 0.1 0.25 0.5" ["lib1 lib2"]     ("-" = the default build)
set -e
one() { timeout -k 10 200 python bench.py --no-cpu --steps 10 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); c=j['config']['rank0_ray_classes']; print('%8.3f ms (kernel %.3f) crc %s tests %.3g groups %.3g exact %.3g' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['config'].get('frame_crc'), c['sphere_tests_run'], c['group_tests_run'], c['exact_tests_run']))"; }
for cfg in "" "--scene grid16k --spp 1" "--width 7680 --height 4320 --depth 8 --spp 1"; do
  echo "== bench.py $cfg"
  for lib in ${2:--}; do
    for f in $1; do
      printf "%-32s far fraction %-6s " "$lib" "$f"
      if [ "$lib" = "-" ]; then RT_AMD_FAR_FRACTION=$f one $cfg; else RT_AMD_LIB=$PWD/$lib RT_AMD_FAR_FRACTION=$f one $cfg; fi
    done
  done
done
