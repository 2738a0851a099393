#!/bin/bash
# Same-box A/B of several builds of the library on several workloads, interleaved and repeated.
# usage (GPU box): bash tools/ab_libs.sh "<lib1> <lib2> ..." [repeats]   (lib paths relative to the repo; "-" = the default build)
set -e
LIBS=$1; REP=${2:-2}
one() { timeout -k 10 200 python bench.py --no-cpu --steps 10 --warmup 2 "$@" | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); c=j['config']['rank0_ray_classes']; print('%8.3f ms (kernel %.3f) crc %s tests %.3g groups %.3g exact %.3g' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['config'].get('frame_crc'), c['sphere_tests_run'], c['group_tests_run'], c['exact_tests_run']))"; }
for cfg in ${AB_CFGS:-"" "--mode fast" "--scene grid16k --spp 1" "--width 7680 --height 4320 --depth 8 --spp 1"}; do
  echo "== bench.py $cfg"
  for i in $(seq $REP); do
    for lib in $LIBS; do
      printf "%-40s " "$lib"
      if [ "$lib" = "-" ]; then one $cfg; else RT_AMD_LIB=$PWD/$lib one $cfg; fi
    done
  done
done
