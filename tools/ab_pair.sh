#!/bin/bash
# Same-box A/B of two builds of the library: RT_AMD_LIB=<variant> against the default, interleaved.
# usage: bash tools/ab_pair.sh raytracer_amd/librt_amd_base.so [bench.py options]
set -e
VAR=$1; shift
one() { timeout -k 10 200 python bench.py --no-cpu --steps 20 --warmup 3 "$@" | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); c=j['config']['rank0_ray_classes']; print('%8.3f ms crc %s  tests %.3g groups %.3g exact %.3g' % (j['ms_per_step'], j['config'].get('frame_crc'), c['sphere_tests_run'], c['group_tests_run'], c['exact_tests_run']))"; }
for i in 1 2 3; do
  echo -n "variant "; RT_AMD_LIB=$PWD/$VAR one "$@"
  echo -n "default "; one "$@"
done
