#!/bin/bash
# Same-box A/B of (environment, library) pairs, interleaved and repeated.
# usage (GPU box): bash tools/ab_matrix.sh <repeats> "<bench args>" "label|ENV=.. ENV2=..|lib" ...      (lib "-" = the default build, env "-" = none)
REP=$1; ARGS=$2; shift 2
one() { timeout -k 10 200 python bench.py --no-cpu --steps 10 --warmup 2 $ARGS 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); c=j['config']['rank0_ray_classes']; print('%8.3f ms (kernel %.3f) crc %s tests %.3g groups %.3g exact %.3g' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['config'].get('frame_crc'), c['sphere_tests_run'], c['group_tests_run'], c['exact_tests_run']))"; }
for i in $(seq $REP); do
  for spec in "$@"; do
    IFS='|' read -r label envs lib <<< "$spec"
    printf "%-28s " "$label"
    ( [ "$envs" != "-" ] && export $envs; [ "$lib" != "-" ] && export RT_AMD_LIB=$PWD/$lib; one )
  done
done
