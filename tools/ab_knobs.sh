#!/bin/bash
# A/B of environment knobs of the library on the bench workloads.  usage (GPU box): bash tools/ab_knobs.sh "VAR=a VAR=b 'VAR=c OTHER=d'" ...
set -e
one() { timeout -k 10 200 python bench.py --no-cpu --steps 10 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); c=j['config']['rank0_ray_classes']; print('%8.3f ms (kernel %.3f) crc %s tests %.3g groups %.3g exact %.3g' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['config'].get('frame_crc'), c['sphere_tests_run'], c['group_tests_run'], c['exact_tests_run']))"; }
for cfg in "" "--scene grid16k --spp 1"; do
  echo "== bench.py $cfg"
  for kv in "$@"; do
    printf "%-44s " "$kv"; env $kv bash -c "$(declare -f one); one $cfg"
  done
done
