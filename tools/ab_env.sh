#!/bin/bash
# Same-box A/B of one build under different environments (A/B knobs), interleaved and repeated.
# usage (GPU box): bash tools/ab_env.sh "VAR=1" [repeats]     compares `VAR=1 bench.py` against plain `bench.py`
set -e
ENVV=$1; REP=${2:-2}
one() { timeout -k 10 200 python bench.py --no-cpu --steps 10 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); c=j['config']['rank0_ray_classes']; print('%8.3f ms (kernel %.3f) crc %s tests %.3g groups %.3g exact %.3g' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['config'].get('frame_crc'), c['sphere_tests_run'], c['group_tests_run'], c['exact_tests_run']))"; }
for cfg in "" "--spp 1" "--scene grid16k --spp 1" "--width 7680 --height 4320 --depth 8 --spp 1"; do
  echo "== bench.py $cfg"
  for i in $(seq $REP); do
    printf "%-28s " "$ENVV"; env $ENVV bash -c "$(declare -f one); one $cfg"
    printf "%-28s " "(default)"; one $cfg
  done
done
