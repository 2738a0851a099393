#!/bin/bash
# Short A/B table: the headline workload and the other configurations, ms per frame each.
# usage (on the GPU box): bash tools/ab_table.sh > gpurun_out/ab.log
set -e
run() { timeout -k 10 200 python bench.py --no-cpu --steps 10 --warmup 2 "$@" | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); print('%-50s %8.3f ms %9.0f Mrays/s crc %s' % (' '.join(sys.argv[1:]), j['ms_per_step'], j['value'], j['config'].get('frame_crc')))" "$@"; }
run
run --spp 1
run --spp 16
run --width 7680 --height 4320 --depth 8 --spp 1
run --scene grid16k --spp 1
run --scene default
