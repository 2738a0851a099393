#!/bin/bash
# the 16k scene at 4K: 1 spp twice, 4 spp once (time and frame checksum)
set -e
for a in "--spp 1" "--spp 1" "--spp 4"; do
  timeout -k 10 150 python bench.py --no-cpu --steps 8 --warmup 2 --scene grid16k $a 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); print('16k $a: %.3f ms crc %s' % (j['ms_per_step'], j['config']['frame_crc']))"
done
