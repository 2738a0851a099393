#!/bin/bash
# the flat sweep (north_star's kernel) on the headline frame and on two smaller scenes.  usage (GPU box): bash tools/ab_flat.sh "<lib> ..." ("-" = default build)
set -e
one() { timeout -k 10 200 python bench.py --no-cpu --steps 4 --warmup 1 "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); f=j.get('roofline_flat') or j['roofline']; print('%8.3f ms  frac %.4f  crc %s' % (f['kernel_ms'], f['frac'], j['config'].get('frame_crc')))"; }
for cfg in "" "--scene default --spp 4" "--scene grid16k --spp 1 --width 960 --height 540"; do
  echo "== bench.py --sweep flat $cfg"
  for lib in $1; do
    printf "%-40s " "$lib"
    if [ "$lib" = "-" ]; then one --sweep flat $cfg; else RT_AMD_LIB=$PWD/$lib one --sweep flat $cfg; fi
  done
done
