set -e
one() { timeout -k 10 200 python bench.py --no-cpu --steps 6 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); print('%8.3f ms (kernel %.3f)' % (j['ms_per_step'], j['roofline']['kernel_ms']))"; }
for cfg in "--spp 1" "--scene grid16k --spp 4" "--scene grid16k --spp 1 --width 1920 --height 1080" "--width 1920 --height 1080"; do
  echo "== bench.py $cfg"
  for f in 0.15 0.25 0.5; do printf "far fraction %-6s " $f; RT_AMD_FAR_FRACTION=$f one $cfg; done
done
