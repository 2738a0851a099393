#!/bin/bash
# dense_sweep in isolation: every lane far (RT_AMD_FAR_FRACTION=1e-9), 16k scene at 640x360; time and instruction counts
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo
RT_AMD_FAR_FRACTION=1e-9 timeout -k 10 120 python3 $R/bench.py --no-cpu --steps 3 --warmup 1 --scene grid16k --spp 1 --width 640 --height 360 | python3 -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); c=j['config']['rank0_ray_classes']; print('all-dense: %.3f ms kernel, rays %d, tests %.4g' % (j['roofline']['kernel_ms'], j['config']['rays_per_frame'], c['sphere_tests_run']))"
RT_AMD_FAR_FRACTION=1e-9 timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/dense_pmc -- python3 $R/bench.py --no-cpu --steps 2 --warmup 1 --scene grid16k --spp 1 --width 640 --height 360 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('/root/repo/gpurun_out/dense_pmc/**/*counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(float); n=collections.Counter()
for r in csv.DictReader(open(f)):
    if 'false, false, false, 6, false, 2' in r['Kernel_Name']:
        acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
for k,v in sorted(acc.items()): print(k, '%.4g per launch' % (v/max(1,n[k]/8) if False else v/ (n[k]/ (n[k]/max(1,len(set([1])))))), n[k])
PY
