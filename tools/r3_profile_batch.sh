#!/bin/bash
# Round-3 measurement batch, final state (GPU box, repo root): bench lines, the PMC summary of the 16k scene through the sorted sweep
# (the headline kernel is the one profiles/r3/headline_pmc.csv and headline_instmix.csv already describe: same object code), the
# per-launch times of a sorted frame, band balance, the orbit.  Everything lands in gpurun_out/r3prof/; what is worth keeping is
# copied to profiles/r3/ by hand afterwards.  (The s_memtime region breakdown of the sorted sweep needs a profiling build and is
# run apart: tools/region_profile_sorted.sh.)
out=gpurun_out/r3prof
mkdir -p $out
timeout -k 10 300 python bench.py > $out/bench.json 2> $out/bench.err || exit 1
timeout -k 10 300 python bench.py --scene grid16k --spp 1 --no-cpu > $out/bench_grid16k.json 2>> $out/bench.err || exit 1
timeout -k 10 300 python bench.py --scene grid16k --spp 1 --no-cpu --sweep grouped > $out/bench_grid16k_grouped.json 2>> $out/bench.err || exit 1
timeout -k 10 300 python bench.py --scene grid16k --spp 1 --no-cpu --width 1920 --height 1080 > $out/bench_grid16k_1080p.json 2>> $out/bench.err || exit 1
timeout -k 10 300 python bench.py --mode fast --no-cpu > $out/bench_fast.json 2>> $out/bench.err || exit 1
timeout -k 10 300 python bench.py --width 7680 --height 4320 --depth 8 --spp 1 --no-cpu > $out/bench_config4.json 2>> $out/bench.err || exit 1
bash tools/pmc_profile.sh r3_grid16k_sorted --scene grid16k --spp 1 > $out/pmc_grid16k_sorted.log 2>&1 || exit 1
bash tools/kernel_trace.sh r3_grid16k_sorted --scene grid16k --spp 1 > $out/grid16k_sorted_kernels.txt 2>&1 || exit 1
python3 tools/sorted_timeline.py gpurun_out/kt_r3_grid16k_sorted > $out/grid16k_sorted_timeline.txt 2>&1 || exit 1
timeout -k 10 300 python -u tools/camera_sweep.py --frames 48 > $out/camera_sweep_grid16k.json 2>> $out/bench.err || exit 1
for g in 2 4 8; do timeout -k 10 300 python -u tools/band_balance.py --gpus $g; done > $out/band_balance.txt 2>&1
for g in 8; do timeout -k 10 300 python -u tools/band_balance.py --gpus $g --scene grid16k --spp 1; done >> $out/band_balance.txt 2>&1
cp gpurun_out/prof_r3_grid16k_sorted/summary.csv $out/grid16k_sorted_pmc.csv
ls -la $out
