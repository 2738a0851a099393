#!/bin/bash
# Round-3 measurement batch (GPU box, repo root): PMC summaries, instruction mix, band balance, bench lines.
# (No s_memtime region breakdown this round: the profiling build of the large-scene kernel left a GPU faulted, see DESIGN.md section 4;
# tools/region_profile.sh is kept for the record but nothing runs it.)
# Everything lands in gpurun_out/r3prof/; the summaries worth keeping are copied to profiles/r3/ by hand afterwards.
out=gpurun_out/r3prof
mkdir -p $out
timeout -k 10 300 python bench.py > $out/bench.json 2> $out/bench.err || exit 1
timeout -k 10 300 python bench.py --scene grid16k --spp 1 --no-cpu > $out/bench_grid16k.json 2>> $out/bench.err || exit 1
timeout -k 10 300 python bench.py --mode fast --no-cpu > $out/bench_fast.json 2>> $out/bench.err || exit 1
bash tools/pmc_profile.sh r3_headline > $out/pmc_headline.log 2>&1 || exit 1
bash tools/pmc_profile.sh r3_grid16k --scene grid16k --spp 1 > $out/pmc_grid16k.log 2>&1 || exit 1
bash tools/pmc_instmix.sh r3_headline > $out/instmix_headline.log 2>&1 || exit 1
INSTMIX_CALIB=0 bash tools/pmc_instmix.sh r3_grid16k --scene grid16k --spp 1 > $out/instmix_grid16k.log 2>&1 || exit 1
for g in 2 4 8; do timeout -k 10 300 python -u tools/band_balance.py --gpus $g; done > $out/band_balance.txt 2>&1
for g in 8; do timeout -k 10 300 python -u tools/band_balance.py --gpus $g --scene grid16k --spp 1; done >> $out/band_balance.txt 2>&1
cp gpurun_out/prof_r3_headline/summary.csv $out/headline_pmc.csv; cp gpurun_out/prof_r3_grid16k/summary.csv $out/grid16k_pmc.csv
cp gpurun_out/instmix_r3_headline/summary.csv $out/headline_instmix.csv; cp gpurun_out/instmix_r3_grid16k/summary.csv $out/grid16k_instmix.csv
ls -la $out
