#!/bin/bash
# Usage (on the GPU box, from the repo root):  tools/pmc_profile.sh <tag> [bench.py args...]
# Collects, each in a run of its own: the kernel trace + stats, two SQ counter passes, FETCH_SIZE,
# WRITE_SIZE + GRBM.  Results land in gpurun_out/prof_<tag>/; tools/pmc_summary.py folds them.
set -u
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
args="--no-cpu --steps 3 --warmup 1 $*"
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$root/bench.py" $args > "$out/trace.log" 2>&1 || exit 1
timeout -s KILL 180 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$out/sq_a" -- python3 "$root/bench.py" $args > "$out/sq_a.log" 2>&1 || exit 1
timeout -s KILL 180 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_SMEM --output-format csv -d "$out/sq_b" -- python3 "$root/bench.py" $args > "$out/sq_b.log" 2>&1 || exit 1
timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 "$root/bench.py" $args > "$out/fetch.log" 2>&1 || exit 1
timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$out/write" -- python3 "$root/bench.py" $args > "$out/write.log" 2>&1 || exit 1
cd "$root" && python3 tools/pmc_summary.py "$out" > "$out/summary.csv" && cat "$out/summary.csv"
