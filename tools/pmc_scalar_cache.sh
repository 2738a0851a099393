#!/bin/bash
# Scalar-cache behaviour of the trace kernel (the grouped sweep reads its records with s_load).
# usage (GPU box, repo root): tools/pmc_scalar_cache.sh <tag> [bench.py args]
set -u
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/sqc_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > "$out/avail.txt" 2>&1
grep -oE "SQC_[A-Z_0-9]+|SQ_WAIT[A-Z_0-9]*|SQ_INST_CYCLES[A-Z_0-9]*|SQ_INSTS_SMEM[A-Z_0-9]*" "$out/avail.txt" | sort -u > "$out/names.txt"
args="--no-cpu --steps 3 --warmup 1 $*"
timeout -s KILL 120 rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE --output-format csv -d "$out/sqc" -- python3 "$root/bench.py" $args > "$out/sqc.log" 2>&1 || exit 1
