#!/bin/bash
# FP32/FP64/integer instruction mix of the trace kernel from the SQ_INSTS_VALU_* counters, each pass a run of its own,
# preceded by the same passes over tools/instmix_calib (kernels of ONE instruction kind and a known count), which tells
# what each counter counts on gfx950 (does a v_pk_fma_f32 count 1 or 2? is v_cndmask an INT32 instruction?).
# usage (GPU box, repo root): tools/pmc_instmix.sh <tag> [bench.py args]   ->   gpurun_out/instmix_<tag>/summary.csv
set -u
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/instmix_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
args="--no-cpu --steps 3 --warmup 1 $*"
PASS_A="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
PASS_B="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_IOPS SQ_INSTS_VALU_FLOPS_FP32_TRANS"
PASS_C="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"
PASS_D="SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES"
if [ "${INSTMIX_CALIB:-1}" = 1 ]; then
  timeout -s KILL 90 rocprofv3 --pmc $PASS_A --output-format csv -d "$out/calib_a" -- "$root/tools/instmix_calib" > "$out/calib_a.log" 2>&1 || exit 1
  timeout -s KILL 90 rocprofv3 --pmc $PASS_B --output-format csv -d "$out/calib_b" -- "$root/tools/instmix_calib" > "$out/calib_b.log" 2>&1 || exit 1
fi
timeout -s KILL 180 rocprofv3 --pmc $PASS_A --output-format csv -d "$out/mix_a" -- python3 "$root/bench.py" $args > "$out/mix_a.log" 2>&1 || exit 1
timeout -s KILL 180 rocprofv3 --pmc $PASS_B --output-format csv -d "$out/mix_b" -- python3 "$root/bench.py" $args > "$out/mix_b.log" 2>&1 || exit 1
timeout -s KILL 180 rocprofv3 --pmc $PASS_C --output-format csv -d "$out/mix_c" -- python3 "$root/bench.py" $args > "$out/mix_c.log" 2>&1 || exit 1
timeout -s KILL 180 rocprofv3 --pmc $PASS_D --output-format csv -d "$out/mix_d" -- python3 "$root/bench.py" $args > "$out/mix_d.log" 2>&1 || exit 1
cd "$root" && python3 tools/instmix_summary.py "$out" > "$out/summary.csv" && cat "$out/summary.csv"
