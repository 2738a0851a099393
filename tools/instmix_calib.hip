// instmix_calib.hip -- what do gfx950's SQ_INSTS_VALU_* counters count?  One kernel per instruction kind, each issuing a
// KNOWN number of wave-instructions of that kind (1024 waves x 256 iterations x 8 = 2 097 152, plus a few dozen of set-up),
// run under `rocprofv3 --pmc ...` by tools/pmc_instmix.sh; tools/instmix_summary.py divides the counters by that number.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/instmix_calib tools/instmix_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float float2v __attribute__((ext_vector_type(2)));
#define ITER 256

#define CALIB_KERNEL(NAME, DECL, BODY)                                                   \
    __global__ __launch_bounds__(256) void NAME(float *out, float b, float c)            \
    {                                                                                    \
        float a[8];                                                                      \
        float2v p[8];                                                                    \
        double d[8];                                                                     \
        int n[8];                                                                        \
        for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 1e-3f + i + 1.0f; p[i] = float2v{a[i], a[i] + 1}; d[i] = a[i]; n[i] = threadIdx.x + i; } \
        float2v pb = {b, b}, pc = {c, c};                                                \
        double db = b, dc = c;                                                           \
        (void)pb; (void)pc; (void)db; (void)dc;                                          \
        DECL                                                                             \
        for (int it = 0; it < ITER; it++) {                                              \
            _Pragma("unroll") for (int i = 0; i < 8; i++) { BODY }                       \
        }                                                                                \
        float s = 0;                                                                     \
        for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y + (float)d[i] + (float)n[i]; \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                  \
    }

CALIB_KERNEL(calib_v_fma_f32, , asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));)
CALIB_KERNEL(calib_v_mul_f32, , asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));)
CALIB_KERNEL(calib_v_add_f32, , asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));)
CALIB_KERNEL(calib_v_pk_fma_f32, , asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));)
CALIB_KERNEL(calib_v_pk_mul_f32, , asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));)
CALIB_KERNEL(calib_v_pk_add_f32, , asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));)
CALIB_KERNEL(calib_v_sqrt_f32, , asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));)
CALIB_KERNEL(calib_v_rcp_f32, , asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));)
CALIB_KERNEL(calib_v_max_f32, , asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));)
CALIB_KERNEL(calib_v_min3_f32, , asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));)
CALIB_KERNEL(calib_v_cmp_f32, , asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(c) : "vcc");)
CALIB_KERNEL(calib_v_cndmask_b32, , asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : "vcc");)
CALIB_KERNEL(calib_v_mov_b32, , asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(c));)
CALIB_KERNEL(calib_v_add_u32, , asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));)
CALIB_KERNEL(calib_v_readlane_b32, int sg = 0;, asm volatile("v_readlane_b32 %0, %1, 3\n\ts_nop 3" : "=s"(sg) : "v"(n[i])); n[i] += sg;)
CALIB_KERNEL(calib_v_cvt_f64_f32, , asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(d[i]) : "v"(a[i]));)
CALIB_KERNEL(calib_v_fma_f64, , asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(db), "v"(dc));)
CALIB_KERNEL(calib_v_mul_f64, , asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));)
CALIB_KERNEL(calib_v_add_f64, , asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc));)
CALIB_KERNEL(calib_v_div_scale_f32, , asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(b) : "vcc");)
CALIB_KERNEL(calib_v_div_fmas_f32, , asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");)
CALIB_KERNEL(calib_v_div_fixup_f32, , asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));)

int main()
{
    float *out;
    const int blocks = 256;     // x 4 waves = 1024 waves
    hipMalloc(&out, blocks * 256 * sizeof(float));
#define RUN(NAME) NAME<<<blocks, 256>>>(out, 1.0001f, 1e-6f); hipDeviceSynchronize();
    RUN(calib_v_fma_f32) RUN(calib_v_mul_f32) RUN(calib_v_add_f32) RUN(calib_v_pk_fma_f32) RUN(calib_v_pk_mul_f32) RUN(calib_v_pk_add_f32)
    RUN(calib_v_sqrt_f32) RUN(calib_v_rcp_f32) RUN(calib_v_max_f32) RUN(calib_v_min3_f32) RUN(calib_v_cmp_f32) RUN(calib_v_cndmask_b32)
    RUN(calib_v_mov_b32) RUN(calib_v_add_u32) RUN(calib_v_readlane_b32) RUN(calib_v_cvt_f64_f32) RUN(calib_v_fma_f64) RUN(calib_v_mul_f64)
    RUN(calib_v_add_f64) RUN(calib_v_div_scale_f32) RUN(calib_v_div_fmas_f32) RUN(calib_v_div_fixup_f32)
    printf("instmix_calib: %d wave-instructions of its kind per kernel\n", blocks * 4 * ITER * 8);
    hipFree(out);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
