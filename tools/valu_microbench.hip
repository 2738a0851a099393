// valu_microbench.hip -- measures FP32 VALU issue rates on gfx950 (decides the sweep loop's
// instruction mix): v_fma_f32, v_mul+v_add (unfused), v_pk_fma_f32, v_pk_mul+v_pk_add.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/valu_microbench tools/valu_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));
#define ITER 32768

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float b, float c)
{
    float a[8];
    float2v p[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 1e-3f + i; p[i] = float2v{a[i], a[i] + 1}; }
    float2v pb = {b, b}, pc = {c, c};
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (MODE == 1) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c)); }
            if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
            if (MODE == 3) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc)); }
            if (MODE == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(b), "v"(c));
            if (MODE == 5) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (MODE == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (MODE == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (MODE == 8) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (MODE == 9) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "s"(c));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, double flop_per_inst_lane, int inst_per_iter, int waves_per_simd)
{
    int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves/block = 1 wave per SIMD) x waves_per_simd
    float *out; hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 1.0001f, 1e-6f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) k<MODE><<<blocks, 256>>>(out, 1.0001f, 1e-6f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double insts = (double)blocks * 4 * ITER * 8 * inst_per_iter;         // wave-instructions
    double flop = insts * 64 * flop_per_inst_lane;
    double cyc_per_inst_per_simd = (ms * 1e-3 * 2.4e9) / (insts / (256.0 * 4));
    printf("%-28s waves/SIMD %d: %.3f ms  %.1f TFLOP/s  %.2f clk/inst/SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms, flop / ms / 1e9, cyc_per_inst_per_simd);
    hipFree(out);
}

int main()
{
    for (int w : {2, 4, 8}) {
        run<5>("v_fmac_f32 (VOP2)", 2, 1, w);
        run<6>("v_mul_f32 (VOP2)", 1, 1, w);
        run<7>("v_add_f32 (VOP2)", 1, 1, w);
        run<8>("v_mul_f32_e64 (VOP3)", 1, 1, w);
        run<9>("v_sub_f32 sgpr (VOP2)", 1, 1, w);
        run<0>("v_fma_f32", 2, 1, w);
        run<4>("v_fma_f32 (sgpr operand)", 2, 1, w);
        run<1>("v_mul_f32+v_add_f32", 1, 2, w);
        run<2>("v_pk_fma_f32", 4, 1, w);
        run<3>("v_pk_mul_f32+v_pk_add_f32", 2, 2, w);
    }
    return 0;
}
