// rt_math.h -- device vec3 arithmetic with the reference's exact operation order.
//
// The reference computes in GLM 0.9.9.0 scalar vec3 code (raytracer_cpp/glm/glm/detail):
//   dot       = (x*x' + y*y') + z*z'                      func_geometric.inl:54-61
//   normalize = v * (1 / sqrt(dot(v, v)))                 func_geometric.inl:88-96, func_exponential.inl:42-48
//   reflect   = I - (N * dot(N, I)) * 2                   func_geometric.inl:110-116
// A translation unit that includes this header for RT_MODE_EXACT must be compiled with
// -ffp-contract=off; division and sqrt are hipcc's default correctly rounded forms.
#ifndef RT_MATH_H
#define RT_MATH_H

#include <hip/hip_runtime.h>

struct V3 {
    float x, y, z;
};

__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(V3 a, V3 b)
{
    float x = a.x * b.x, y = a.y * b.y, z = a.z * b.z;
    return (x + y) + z;
}
__device__ __forceinline__ V3 normalize(V3 v)
{
    float inv = 1.0f / sqrtf(dot(v, v));
    return v * inv;
}
__device__ __forceinline__ V3 reflect(V3 i, V3 n)
{
    float d = dot(n, i);
    return i - (n * d) * 2.0f;
}

// float(pow(double(x), 10.0)) for x > 0 (main.cpp:169 resolves to ::pow(double,double) under g++).
// x^10 is evaluated in double-double so that the double result is the correctly rounded one,
// which is what glibc's pow returns (its error bound is < 1 ULP and it is exact-rounded except
// within ~2^-60 of a rounding boundary).
__device__ __forceinline__ float pow10_like_libm(float xf)
{
    double x2 = (double)xf * (double)xf;             // exact: 48 significant bits
    double h4 = x2 * x2;
    double l4 = fma(x2, x2, -h4);                    // x^4 = h4 + l4 exactly
    double h8 = h4 * h4;
    double l8 = fma(h4, h4, -h8) + 2.0 * h4 * l4;    // x^8 = h8 + l8 to ~2^-104
    double p = h8 * x2;
    double e = fma(h8, x2, -p) + l8 * x2;
    return (float)(p + e);
}

// int(double) as x86-64 cvttsd2si does it: NaN and out-of-range give INT_MIN (the reference's
// checkerboard depends on this for |pos| > 2^31).  AMD's v_cvt_i32_f64 saturates.
__device__ __forceinline__ int x86_double_to_int(double d)
{
    if (!(d > -2147483649.0 && d < 2147483648.0)) return (int)0x80000000;
    return (int)d;
}

// TiledPlane::GetMaterial, sceneobjects.h:106-115: floor() is ::floor(double), the sum is double.
__device__ __forceinline__ bool checker_white(V3 pos)
{
    double s = floor((double)pos.x) + floor((double)pos.z);
    return (x86_double_to_int(s) & 1) == 0;
}

// uint8_t(float) after clamp(color*255, 0, 255), main.cpp:213-216 (glm clamp = min(max(x,0),255)
// with '>' and '<', func_common.inl:15-28): NaN -> 0, truncation toward zero.
__device__ __forceinline__ uint32_t to_u8(float c)
{
    float v = c * 255.0f;
    v = v > 0.0f ? v : 0.0f;
    v = v < 255.0f ? v : 255.0f;
    return (uint32_t)(int)v & 0xffu;
}

#endif
