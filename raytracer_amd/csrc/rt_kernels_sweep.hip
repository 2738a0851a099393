// rt_kernels_sweep.hip -- the production trace kernel for gfx950 (RT_MODE_EXACT and RT_MODE_FAST).
// Compile with -ffp-contract=off (the explicit fma()s below are kept).
//
// Geometry: one wave64 = one 8x8 pixel tile.  Spheres are stored in PAIRS, 32 B per pair,
//     A = {cx0, cx1, cy0, cy1}   B = {cz0, cz1, -r0^2, -r1^2}
// so that one record feeds v_pk_*_f32 instructions that test two spheres at once.  Two sweeps:
//   FLAT     every ray against every sphere.  A 256-thread workgroup = four tiles sharing ONE LDS
//            copy of the list (broadcast ds_read_b128 x2 per pair), streamed in chunks when it
//            does not fit; scenes of a few spheres run as single-wave workgroups.
//   GROUPED  groups of 8 spheres under two levels of bounding spheres; a wave visits a group only
//            if the line of one of its rays can touch the bound.  Records come through wave-uniform
//            scalar loads, workgroups are single waves, no LDS (see group_mask()).
// Both give the same bits.
//
// Per path segment a lane needs the nearest hit of 1 + E rays that all leave the same surface point
// `pos` (the reflection ray and one shadow ray per emissive object, main.cpp:108 and :112-151).
// They are swept TOGETHER, K <= 5 at a time, with a two-stage test per (ray, sphere):
//
//  1. conservative FILTER (the hot loop, packed FMA): with u = c - pos, uu = u.u, s = u.dn the
//     squared distance from the sphere centre to the ray's LINE is uu - s^2 wherever on the line the
//     origin sits, so u and uu are shared by all K rays and each ray costs 3 FMA + 1 FMA.  The
//     sphere is a candidate when  uu*(1-2^-17) - r^2 - s^2 <= 2^-19 * (|pos|^2 + 2^-13); the two
//     margins bound every rounding difference between this expression and the reference's
//     dSquared (DESIGN.md "filter soundness"), so no sphere the reference accepts is ever dropped.
//  2. EXACT test on candidates only (rare, divergent): glm::intersectRaySphere's arithmetic in the
//     reference's operation order with the ray's true origin pos + dir*0.001.
//
// The result is bit-identical to the reference while the hot loop runs ~6 VALU lane-ops per test
// instead of the ~17 of the straight transcription (rt_kernels_exact.hip, kept for A/B).
//
// Ray compaction (flat sweep): lanes of a tile die at different depths, so after the primary sweep the live
// hit points of the whole workgroup (4 tiles) are packed into a dense request queue in LDS each
// level; thread t then sweeps request t, so the waves that do sweep are full, and the result goes
// back through the same LDS slots to the pixel's home lane.  Requests that need a reflection ray
// are packed first, so trailing waves sweep K = E rays instead of E + 1.
//
// The recursion of TraceRay (main.cpp:84-188) is unrolled without a stack: the hit chain is traced
// forward (one combined sweep per segment also settles that segment's shadow rays as a bit mask),
// then shading runs deepest-first in the reference's operation order (RT_MODE_EXACT) or is
// folded in as each segment completes (RT_MODE_FAST, see the kernel's comment).
#include "rt_device.h"
#include "rt_math.h"

#include <float.h>

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));

#ifndef RT_WAVES_PER_BLOCK
#define RT_WAVES_PER_BLOCK 4
#endif
constexpr int kWavesPerBlock = RT_WAVES_PER_BLOCK;
constexpr int kBlock = 64 * kWavesPerBlock;
constexpr int kMaxK = 5;                       // rays swept together
#ifndef RT_WAVES_PER_SIMD
#define RT_WAVES_PER_SIMD 4
#endif
#ifndef RT_WAVES_PER_SIMD_GROUPED
#define RT_WAVES_PER_SIMD_GROUPED 4
#endif
#ifndef RT_WAVES_PER_SIMD_SMALL
#define RT_WAVES_PER_SIMD_SMALL 2                // scenes of a few spheres: registers over occupancy
#endif
constexpr float kRelKeep = 1.0f - 0x1p-17f;    // filter: uu * (1 - 2^-17)
constexpr float kAbsScale = 0x1p-19f;          // filter: 2^-19 * (|pos|^2 + 2^-13)
constexpr float kAbsFloor = 0x1p-13f;

// A wave-uniform read of read-only scene data through the scalar cache (s_load_dwordx4): the
// constant address space is what makes the compiler pick scalar loads.
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 uniform_load(const float4 *base, int idx)
{
    const __attribute__((address_space(4))) f4v *cp = (const __attribute__((address_space(4))) f4v *)base;
    const f4v v = cp[idx];
    return make_float4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ f2 splat(float x) { return f2{x, x}; }
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// glm::intersectRaySphere, gtx/intersect.inl:105-123, via Sphere::Intersects (sceneobjects.h:69-73)
__device__ __forceinline__ bool sphere_hit(float cx, float cy, float cz, float r2, V3 o, V3 dn, float &dist)
{
    V3 diff = v3(cx, cy, cz) - o;
    float t0 = dot(diff, dn);
    float dsq = dot(diff, diff) - t0 * t0;
    if (dsq > r2) return false;
    float t1 = sqrtf(r2 - dsq);
    dist = t0 > t1 + FLT_EPSILON ? t0 - t1 : t0 + t1;
    return dist > FLT_EPSILON;
}

// glm::intersectRayPlane, gtx/intersect.inl:7-24, via TiledPlane::Intersects (sceneobjects.h:127-130)
__device__ __forceinline__ bool plane_hit(const RtPlaneDev &p, V3 o, V3 d, float &dist)
{
    V3 n = v3(p.nx, p.ny, p.nz);
    float dn = dot(d, n);
    if (dn < FLT_EPSILON) {
        dist = dot(v3(p.ox, p.oy, p.oz) - o, n) / dn;
        return true;
    }
    return false;
}

__device__ __forceinline__ float filter_bound(V3 pos)
{
    return kAbsScale * (dot(pos, pos) + kAbsFloor);
}

template <int K>
struct RaySet {
    V3 o[K];        // true origin (exact test, plane test)
    V3 d[K];        // direction as the reference passes it (plane test, pos = o + d*t)
    V3 dn[K];       // normalize(d) (filter and exact sphere test)
    bool act[K];
    float best[K];
    int slot[K];    // >= 0 sphere slot, INT_MIN none, < 0 plane: -1 - plane slot
    int id[K];      // list index of the hit (filled by merge_planes)
    float dec[K];   // grouped sweep, shadow rays: a hit nearer than this decides the ray ("not lit"); -inf otherwise
    unsigned pairs_run, heads_run;   // pair records / head records this lane put its rays through (stats)
    unsigned cand_run;               // (ray, sphere) candidates that went to the exact test
};

// LDS image: [0, 2*cap) geometry pairs, [2*cap, 4*cap) camera-relative pairs (primary rays).
template <bool CHUNKED>
__device__ __forceinline__ void stage(const RtSceneDev &sc, float4 *lds, int cap, int base, int count,
                                      bool with_cam, V3 cam)
{
    if (CHUNKED) __syncthreads();              // every wave is done with the previous chunk
    for (int p = threadIdx.x; p < count; p += kBlock) {
        float4 A = sc.pairs[2 * (base + p)], B = sc.pairs[2 * (base + p) + 1];
        lds[2 * p] = A;
        lds[2 * p + 1] = B;
        if (with_cam) {
            // u = c - cam, w = uu*(1-2^-17) - r^2 for both spheres of the pair
            f2 ux = f2{A.x, A.y} - splat(cam.x), uy = f2{A.z, A.w} - splat(cam.y), uz = f2{B.x, B.y} - splat(cam.z);
            f2 uu = pk_fma(uz, uz, pk_fma(uy, uy, ux * ux));
            f2 w = pk_fma(uu, splat(kRelKeep), f2{B.z, B.w});
            lds[2 * cap + 2 * p] = make_float4(ux.x, ux.y, uy.x, uy.y);
            lds[2 * cap + 2 * p + 1] = make_float4(uz.x, uz.y, w.x, w.y);
        }
    }
    __syncthreads();
}

// Camera-relative table only (LDS-resident scenes): rebuilt per sample because the request queue
// of the bounce levels reuses its LDS region.
__device__ __forceinline__ void build_cam_table(float4 *lds, int cap, int count, V3 cam)
{
    __syncthreads();
    for (int p = threadIdx.x; p < count; p += kBlock) {
        const float4 A = lds[2 * p], B = lds[2 * p + 1];
        f2 ux = f2{A.x, A.y} - splat(cam.x), uy = f2{A.z, A.w} - splat(cam.y), uz = f2{B.x, B.y} - splat(cam.z);
        f2 uu = pk_fma(uz, uz, pk_fma(uy, uy, ux * ux));
        f2 w = pk_fma(uu, splat(kRelKeep), f2{B.z, B.w});
        lds[2 * cap + 2 * p] = make_float4(ux.x, ux.y, uy.x, uy.y);
        lds[2 * cap + 2 * p + 1] = make_float4(uz.x, uz.y, w.x, w.y);
    }
    __syncthreads();
}

constexpr int kQueueFields = 7;                          // request: pos, refl, -; result: lit, alt, dist, slot, id
constexpr int kQueueFloat4 = kQueueFields * kBlock / 4;  // LDS the queue needs, in float4

// FindNearestObject keeps the first object of the list among equally near ones (strict '>',
// main.cpp:72).  Flat layout: slots are visited in list order, so strict '>' here does the same.
// Grouped layout: slot order is not list order, so an exact tie goes to the lower list index.
template <bool GROUPED>
__device__ __forceinline__ bool nearer(const RtSceneDev &sc, float best, int best_slot, float dist, int slot)
{
    if (best > dist) return true;
    if (GROUPED && best == dist && best_slot >= 0) return sc.sph_id[slot] < sc.sph_id[best_slot];
    return false;
}

// The reference's test of ray k against both spheres of a pair record (Sphere::Intersects,
// sceneobjects.h:69-73): glm::intersectRaySphere's
// dSquared (gtx/intersect.inl:113-116) for the two spheres at once with unfused packed mul/add -- the
// same IEEE operations in the same order as the scalar code, (xx + yy) + zz for each dot -- then the
// square root and the distance only for a sphere that passes dSquared <= r*r (:117-122).  A lane far
// from the origin makes most spheres near its line candidates (the filter's allowance grows with
// |pos|^2 like the reference's own rounding), so in the grouped sweep, where such lanes set the length
// of the frame's slowest waves, this path has to be cheap, not just rare.
template <int K, bool GROUPED>
__device__ __forceinline__ void exact_pair(const RtSceneDev &sc, RaySet<K> &r, const f2 (&q)[K], float bound, float4 A,
                                           float4 B, int slot0, f2 (&dirx)[K], f2 (&diry)[K], f2 (&dirz)[K])
{
    const bool second = GROUPED || slot0 + 1 < sc.n_spheres;    // (padding spheres have -r*r = +FLT_MAX: q is never <= bound)
    if constexpr (!GROUPED) {
        // flat sweep: one sphere at a time (measured: the packed form below costs the 6-object scene 7 %
        // and gains the compacted kernels nothing)
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (!r.act[k]) continue;
            float dist;
            r.cand_run += (q[k].x <= bound ? 1u : 0u) + (q[k].y <= bound ? 1u : 0u);
            if (q[k].x <= bound && sphere_hit(A.x, A.z, B.x, -B.z, r.o[k], r.dn[k], dist) &&
                nearer<false>(sc, r.best[k], r.slot[k], dist, slot0)) {
                r.best[k] = dist;
                r.slot[k] = slot0;
            }
            if (q[k].y <= bound && second && sphere_hit(A.y, A.w, B.y, -B.w, r.o[k], r.dn[k], dist) &&
                nearer<false>(sc, r.best[k], r.slot[k], dist, slot0 + 1)) {
                r.best[k] = dist;
                r.slot[k] = slot0 + 1;
            }
        }
        return;
    }
    const f2 cx = f2{A.x, A.y}, cy = f2{A.z, A.w}, cz = f2{B.x, B.y};
#pragma unroll
    for (int k = 0; k < K; k++) {
        const bool c0 = r.act[k] && q[k].x <= bound, c1 = r.act[k] && q[k].y <= bound && second;
        if (!(c0 || c1)) continue;
        r.cand_run += (c0 ? 1u : 0u) + (c1 ? 1u : 0u);
        const f2 dx = cx - splat(r.o[k].x), dy = cy - splat(r.o[k].y), dz = cz - splat(r.o[k].z);       // diff = center - rayStarting
        const f2 t0 = (dx * splat(r.dn[k].x) + dy * splat(r.dn[k].y)) + dz * splat(r.dn[k].z);     // dot(diff, rayNormalizedDirection)
        const f2 dsq = ((dx * dx + dy * dy) + dz * dz) - t0 * t0;                                   // dot(diff, diff) - t0 * t0
        if (c0 && !(dsq.x > -B.z)) {
            const float t1 = sqrtf(-B.z - dsq.x);
            const float dist = t0.x > t1 + FLT_EPSILON ? t0.x - t1 : t0.x + t1;
            if (dist > FLT_EPSILON && nearer<GROUPED>(sc, r.best[k], r.slot[k], dist, slot0)) {
                r.best[k] = dist;
                r.slot[k] = slot0;
                if (dist < r.dec[k]) dirx[k] = diry[k] = dirz[k] = splat(0.0f);
            }
        }
        if (c1 && !(dsq.y > -B.w)) {
            const float t1 = sqrtf(-B.w - dsq.y);
            const float dist = t0.y > t1 + FLT_EPSILON ? t0.y - t1 : t0.y + t1;
            if (dist > FLT_EPSILON && nearer<GROUPED>(sc, r.best[k], r.slot[k], dist, slot0 + 1)) {
                r.best[k] = dist;
                r.slot[k] = slot0 + 1;
                if (dist < r.dec[k]) dirx[k] = diry[k] = dirz[k] = splat(0.0f);
            }
        }
    }
}

// The filter + exact test of K rays against one pair record.
template <int K, bool GROUPED>
__device__ __forceinline__ void test_pair(const RtSceneDev &sc, RaySet<K> &r, const float4 A, const float4 B, int slot0,
                                          const f2 px, const f2 py, const f2 pz, f2 (&dx)[K], f2 (&dy)[K],
                                          f2 (&dz)[K], float bound)
{
    const f2 ux = f2{A.x, A.y} - px, uy = f2{A.z, A.w} - py, uz = f2{B.x, B.y} - pz;
    const f2 uu = pk_fma(uz, uz, pk_fma(uy, uy, ux * ux));
    const f2 w = pk_fma(uu, splat(kRelKeep), f2{B.z, B.w});
    f2 q[K];
    // the K chains are independent: issue them stage by stage
#pragma unroll
    for (int k = 0; k < K; k++) q[k] = ux * dx[k];
#pragma unroll
    for (int k = 0; k < K; k++) q[k] = pk_fma(uy, dy[k], q[k]);
#pragma unroll
    for (int k = 0; k < K; k++) q[k] = pk_fma(uz, dz[k], q[k]);
#pragma unroll
    for (int k = 0; k < K; k++) q[k] = pk_fma(-q[k], q[k], w);
    float m = fminf(q[0].x, q[0].y);
#pragma unroll
    for (int k = 1; k < K; k++) m = fminf(fminf(m, q[k].x), q[k].y);
    if (m <= bound) exact_pair<K, GROUPED>(sc, r, q, bound, A, B, slot0, dx, dy, dz);
}

// Grouped sweep: which of the bounding spheres [2*h0, 2*h0 + 2*hn) of `heads` can any ray of this
// WAVE touch?  (Group heads, or the super-group heads one level up: same record, same test.)
// A bounding sphere (centre g, radius R) is skipped when every ray {pos + t d, t >= 0} passes it at
// more than R + e, with e^2 = bound + 2^-14 (|g - pos|^2 + R^2).  Soundness: a sphere the reference
// accepts has true line distance^2 <= r^2 + 2^-20 |c - o|^2 (its float dSquared <= r*r) and its hit
// lies ahead of o (distance > FLT_EPSILON, gtx/intersect.inl:121-122), i.e. on the ray from o, a point
// of which is within r of c; the ray from
// o = pos + d*0.001 is part of the ray from pos up to the rounding of o (covered by `bound`); and
// dist(ray, g) <= dist(ray, c) + |c - g| <= dist(ray, c) + R - r, so the bound passes with room
// to spare: e^2 is 32x the rounding it has to cover, which also absorbs the float evaluation of this
// test itself.  The records are read with wave-uniform scalar loads; no LDS is involved.
template <int K>
__device__ __forceinline__ unsigned long long group_mask(const float4 *heads, int h0, int hn, int n_total, bool lane_on,
                                                         const f2 px, const f2 py, const f2 pz, const f2 (&dx)[K],
                                                         const f2 (&dy)[K], const f2 (&dz)[K], float bound)
{
    unsigned long long mask = 0;
    for (int h = 0; h < hn; h++) {
        const float4 A = uniform_load(heads, 2 * (h0 + h)), B = uniform_load(heads, 2 * (h0 + h) + 1);
        const f2 ux = f2{A.x, A.y} - px, uy = f2{A.z, A.w} - py, uz = f2{B.x, B.y} - pz;
        const f2 uu = pk_fma(uz, uz, pk_fma(uy, uy, ux * ux));
        const f2 R = f2{B.z, B.w};
        const f2 e2 = pk_fma(pk_fma(R, R, uu), splat(0x1p-14f), splat(bound));
        const f2 Re = R + f2{__builtin_amdgcn_sqrtf(e2.x), __builtin_amdgcn_sqrtf(e2.y)};
        const f2 w = pk_fma(-Re, Re, uu);
        f2 q[K];
#pragma unroll
        for (int k = 0; k < K; k++) q[k] = ux * dx[k];
#pragma unroll
        for (int k = 0; k < K; k++) q[k] = pk_fma(uy, dy[k], q[k]);
#pragma unroll
        for (int k = 0; k < K; k++) q[k] = pk_fma(uz, dz[k], q[k]);
        // the RAY, not its line: a centre behind the origin (s = u.d < 0) is |u| away from the ray, so
        // q = w - max(s, 0) * s (written so that a NaN s still gives a NaN q and keeps the group)
#pragma unroll
        for (int k = 0; k < K; k++) q[k] = pk_fma(-f2{fmaxf(q[k].x, 0.0f), fmaxf(q[k].y, 0.0f)}, q[k], w);
        float m0 = q[0].x, m1 = q[0].y;
#pragma unroll
        for (int k = 1; k < K; k++) { m0 = fminf(m0, q[k].x); m1 = fminf(m1, q[k].y); }
        // !(m > 0): a NaN (overflowing radius, non-finite point) keeps the group
        const bool p0 = lane_on && !(m0 > 0.0f), p1 = lane_on && !(m1 > 0.0f) && 2 * (h0 + h) + 1 < n_total;
        if (__ballot(p0)) mask |= 1ull << (2 * h);
        if (__ballot(p1)) mask |= 2ull << (2 * h);
    }
    return mask;
}

// K rays leaving `pos` against every sphere.  FindNearestObject's order (main.cpp:65-82): spheres
// are visited in list order with a strict '>' so the lowest index wins ties.
template <int K, bool CHUNKED, bool GROUPED>
__device__ __forceinline__ void sweep(const RtSceneDev &sc, float4 *lds, int cap, bool lane_on, V3 pos, RaySet<K> &r)
{
    f2 dx[K], dy[K], dz[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        r.best[k] = FLT_MAX;
        r.slot[k] = INT_MIN;
        V3 dn = r.act[k] ? r.dn[k] : v3(0.0f, 0.0f, 0.0f);
        dx[k] = splat(dn.x); dy[k] = splat(dn.y); dz[k] = splat(dn.z);
    }
    const float bound = filter_bound(pos);
    const f2 px = splat(pos.x), py = splat(pos.y), pz = splat(pos.z);
    r.pairs_run = r.heads_run = r.cand_run = 0;
    if constexpr (GROUPED) {
        // two levels: super-groups of RT_SUPER_GROUPS groups, groups of 2*RT_GROUP_PAIRS spheres; the
        // masks are wave-uniform (SGPRs), so all of this is scalar control flow
        constexpr int kHeadsPerSuper = RT_SUPER_GROUPS / 2;
        const int n_super_heads = (sc.n_supers + 1) >> 1, n_heads = (sc.n_groups + 1) >> 1;
        for (int s0 = 0; s0 < n_super_heads; s0 += 32) {
            const int sn = min(32, n_super_heads - s0);
            unsigned long long smask = group_mask<K>(sc.supers, s0, sn, sc.n_supers, lane_on, px, py, pz, dx, dy, dz, bound);
            r.heads_run += sn;
            while (smask) {
                const int sg = 2 * s0 + __builtin_ctzll(smask);
                smask &= smask - 1;
                const int h0 = sg * kHeadsPerSuper, hn = min(kHeadsPerSuper, n_heads - h0);
                unsigned long long mask = group_mask<K>(sc.heads, h0, hn, sc.n_groups, lane_on, px, py, pz, dx, dy, dz, bound);
                r.heads_run += hn;
                while (mask) {
                    const int g = 2 * h0 + __builtin_ctzll(mask);
                    mask &= mask - 1;
                    r.pairs_run += RT_GROUP_PAIRS;
                    if (lane_on) {
#pragma unroll 2
                        for (int p = g * RT_GROUP_PAIRS; p < (g + 1) * RT_GROUP_PAIRS; p++)
                            test_pair<K, true>(sc, r, uniform_load(sc.pairs, 2 * p), uniform_load(sc.pairs, 2 * p + 1), 2 * p,
                                               px, py, pz, dx, dy, dz, bound);
                    }
                }
            }
        }
    } else {
        for (int base = 0; base < sc.n_pairs; base += cap) {
            const int count = min(cap, sc.n_pairs - base);
            if (CHUNKED) stage<true>(sc, lds, cap, base, count, false, pos);
            r.pairs_run += count;
            if (lane_on) {
                // (Measured: prefetching the next pair by hand costs 15 %; four waves per SIMD already
                // cover the LDS latency.)
#pragma unroll 2
                for (int p = 0; p < count; p++)
                    test_pair<K, false>(sc, r, lds[2 * p], lds[2 * p + 1], 2 * (base + p), px, py, pz, dx, dy, dz, bound);
            }
        }
    }
}

// Primary rays: every lane's origin is the camera, so u and w come precomputed from LDS.
template <bool CHUNKED>
__device__ __forceinline__ void sweep_primary(const RtSceneDev &sc, float4 *lds, int cap, bool lane_on, V3 cam,
                                              RaySet<1> &r)
{
    r.best[0] = FLT_MAX;
    r.slot[0] = INT_MIN;
    r.pairs_run = (unsigned)sc.n_pairs;
    r.heads_run = r.cand_run = 0;
    const f2 dx = splat(r.dn[0].x), dy = splat(r.dn[0].y), dz = splat(r.dn[0].z);
    const float bound = filter_bound(cam);
    const int n_pairs = sc.n_pairs;
    for (int base = 0; base < n_pairs; base += cap) {
        const int count = min(cap, n_pairs - base);
        if (CHUNKED) stage<true>(sc, lds, cap, base, count, true, cam);
        if (lane_on) {
#pragma unroll 4
            for (int p = 0; p < count; p++) {
                const float4 U = lds[2 * cap + 2 * p], W = lds[2 * cap + 2 * p + 1];
                const f2 s = pk_fma(f2{W.x, W.y}, dz, pk_fma(f2{U.z, U.w}, dy, f2{U.x, U.y} * dx));
                f2 q[1];
                q[0] = pk_fma(-s, s, f2{W.z, W.w});
                if (fminf(q[0].x, q[0].y) <= bound) {
                    f2 ex[1] = {dx}, ey[1] = {dy}, ez[1] = {dz};    // (unused by the flat sweep's exact test)
                    exact_pair<1, false>(sc, r, q, bound, lds[2 * p], lds[2 * p + 1], 2 * (base + p), ex, ey, ez);
                }
            }
        }
    }
}

// Planes are few: the reference's own arithmetic per ray, merged with an explicit
// (distance, list index) comparison, which is what the in-order strict '>' sweep amounts to.
template <int K>
__device__ __forceinline__ void merge_planes(const RtSceneDev &sc, bool lane_on, RaySet<K> &r)
{
#pragma unroll
    for (int k = 0; k < K; k++) {
        r.id[k] = -1;
        if (!lane_on || !r.act[k]) continue;
        if (r.slot[k] >= 0) r.id[k] = sc.sph_id[r.slot[k]];
        for (int j = 0; j < sc.n_planes; j++) {
            const RtPlaneDev p = sc.planes[j];
            float dist;
            if (plane_hit(p, r.o[k], r.d[k], dist)) {
                if (r.best[k] > dist || (r.best[k] == dist && r.id[k] >= 0 && p.id < r.id[k])) {
                    r.best[k] = dist;
                    r.slot[k] = -1 - j;
                    r.id[k] = p.id;
                }
            }
        }
    }
}

template <bool CHUNKED>
__device__ __forceinline__ bool block_any(bool v)
{
    if (CHUNKED) return __syncthreads_or(v) != 0;
    return __any(v) != 0;
}

struct SweepCount {       // stats: (ray, sphere) filters, (ray, group) tests, exact tests this lane executed
    unsigned tests, gtests, cands;
};

struct Segment {          // what the next level needs from the sweep that found it
    V3 o, d;              // the ray that produced the hit
    float dist;
    int slot, id;
};

// Sweeps `ne` shadow rays (emitters e0 .. e0+ne-1) and, when WITH_REFL, the reflection ray in slot 0.
template <int K, bool WITH_REFL, bool CHUNKED, bool GROUPED>
__device__ __forceinline__ void trace_group(const RtSceneDev &sc, float4 *lds, int cap, bool on, V3 pos, V3 refl,
                                            bool want_refl, int e0, unsigned long long &lit,
                                            unsigned long long &alt, Segment &next, bool &next_hit, SweepCount &cnt)
{
    RaySet<K> r;
    constexpr int kFirst = WITH_REFL ? 1 : 0;
    if (WITH_REFL) {
        r.o[0] = pos + refl * 0.001f;                                    // main.cpp:108
        r.d[0] = refl;
        r.dn[0] = normalize(refl);                                       // sceneobjects.h:71
        r.act[0] = want_refl;
        r.dec[0] = -INFINITY;
    }
#pragma unroll
    for (int k = kFirst; k < K; k++) {
        const RtEmitterDev em = sc.emit[sc.emit[e0 + k - kFirst].order];
        const V3 ev = v3(em.cx, em.cy, em.cz) - pos;
        const V3 edir = normalize(ev);                                   // GetRayFrom, main.cpp:114
        r.o[k] = pos + edir * 0.001f;                                    // main.cpp:121
        r.d[k] = edir;
        r.dn[k] = normalize(edir);
        r.act[k] = on;
        // The shadow ray is lit only if the emitter itself is the nearest hit, and the emitter cannot be
        // hit nearer than |c - pos| - r: any hit clearly in front of that decides "not lit", and a
        // decided ray stops asking for groups (grouped sweep; its direction is zeroed in exact_pair).
        // The allowance covers the 0.001 step to the origin and the rounding of both distances.
        if (GROUPED) {
            const float L = __builtin_amdgcn_sqrtf(dot(ev, ev));
            r.dec[k] = (L - em.radius) - (0.01f + 0x1p-10f * (L + em.radius));
        }
    }
    sweep<K, CHUNKED, GROUPED>(sc, lds, cap, on, pos, r);
    merge_planes<K>(sc, on, r);
    if (on) {
        const unsigned n_act = (unsigned)(K - kFirst) + ((WITH_REFL && want_refl) ? 1u : 0u);
        cnt.tests += 2u * n_act * r.pairs_run;
        cnt.gtests += 2u * n_act * r.heads_run;
        cnt.cands += r.cand_run;
#pragma unroll
        for (int k = kFirst; k < K; k++) {
            // lit iff the nearest object along the shadow ray (lowest index on ties) is the emitter
            // itself and its material at that point is emissive (main.cpp:119-157)
            const int e = sc.emit[e0 + k - kFirst].order;
            const RtEmitterDev em = sc.emit[e];
            bool is_lit = r.slot[k] != INT_MIN && r.id[k] == em.id;
            if (is_lit && em.is_plane) {
                const bool white = checker_white(pos + r.d[k] * r.best[k]);        // main.cpp:130
                const rt_material pm = sc.mats[2 * em.id + (white ? 0 : 1)];
                is_lit = !(pm.emissive[0] == 0.0f && pm.emissive[1] == 0.0f && pm.emissive[2] == 0.0f);
                if (!white) alt |= 1ull << e;
            }
            if (is_lit) lit |= 1ull << e;
        }
        if (WITH_REFL && want_refl) {
            next_hit = r.slot[0] != INT_MIN;
            next.o = r.o[0];
            next.d = r.d[0];
            next.dist = r.best[0];
            next.slot = r.slot[0];
            next.id = r.id[0];
        }
    }
}

template <bool WITH_REFL, bool CHUNKED, bool GROUPED>
__device__ __forceinline__ void trace_group_k(int k_total, const RtSceneDev &sc, float4 *lds, int cap, bool on, V3 pos,
                                              V3 refl, bool want_refl, int e0, unsigned long long &lit,
                                              unsigned long long &alt, Segment &next, bool &next_hit, SweepCount &cnt)
{
    switch (k_total) {       // uniform across the workgroup
    case 1: trace_group<1, WITH_REFL, CHUNKED, GROUPED>(sc, lds, cap, on, pos, refl, want_refl, e0, lit, alt, next, next_hit, cnt); break;
    case 2: trace_group<2, WITH_REFL, CHUNKED, GROUPED>(sc, lds, cap, on, pos, refl, want_refl, e0, lit, alt, next, next_hit, cnt); break;
    case 3: trace_group<3, WITH_REFL, CHUNKED, GROUPED>(sc, lds, cap, on, pos, refl, want_refl, e0, lit, alt, next, next_hit, cnt); break;
    case 4: trace_group<4, WITH_REFL, CHUNKED, GROUPED>(sc, lds, cap, on, pos, refl, want_refl, e0, lit, alt, next, next_hit, cnt); break;
    default: trace_group<5, WITH_REFL, CHUNKED, GROUPED>(sc, lds, cap, on, pos, refl, want_refl, e0, lit, alt, next, next_hit, cnt); break;
    }
}

// The emitter loop of TraceRay (main.cpp:112-184) for the emitters whose shadow ray came back lit:
// adds each light's contribution to `out` in list order.  FASTPOW: x^10 by four float multiplies
// instead of the double-precision pow the reference calls (RT_MODE_FAST only).
template <bool STATS, bool FASTPOW>
__device__ __forceinline__ V3 add_lights(const RtSceneDev &sc, V3 out, V3 pos, V3 n, V3 refl, const rt_material &m,
                                         unsigned long long lit, unsigned long long alt, unsigned &n_lit)
{
    for (int e = 0; e < sc.n_emit; e++) {                                    // main.cpp:112
        if (!((lit >> e) & 1ull)) continue;
        const RtEmitterDev em = sc.emit[e];
        const V3 edir = normalize(v3(em.cx, em.cy, em.cz) - pos);            // main.cpp:114
        float diffuse = dot(n, edir);                                        // main.cpp:162
        float spec = 0.0f;
        if (diffuse > 0.0f) {
            spec = dot(refl, edir);                                          // main.cpp:166
            if (spec > 0.0f) {
                if (FASTPOW) {
                    const float s2 = spec * spec, s4 = s2 * s2;
                    spec = (s4 * s4) * s2;
                } else {
                    spec = pow10_like_libm(spec);                            // main.cpp:169
                }
                spec = 0.0f < spec ? spec : 0.0f;                            // main.cpp:170
            } else {
                spec = 0.0f;
            }
        } else {
            diffuse = 0.0f;
        }
        if (STATS) n_lit++;
        const float *ee = sc.mats[2 * em.id + (int)((alt >> e) & 1ull)].emissive;
        const V3 l = (v3(ee[0], ee[1], ee[2]) * v3(m.albedo[0], m.albedo[1], m.albedo[2])) * diffuse +
                     v3(m.specular[0], m.specular[1], m.specular[2]) * spec;   // main.cpp:183
        out = out + l;
    }
    return out;
}

// main.cpp:210-216 and the options of raytracer_win: scale the sum of the samples, fold it into the
// running mean when asked to, store the float pixel and its 8-bit quantisation.
__device__ __forceinline__ void write_pixel(const RtFrameDev &fr, size_t idx, V3 color, float w)
{
    color = color * fr.inv_spp;                                              // main.cpp:210
    if (fr.accumulate) {
        // raytracer_win/main.cpp:314: bufferVal = ((bufferVal * k1) + vec4(color, 1)) * k2; sample 0
        // starts from the zero-initialised buffer (:105)
        const float4 old = fr.k1 == 0.0f ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : fr.f32x4[idx];
        color = ((v3(old.x, old.y, old.z) * fr.k1) + color) * fr.k2;
        w = ((old.w * fr.k1) + 1.0f) * fr.k2;
    }
    if (fr.f32x4) fr.f32x4[idx] = make_float4(color.x, color.y, color.z, w);
    if (fr.rgba8) fr.rgba8[idx] = to_u8(color.x) | (to_u8(color.y) << 8) | (to_u8(color.z) << 16) | 0xff000000u;
}

// Sample split, second half: a pixel's samples were traced by different waves; add them in sample
// order -- main.cpp:203-209's `color +=` sequence, so the sum is the same float -- and write the pixel.
__global__ __launch_bounds__(256) void resolve_samples_kernel(RtFrameDev fr, unsigned n_px)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_px) return;
    V3 color = v3(0.0f, 0.0f, 0.0f);
    float rays = 0.0f;
    for (int s = 0; s < fr.spp; s++) {
        const float4 v = fr.sample_buf[(size_t)s * n_px + i];
        color = color + v3(v.x, v.y, v.z);                                   // main.cpp:208
        rays += v.w;                                                         // small integers: exact
    }
    write_pixel(fr, i, color, rays);
}

// COMPACT: pack the live hit points of the workgroup before every bounce sweep (pays off once a
// sweep is long enough to dwarf three barriers: scenes of more than a few dozen spheres).
//
// FAST (RT_MODE_FAST): north_star's formulation -- no per-segment records.  The recursion
//     out_l = (out_{l+1} * r_l + L_l) * (1 - r_l) + e_l                    (main.cpp:109-186)
// is unrolled top-down with a per-lane throughput T_0 = 1, T_{l+1} = T_l * r_l * (1 - r_l):
//     out_0 = sum_l T_l * (L_l * (1 - r_l) + e_l)  [+ T_{last+1} * 0.1 when the last ray misses],
// so a segment is shaded as soon as its shadow mask is known and nothing is kept in private memory.
// Same hits, same lights; the sum associates differently, so radiance agrees with the reference to
// rounding (<= 1e-5 relative, tested at 1e-4), not bit for bit.
template <bool CHUNKED, bool COMPACT, bool STATS, int LEVELS, bool FAST, bool GROUPED>
__global__ __launch_bounds__((GROUPED || (!CHUNKED && !COMPACT)) ? 64 : kBlock, GROUPED ? RT_WAVES_PER_SIMD_GROUPED : (CHUNKED || COMPACT) ? RT_WAVES_PER_SIMD : RT_WAVES_PER_SIMD_SMALL) void trace_sweep_kernel(RtSceneDev sc, rt_camera cam, RtFrameDev fr, RtSamples smp,
                                                             RtStatsDev *stats)
{
    extern __shared__ float4 lds[];
    const int cap = GROUPED ? 0 : fr.chunk;           // pairs resident in LDS

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int rows = fr.row_end - fr.row_begin;
    const int tiles_x = (fr.width + fr.tile_w - 1) / fr.tile_w;
    const int tiles_y = (rows + fr.tile_h - 1) / fr.tile_h;
    // grouped sweep: nothing is shared between waves (no LDS scene copy, no compaction), so a
    // workgroup is a single wave and a finished tile frees its slot at once
    // (the same goes for scenes of a few spheres, which do not compact: their LDS copy is tiny)
    //
    // Two ways to give a wave one sample per lane instead of a loop over samples (a tile whose rays are
    // expensive -- grazing rays near the horizon take 12x the average -- then holds its waves a
    // fraction as long):
    //  * samples in lanes (single-wave kernels, spp a power of two): the wave covers 64/spp pixels
    //    (4x4 at 4 spp), lane = pixel * spp + sample; the samples of a pixel are added up in sample
    //    order across lanes at the end.  The smaller footprint also narrows the set of groups a wave
    //    has to visit, and the samples of a pixel end their paths together.
    //  * sample split (otherwise): spp waves per 8x8 tile, each writes its sample's radiance to a
    //    scratch plane that resolve_samples_kernel adds up in sample order.
    const int ls = fr.lane_samples;                                          // 0 or spp
    const bool split = fr.sample_buf != nullptr;
    const int unit = blockIdx.x * ((GROUPED || (!CHUNKED && !COMPACT)) ? 1 : kWavesPerBlock) + wave;
    const int tile = split ? unit / fr.spp : unit;
    const int lane_sample = ls ? lane & (ls - 1) : 0, lane_pixel = ls ? lane / ls : lane;
    const int s_begin = split ? unit - tile * fr.spp : lane_sample, s_end = (split || ls) ? s_begin + 1 : fr.spp;
    const int tx = tile % tiles_x;
    int ty = tile / tiles_x + fr.first_tile_row;                             // launch order: hot rows first
    if (ty >= tiles_y && tile < tiles_x * tiles_y) ty -= tiles_y;
    const int x = tx * fr.tile_w + lane_pixel % fr.tile_w;
    const int yr = ty * fr.tile_h + lane_pixel / fr.tile_w;
    const int y = fr.row_begin + yr;
    // column strips dealt round-robin to several renders: x is the column of the (narrower) output planes,
    // x_img the image column the ray goes through (tiles are at most RT_STRIP_COLS wide: none straddles strips)
    const int x_img = fr.strip_stride ? (((x >> 3) * fr.strip_stride + fr.strip_phase) << 3) + (x & 7) : x;
    const bool in_image = (tile < tiles_x * tiles_y) && x < fr.width && yr < rows;

    const V3 cam_pos = v3(cam.position[0], cam.position[1], cam.position[2]);
    const V3 cam_view = v3(cam.view_direction[0], cam.view_direction[1], cam.view_direction[2]);
    const V3 cam_right = v3(cam.right[0], cam.right[1], cam.right[2]);
    const V3 cam_up = v3(cam.up[0], cam.up[1], cam.up[2]);

    __shared__ int q_count[2];                        // [0] requests with a reflection ray, [1] without
    unsigned *const queue = reinterpret_cast<unsigned *>(lds + 2 * cap);   // aliases the camera table
    if (threadIdx.x == 0) q_count[0] = q_count[1] = 0;
    // (grouped layout: the scene is read through the scalar cache, LDS holds the request queue only)
    if (!CHUNKED && !GROUPED) stage<false>(sc, lds, cap, 0, sc.n_pairs, false, cam_pos);
    else if (GROUPED) __syncthreads();                 // q_count

    // one record per path segment, read back deepest-first (runtime-indexed -> private memory)
    // LEVELS >= max_depth + 1 (the launcher picks the smallest instantiation that fits: the
    // footprint of this private memory, times 262 144 resident lanes, decides whether the record
    // traffic stays in the 256 MB Infinity Cache or goes to HBM)
    constexpr int kRec = FAST ? 1 : LEVELS;
    float rec_pos[kRec][3], rec_n[kRec][3], rec_r[kRec][3];
    int rec_mat[kRec];
    unsigned long long rec_lit[kRec], rec_alt[kRec];

    unsigned n_refl = 0, n_shadow = 0, n_hits = 0, n_lit = 0;
    SweepCount cnt = {0u, 0u, 0u};
    V3 color = v3(0.0f, 0.0f, 0.0f);

    for (int s = s_begin; s < s_end; s++) {                                  // main.cpp:203
        Segment seg;
        bool alive;
        {
            // Camera::GetWorldRay, camera.h:48-62
            float sx = (float)x_img + smp.xy[2 * s], sy = (float)y + smp.xy[2 * s + 1];   // main.cpp:205
            float px = ((sx * 2.0f) / cam.image_width) - 1.0f;
            float py = ((sy * 2.0f) / cam.image_height) - 1.0f;
            V3 d = cam_view;
            d = d + cam_right * (cam.half_angle * cam.aspect_ratio * px);
            d = d - cam_up * (cam.half_angle * py);
            RaySet<1> r;
            if (fr.lens_on) {
                // thin lens, raytracer_win/camera.h:160-171: aim at the pinhole ray's point at focus
                // distance from the sample's point on the lens
                const float ft = (fr.focal_length - 1.0f) / sqrtf(dot(d, d));
                const V3 focus = cam_pos + d * ft;
                V3 lens = cam_pos;
                lens = lens + cam_right * smp.lens[2 * s];
                lens = lens + cam_up * smp.lens[2 * s + 1];
                r.o[0] = lens;
                r.d[0] = normalize(focus - lens);
            } else {
                r.o[0] = cam_pos;
                r.d[0] = normalize(d);
            }
            r.dn[0] = normalize(r.d[0]);                                     // sceneobjects.h:71
            r.act[0] = in_image;
            r.dec[0] = -INFINITY;
            if constexpr (GROUPED) {
                sweep<1, false, true>(sc, lds, cap, in_image, r.o[0], r);
            } else if (fr.lens_on) {
                // every lane has its own origin: the general sweep (the camera table assumes one origin)
                sweep<1, CHUNKED, false>(sc, lds, cap, in_image, r.o[0], r);
            } else {
                if (!CHUNKED) build_cam_table(lds, cap, sc.n_pairs, cam_pos);
                sweep_primary<CHUNKED>(sc, lds, cap, in_image, cam_pos, r);
            }
            merge_planes<1>(sc, in_image, r);
            if (in_image) { cnt.tests += 2u * r.pairs_run; cnt.gtests += 2u * r.heads_run; cnt.cands += r.cand_run; }
            seg.o = r.o[0]; seg.d = r.d[0]; seg.dist = r.best[0]; seg.slot = r.slot[0]; seg.id = r.id[0];
            alive = in_image && r.slot[0] != INT_MIN;
        }
        bool miss_tail = in_image && !alive;                                 // main.cpp:90-93
        int nlev = 0;
        float tp = 1.0f;                                                     // FAST: throughput
        V3 acc = miss_tail ? v3(0.1f, 0.1f, 0.1f) : v3(0.0f, 0.0f, 0.0f);    // FAST: radiance so far

        // ---- forward: hit chain + shadow masks, live hit points compacted across the workgroup ----
        for (int depth = 0; depth <= fr.max_depth; depth++) {
            V3 pos = cam_pos, refl = v3(0, 0, 0), n = v3(0, 0, 0);
            bool want_refl = false;
            int my_slot = 0, mat = 0;
            if (COMPACT) __syncthreads();     // previous level's results are consumed, counters are reset
            else if (!block_any<CHUNKED>(alive)) break;
            {
                if (alive) {
                    if (STATS) n_hits++;
                    pos = seg.o + seg.d * seg.dist;                          // main.cpp:94
                    if (seg.slot >= 0) {
                        const float4 A = sc.pairs[2 * (seg.slot >> 1)], B = sc.pairs[2 * (seg.slot >> 1) + 1];
                        const V3 c = (seg.slot & 1) ? v3(A.y, A.w, B.y) : v3(A.x, A.z, B.x);
                        n = normalize(pos - c);                              // sceneobjects.h:59-62
                        mat = 2 * seg.id;
                    } else {
                        const RtPlaneDev p = sc.planes[-1 - seg.slot];
                        n = v3(p.nx, p.ny, p.nz);                            // sceneobjects.h:117-120
                        mat = 2 * seg.id + (checker_white(pos) ? 0 : 1);     // sceneobjects.h:106-115
                    }
                    refl = normalize(reflect(seg.d, n));                     // main.cpp:100
                    if constexpr (!FAST) {
                        rec_pos[nlev][0] = pos.x; rec_pos[nlev][1] = pos.y; rec_pos[nlev][2] = pos.z;
                        rec_n[nlev][0] = n.x; rec_n[nlev][1] = n.y; rec_n[nlev][2] = n.z;
                        rec_r[nlev][0] = refl.x; rec_r[nlev][1] = refl.y; rec_r[nlev][2] = refl.z;
                        rec_mat[nlev] = mat;
                    }
                    want_refl = depth < fr.max_depth && sc.mats[mat].reflectance > 0.0f;   // main.cpp:103
                    if (want_refl) n_refl++;
                    n_shadow += sc.n_emit;
                }
                // enqueue: one LDS atomic per wave and class, lanes take consecutive slots
                if constexpr (COMPACT) {
                const unsigned long long m_refl = __ballot(alive && want_refl), m_plain = __ballot(alive && !want_refl);
                const unsigned long long below = (1ull << lane) - 1ull;
                int base_refl = 0, base_plain = 0;
                if (lane == 0) {
                    if (m_refl) base_refl = atomicAdd(&q_count[0], __popcll(m_refl));
                    if (m_plain) base_plain = atomicAdd(&q_count[1], __popcll(m_plain));
                }
                base_refl = __shfl(base_refl, 0, 64);
                base_plain = __shfl(base_plain, 0, 64);
                if (alive) {
                    my_slot = want_refl ? base_refl + __popcll(m_refl & below)
                                        : kBlock - 1 - (base_plain + __popcll(m_plain & below));
                    queue[0 * kBlock + my_slot] = __float_as_uint(pos.x);
                    queue[1 * kBlock + my_slot] = __float_as_uint(pos.y);
                    queue[2 * kBlock + my_slot] = __float_as_uint(pos.z);
                    queue[3 * kBlock + my_slot] = __float_as_uint(refl.x);
                    queue[4 * kBlock + my_slot] = __float_as_uint(refl.y);
                    queue[5 * kBlock + my_slot] = __float_as_uint(refl.z);
                }
                }
            }
            int n_front = 0, n_total = 0;
            if constexpr (COMPACT) {
                __syncthreads();
                n_front = q_count[0];
                n_total = n_front + q_count[1];
                if (n_total == 0) break;                                      // uniform: no live path in the workgroup
            }

            // ---- worker phase: thread t sweeps request t (its own hit point when not compacting) ----
            unsigned long long lit = 0, alt = 0;
            Segment next;
            next.dist = 0.0f; next.slot = INT_MIN; next.id = -1;
            bool next_hit = false;
            {
                const int t = threadIdx.x;
                const bool w_on = COMPACT ? t < n_total : alive;
                const bool w_refl = COMPACT ? t < n_front : want_refl;
                const int w_slot = w_refl ? t : kBlock - 1 - (t - n_front);
                V3 w_pos = pos, w_dir = refl;
                if (COMPACT && w_on) {
                    w_pos = v3(__uint_as_float(queue[0 * kBlock + w_slot]), __uint_as_float(queue[1 * kBlock + w_slot]),
                               __uint_as_float(queue[2 * kBlock + w_slot]));
                    w_dir = v3(__uint_as_float(queue[3 * kBlock + w_slot]), __uint_as_float(queue[4 * kBlock + w_slot]),
                               __uint_as_float(queue[5 * kBlock + w_slot]));
                }
                // chunked sweeps contain workgroup barriers, so every wave must take the same path there
                const bool any_refl = COMPACT ? (CHUNKED ? n_front > 0 : wave * 64 < n_front) : block_any<CHUNKED>(want_refl);
                int e0 = 0;
                // a wave with no request does nothing (chunked sweeps hold workgroup barriers: all waves go)
                if (!CHUNKED && !__any(w_on)) e0 = sc.n_emit;
                else if (any_refl) {
                    const int ne = min(sc.n_emit, kMaxK - 1);
                    trace_group_k<true, CHUNKED, GROUPED>(1 + ne, sc, lds, cap, w_on, w_pos, w_dir, w_refl, 0, lit, alt, next, next_hit, cnt);
                    e0 = ne;
                }
                while (e0 < sc.n_emit) {
                    const int ne = min(sc.n_emit - e0, kMaxK);
                    trace_group_k<false, CHUNKED, GROUPED>(ne, sc, lds, cap, w_on, w_pos, w_dir, false, e0, lit, alt, next, next_hit, cnt);
                    e0 += ne;
                }
                if (COMPACT && w_on) {
                    queue[0 * kBlock + w_slot] = (unsigned)lit;
                    queue[1 * kBlock + w_slot] = (unsigned)(lit >> 32);
                    queue[2 * kBlock + w_slot] = (unsigned)alt;
                    queue[3 * kBlock + w_slot] = (unsigned)(alt >> 32);
                    queue[4 * kBlock + w_slot] = __float_as_uint(next.dist);
                    queue[5 * kBlock + w_slot] = (unsigned)(next_hit ? next.slot : INT_MIN);
                    queue[6 * kBlock + w_slot] = (unsigned)next.id;
                }
            }
            if constexpr (COMPACT) {
                __syncthreads();
                if (threadIdx.x == 0) q_count[0] = q_count[1] = 0;
                if (alive) {         // home phase: take the result back
                    lit = (unsigned long long)queue[0 * kBlock + my_slot] | ((unsigned long long)queue[1 * kBlock + my_slot] << 32);
                    alt = (unsigned long long)queue[2 * kBlock + my_slot] | ((unsigned long long)queue[3 * kBlock + my_slot] << 32);
                    next.dist = __uint_as_float(queue[4 * kBlock + my_slot]);
                    next.slot = (int)queue[5 * kBlock + my_slot];
                    next.id = (int)queue[6 * kBlock + my_slot];
                }
            } else if (!next_hit) {
                next.slot = INT_MIN;
            }
            if (alive) {
                if constexpr (FAST) {
                    const rt_material m = sc.mats[mat];
                    V3 local = add_lights<STATS, true>(sc, v3(0.0f, 0.0f, 0.0f), pos, n, refl, m, lit, alt, n_lit);
                    local = local * (1.f - m.reflectance) + v3(m.emissive[0], m.emissive[1], m.emissive[2]);
                    acc = acc + local * tp;
                    tp = tp * (m.reflectance * (1.f - m.reflectance));
                } else {
                    rec_lit[nlev] = lit;
                    rec_alt[nlev] = alt;
                    nlev++;
                }
                if (want_refl) {
                    const int hit_slot = next.slot;
                    seg.o = pos + refl * 0.001f;                             // main.cpp:108
                    seg.d = refl;
                    seg.dist = next.dist;
                    seg.slot = hit_slot;
                    seg.id = next.id;
                    alive = hit_slot != INT_MIN;
                    miss_tail = !alive;
                    if (FAST && !alive) acc = acc + v3(0.1f, 0.1f, 0.1f) * tp;   // main.cpp:92 through the chain
                } else {
                    alive = false;
                }
            }
        }

        // ---- backward: shade deepest segment first (main.cpp:109-187) ----
        V3 child = v3(0.1f, 0.1f, 0.1f);           // what a missing ray returns (main.cpp:92)
        bool has_child = miss_tail;
        if constexpr (FAST) child = acc;
        for (int lev = nlev - 1; lev >= 0; lev--) {
            const V3 pos = v3(rec_pos[lev][0], rec_pos[lev][1], rec_pos[lev][2]);
            const V3 n = v3(rec_n[lev][0], rec_n[lev][1], rec_n[lev][2]);
            const V3 refl = v3(rec_r[lev][0], rec_r[lev][1], rec_r[lev][2]);
            const rt_material m = sc.mats[rec_mat[lev]];
            const unsigned long long lit = rec_lit[lev], alt = rec_alt[lev];
            V3 out = v3(0.0f, 0.0f, 0.0f);
            if (has_child) out = child * m.reflectance;                      // main.cpp:109
            out = add_lights<STATS, false>(sc, out, pos, n, refl, m, lit, alt, n_lit);
            out = out * (1.f - m.reflectance);                               // main.cpp:185
            out = out + v3(m.emissive[0], m.emissive[1], m.emissive[2]);     // main.cpp:186
            child = out;
            has_child = true;
        }
        if (in_image) color = color + child;                                 // main.cpp:208
    }

    {
        const size_t idx = (size_t)yr * (size_t)fr.width + (size_t)x;
        float rays = (float)((unsigned)(s_end - s_begin) + n_refl + n_shadow);
        // RT_OUT_COST: what the pixel's rays cost instead (this lane did all its sweeps unless the
        // workgroup compacts its rays, where the counters belong to other pixels)
        // (tests executed, exact tests at 32 filter tests each, and 300 per ray for its set-up, plane
        // tests and shading: weights read off the instruction counts of DESIGN.md section 4)
        if (STATS && !COMPACT && fr.cost_w) rays = (float)cnt.tests + (float)cnt.gtests + 32.0f * (float)cnt.cands + 300.0f * rays;
        if (ls) {
            // samples in lanes: main.cpp:203-209's `color +=` over the pixel's lanes, in sample order
            const int base = lane - lane_sample;
            V3 sum = v3(0.0f, 0.0f, 0.0f);
            float all_rays = 0.0f;
            for (int t = 0; t < ls; t++) {
                sum = sum + v3(__shfl(color.x, base + t, 64), __shfl(color.y, base + t, 64), __shfl(color.z, base + t, 64));
                all_rays += __shfl(rays, base + t, 64);
            }
            if (in_image && lane_sample == 0) write_pixel(fr, idx, sum, all_rays);
        } else if (in_image) {
            if (split) fr.sample_buf[(size_t)s_begin * ((size_t)fr.width * (size_t)rows) + idx] = make_float4(color.x, color.y, color.z, rays);
            else write_pixel(fr, idx, color, rays);
        }
    }

    if (STATS) {
        unsigned v[4] = {n_refl, n_shadow, n_hits, n_lit};
        unsigned long long *dst[4] = {&stats->reflection, &stats->shadow, &stats->hits, &stats->lit};
        for (int k = 0; k < 4; k++) {
            unsigned t = in_image ? v[k] : 0u;
            for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
            if (lane == 0 && t) atomicAdd(dst[k], (unsigned long long)t);
        }
        // executed tests are counted by the lane that swept (under compaction not the pixel's own)
        unsigned long long w[3] = {cnt.tests, cnt.gtests, cnt.cands};
        unsigned long long *wdst[3] = {&stats->sphere_tests_run, &stats->group_tests_run, &stats->exact_tests_run};
        for (int k = 0; k < 3; k++) {
            unsigned long long t = w[k];
            for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
            if (lane == 0 && t) atomicAdd(wdst[k], t);
        }
    }
}

} // namespace

// The sample split pays wherever sweeps are long (grouped sweep: 8-way bands 4.6 -> 2.4 ms; flat sweep
// of the 1028-sphere scene 33.5 -> 32.2 ms, its four waves then tracing four samples of one tile); on
// scenes of a few spheres the scratch plane's traffic costs more than the tail it removes.
// Samples in lanes: the grouped sweep (single-wave workgroups), when spp is a power of two that fits a
// wave.  (Measured: grid-1024 4K 4 spp 12.05 ms against 12.6-13.0 with the sample split and 13.3 with a
// loop over samples; the flat sweep of the 6-object scene loses 10-20 % to it, so it keeps its loop.)
bool rt_sweep_can_put_samples_in_lanes(const RtSceneDev &scene, const RtFrameDev &frame, bool grouped)
{
    (void)scene;
    return grouped && frame.spp > 1 && frame.spp <= 64 && (frame.spp & (frame.spp - 1)) == 0;
}

bool rt_sweep_can_split_samples(const RtSceneDev &scene, const RtFrameDev &frame, bool grouped)
{
    return frame.spp > 1 && (grouped || scene.n_pairs >= 24);
}

hipError_t rt_launch_resolve_samples(const RtFrameDev &frame, hipStream_t stream)
{
    const size_t n_px = (size_t)frame.width * (size_t)(frame.row_end - frame.row_begin);
    if (n_px == 0 || n_px > 0xffffffffull) return n_px ? hipErrorInvalidValue : hipSuccess;
    hipLaunchKernelGGL(resolve_samples_kernel, dim3((unsigned)((n_px + 255) / 256)), dim3(256), 0, stream, frame, (unsigned)n_px);
    return hipGetLastError();
}

hipError_t rt_launch_trace_sweep(const RtSceneDev &scene, const rt_camera &cam, const RtFrameDev &frame,
                                 const RtSamples &samples, RtStatsDev *stats, bool fast, bool grouped,
                                 hipStream_t stream)
{
    const int rows = frame.row_end - frame.row_begin;
    const int tiles = ((frame.width + frame.tile_w - 1) / frame.tile_w) * ((rows + frame.tile_h - 1) / frame.tile_h);
    if (tiles <= 0) return hipSuccess;
    const int n_pairs = scene.n_pairs;
    const bool chunked = !grouped && frame.chunk < n_pairs;
    const int cap = frame.chunk > 0 ? frame.chunk : 1;
    // compaction costs three workgroup barriers per bounce level: worth it once sweeps are long
    // (and not in the grouped sweep at all: mixing the rays of four tiles widens the set of groups a
    // wave has to visit, which costs more than the idle lanes: 15.8 -> 14.2 ms on grid-1024)
    const bool compact = n_pairs >= 24 && !grouped;
    // single-wave workgroups wherever waves share nothing that matters (stage() strides by kBlock: fewer than 64 pairs)
    const bool one_wave = grouped || (!compact && !chunked);
    // LDS: flat = pairs | camera table, which the request queue aliases when compacting; grouped = none
    const int second = compact || chunked ? (2 * cap > kQueueFloat4 ? 2 * cap : kQueueFloat4) : 2 * cap;
    const size_t lds_bytes = grouped ? 0 : (size_t)(2 * cap + second) * sizeof(float4);
    const int units = frame.sample_buf ? tiles * frame.spp : tiles;          // sample split: one wave per (tile, sample)
    const dim3 grid(one_wave ? units : (units + kWavesPerBlock - 1) / kWavesPerBlock), block(one_wave ? 64 : kBlock);
#define RT_LAUNCH_G(C, P, S, L, F, G)                                                                                   \
    do {                                                                                                            \
        if (lds_bytes > 48 * 1024) {                                                                                \
            hipError_t e_ = hipFuncSetAttribute((const void *)trace_sweep_kernel<C, P, S, L, F, G>,                     \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);        \
            if (e_ != hipSuccess) return e_;                                                                        \
        }                                                                                                           \
        hipLaunchKernelGGL((trace_sweep_kernel<C, P, S, L, F, G>), grid, block, lds_bytes, stream, scene, cam, frame, samples, stats); \
    } while (0)
#define RT_LAUNCH(C, P, S, G)                                             \
    do {                                                                  \
        if (fast) RT_LAUNCH_G(C, P, S, 1, true, G);                       \
        else if (frame.max_depth < 6) RT_LAUNCH_G(C, P, S, 6, false, G);  \
        else RT_LAUNCH_G(C, P, S, RT_MAX_LEVELS, false, G);               \
    } while (0)
#define RT_LAUNCH_S(C, P, G)                                              \
    do {                                                                  \
        if (stats) RT_LAUNCH(C, P, true, G); else RT_LAUNCH(C, P, false, G); \
    } while (0)
    if (chunked) {
        RT_LAUNCH_S(true, true, false);
    } else if (grouped) {
        RT_LAUNCH_S(false, false, true);
    } else {
        if (compact) RT_LAUNCH_S(false, true, false); else RT_LAUNCH_S(false, false, false);
    }
#undef RT_LAUNCH_S
#undef RT_LAUNCH
#undef RT_LAUNCH_G
    return hipGetLastError();
}
