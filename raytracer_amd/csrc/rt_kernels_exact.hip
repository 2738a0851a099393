// rt_kernels_exact.hip -- RT_MODE_EXACT trace kernel for gfx950.  Compile with -ffp-contract=off.
//
// One wave64 owns one 8x8 pixel tile; a 256-thread workgroup owns four tiles and ONE LDS copy of
// the sphere geometry (float4 {cx,cy,cz,r^2}, 16 B per sphere), which every lane reads as a
// broadcast (identical LDS address across the wave: no bank conflicts).  Scenes larger than the
// LDS budget are streamed through LDS in chunks with workgroup barriers (CHUNKED).
//
// The reference's recursion (TraceRay, raytracer_cpp/main.cpp:84-188) is unrolled without a call
// stack in two passes per sample: the hit chain (which never depends on lighting) is traced forward
// and one small record per path segment is kept; shading then runs deepest-first so that every
// float operation happens in the reference's order:
//     out = child * refl ; out += light_0 ; out += light_1 ... ; out *= (1 - refl) ; out += emissive
// Emitters are the objects with non-zero emissive only (bit-identical to the reference's loop over
// all objects).
#include "rt_device.h"
#include "rt_math.h"

#include <float.h>

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kBlock = 64 * kWavesPerBlock;

struct Ray {
    V3 o, d, dn;   // origin, direction as the reference passes it, normalize(direction)
};

// Sphere::Intersects -> glm::intersectRaySphere, sceneobjects.h:69-73 / gtx/intersect.inl:105-123
__device__ __forceinline__ bool sphere_hit(float4 g, const Ray &r, float &dist)
{
    V3 diff = v3(g.x, g.y, g.z) - r.o;
    float t0 = dot(diff, r.dn);
    float dsq = dot(diff, diff) - t0 * t0;
    if (dsq > g.w) return false;
    float t1 = sqrtf(g.w - dsq);
    dist = t0 > t1 + FLT_EPSILON ? t0 - t1 : t0 + t1;
    return dist > FLT_EPSILON;
}

// TiledPlane::Intersects -> glm::intersectRayPlane, sceneobjects.h:127-130 / gtx/intersect.inl:7-24
__device__ __forceinline__ bool plane_hit(const RtPlaneDev &p, const Ray &r, float &dist)
{
    V3 n = v3(p.nx, p.ny, p.nz);
    float d = dot(r.d, n);
    if (d < FLT_EPSILON) {
        dist = dot(v3(p.ox, p.oy, p.oz) - r.o, n) / d;
        return true;
    }
    return false;
}

template <bool CHUNKED>
__device__ __forceinline__ void stage(const RtSceneDev &sc, float4 *lds, int base, int count)
{
    if (CHUNKED) __syncthreads();              // previous chunk fully consumed
    for (int i = threadIdx.x; i < count; i += kBlock) lds[i] = sc.geom[base + i];
    __syncthreads();
}

// FindNearestObject, main.cpp:65-82.  hit_slot: >= 0 sphere slot, < 0: -1 - plane slot.
// Strict '>' with the list swept in order means the lowest list index wins ties; spheres are kept
// in list order, planes are merged afterwards with an explicit (distance, list index) comparison.
template <bool CHUNKED>
__device__ __forceinline__ void find_nearest(const RtSceneDev &sc, const float4 *lds, int chunk, bool active,
                                             const Ray &r, float &best, int &hit_slot, int &hit_id)
{
    best = FLT_MAX;
    hit_slot = INT_MIN;
    for (int base = 0; base < sc.n_spheres; base += chunk) {
        int count = min(chunk, sc.n_spheres - base);
        if (CHUNKED) stage<true>(sc, const_cast<float4 *>(lds), base, count);
        if (active) {
#pragma unroll 4
            for (int i = 0; i < count; i++) {
                float d;
                if (sphere_hit(lds[i], r, d) && best > d) {
                    best = d;
                    hit_slot = base + i;
                }
            }
        }
    }
    hit_id = -1;
    if (active) {
        if (hit_slot >= 0) hit_id = sc.sph_id[hit_slot];
        for (int k = 0; k < sc.n_planes; k++) {
            RtPlaneDev p = sc.planes[k];
            float d;
            if (plane_hit(p, r, d)) {
                if (best > d || (best == d && hit_id >= 0 && p.id < hit_id)) {
                    best = d;
                    hit_slot = -1 - k;
                    hit_id = p.id;
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

struct Counters {
    unsigned reflection = 0, shadow = 0, hits = 0, lit = 0;
};

template <bool CHUNKED, bool STATS>
__global__ __launch_bounds__(kBlock) void trace_exact_kernel(RtSceneDev sc, rt_camera cam, RtFrameDev fr,
                                                             RtSamples smp, RtStatsDev *stats)
{
    extern __shared__ float4 lds_geom[];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int tiles_x = (fr.width + 7) >> 3;
    const int rows = fr.row_end - fr.row_begin;
    const int tiles_y = (rows + 7) >> 3;
    const int tile = blockIdx.x * kWavesPerBlock + wave;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int x = tx * 8 + (lane & 7);
    const int yr = ty * 8 + (lane >> 3);              // row inside the band
    const int y = fr.row_begin + yr;
    const bool in_image = (tile < tiles_x * tiles_y) && x < fr.width && yr < rows;

    if (!CHUNKED) stage<false>(sc, lds_geom, 0, sc.n_spheres);

    const V3 cam_pos = v3(cam.position[0], cam.position[1], cam.position[2]);
    const V3 cam_view = v3(cam.view_direction[0], cam.view_direction[1], cam.view_direction[2]);
    const V3 cam_right = v3(cam.right[0], cam.right[1], cam.right[2]);
    const V3 cam_up = v3(cam.up[0], cam.up[1], cam.up[2]);

    // per-segment records (deepest-first shading needs them); runtime-indexed -> private memory
    float rec_pos[RT_MAX_LEVELS][3], rec_n[RT_MAX_LEVELS][3], rec_r[RT_MAX_LEVELS][3];
    int rec_mat[RT_MAX_LEVELS];

    Counters cnt;
    V3 color = v3(0.0f, 0.0f, 0.0f);

    for (int s = 0; s < fr.spp; s++) {                                       // main.cpp:203
        // Camera::GetWorldRay, camera.h:48-62
        Ray ray;
        {
            float sx = (float)x + smp.xy[2 * s], sy = (float)y + smp.xy[2 * s + 1];   // main.cpp:205
            float px = ((sx * 2.0f) / cam.image_width) - 1.0f;
            float py = ((sy * 2.0f) / cam.image_height) - 1.0f;
            V3 d = cam_view;
            d = d + cam_right * (cam.half_angle * cam.aspect_ratio * px);
            d = d - cam_up * (cam.half_angle * py);
            ray.d = normalize(d);
            ray.o = cam_pos;
        }

        // ---- forward: the hit chain (main.cpp:88-110 without the lighting) ----
        int nlev = 0;
        bool miss_tail = false;
        bool alive = in_image;
        for (int depth = 0; depth <= fr.max_depth; depth++) {
            if (!block_any<CHUNKED>(alive)) break;
            ray.dn = normalize(ray.d);                                       // sceneobjects.h:71
            float dist; int slot, id;
            find_nearest<CHUNKED>(sc, lds_geom, fr.chunk, alive, ray, dist, slot, id);
            if (alive) {
                if (slot == INT_MIN) {                                       // main.cpp:90-93
                    miss_tail = true;
                    alive = false;
                } else {
                    if (STATS) cnt.hits++;
                    V3 pos = ray.o + ray.d * dist;                           // main.cpp:94
                    V3 n; int mat;
                    if (slot >= 0) {
                        float4 g = sc.geom[slot];
                        n = normalize(pos - v3(g.x, g.y, g.z));              // sceneobjects.h:59-62
                        mat = 2 * id;
                    } else {
                        RtPlaneDev p = sc.planes[-1 - slot];
                        n = v3(p.nx, p.ny, p.nz);                            // sceneobjects.h:117-120
                        mat = 2 * id + (checker_white(pos) ? 0 : 1);         // sceneobjects.h:106-115
                    }
                    V3 refl = normalize(reflect(ray.d, n));                  // main.cpp:100
                    rec_pos[nlev][0] = pos.x; rec_pos[nlev][1] = pos.y; rec_pos[nlev][2] = pos.z;
                    rec_n[nlev][0] = n.x; rec_n[nlev][1] = n.y; rec_n[nlev][2] = n.z;
                    rec_r[nlev][0] = refl.x; rec_r[nlev][1] = refl.y; rec_r[nlev][2] = refl.z;
                    rec_mat[nlev] = mat;
                    nlev++;
                    float reflectance = sc.mats[mat].reflectance;
                    if (depth < fr.max_depth && reflectance > 0.0f) {        // main.cpp:103
                        cnt.reflection++;
                        ray.o = pos + refl * 0.001f;                         // main.cpp:108
                        ray.d = refl;
                    } else {
                        alive = false;
                    }
                }
            }
        }

        // ---- backward: shade deepest segment first (main.cpp:109-187) ----
        V3 child = v3(0.1f, 0.1f, 0.1f);       // what the deepest segment's reflection returned
        bool has_child = miss_tail;
        for (int lev = fr.max_depth; lev >= 0; lev--) {
            const bool on = in_image && lev < nlev;
            if (!block_any<CHUNKED>(on)) continue;
            V3 pos = v3(0, 0, 0), n = pos, refl = pos, out = pos;
            rt_material m = {};
            if (on) {
                pos = v3(rec_pos[lev][0], rec_pos[lev][1], rec_pos[lev][2]);
                n = v3(rec_n[lev][0], rec_n[lev][1], rec_n[lev][2]);
                refl = v3(rec_r[lev][0], rec_r[lev][1], rec_r[lev][2]);
                m = sc.mats[rec_mat[lev]];
                if (has_child) out = child * m.reflectance;                  // main.cpp:109
            }
            for (int e = 0; e < sc.n_emit; e++) {                            // main.cpp:112
                RtEmitterDev em = sc.emit[e];
                Ray sh;
                V3 edir = normalize(v3(em.cx, em.cy, em.cz) - pos);          // GetRayFrom, main.cpp:114
                sh.o = pos + edir * 0.001f;                                  // main.cpp:121
                sh.d = edir;
                sh.dn = normalize(edir);
                float dist; int slot, id;
                find_nearest<CHUNKED>(sc, lds_geom, fr.chunk, on, sh, dist, slot, id);
                if (on) {
                    cnt.shadow++;
                    // lit iff the nearest object along the shadow ray (lowest index on ties) is
                    // the emitter itself and its material there is emissive (main.cpp:119-157)
                    bool lit = (slot != INT_MIN) && id == em.id;
                    int emat = 2 * em.id;
                    if (lit && em.is_plane) {
                        emat += checker_white(pos + edir * dist) ? 0 : 1;    // main.cpp:130
                        rt_material pm = sc.mats[emat];
                        lit = !(pm.emissive[0] == 0.0f && pm.emissive[1] == 0.0f && pm.emissive[2] == 0.0f);
                    }
                    if (lit) {
                        float diffuse = dot(n, edir);                        // main.cpp:162
                        float spec = 0.0f;
                        if (diffuse > 0.0f) {
                            spec = dot(refl, edir);                          // main.cpp:166
                            if (spec > 0.0f) {
                                spec = pow10_like_libm(spec);                // main.cpp:169
                                spec = 0.0f < spec ? spec : 0.0f;            // main.cpp:170
                            } else {
                                spec = 0.0f;
                            }
                        } else {
                            diffuse = 0.0f;
                        }
                        if (STATS) cnt.lit++;
                        const float *ee = sc.mats[emat].emissive;
                        V3 l = (v3(ee[0], ee[1], ee[2]) * v3(m.albedo[0], m.albedo[1], m.albedo[2])) * diffuse +
                               v3(m.specular[0], m.specular[1], m.specular[2]) * spec;   // main.cpp:183
                        out = out + l;
                    }
                }
            }
            if (on) {
                out = out * (1.f - m.reflectance);                           // main.cpp:185
                out = out + v3(m.emissive[0], m.emissive[1], m.emissive[2]); // main.cpp:186
                child = out;
                has_child = true;
            }
        }
        if (in_image) color = color + child;                                 // main.cpp:208
    }

    if (in_image) {
        color = color * fr.inv_spp;                                          // main.cpp:210
        size_t idx = (size_t)yr * (size_t)fr.width + (size_t)x;
        if (fr.f32x4) fr.f32x4[idx] = make_float4(color.x, color.y, color.z, (float)(fr.spp + cnt.reflection + cnt.shadow));
        if (fr.rgba8)
            fr.rgba8[idx] = to_u8(color.x) | (to_u8(color.y) << 8) | (to_u8(color.z) << 16) | 0xff000000u;
    }

    if (STATS) {
        unsigned v[4] = {cnt.reflection, cnt.shadow, cnt.hits, cnt.lit};
        unsigned long long *dst[4] = {&stats->reflection, &stats->shadow, &stats->hits, &stats->lit};
        for (int k = 0; k < 4; k++) {
            unsigned t = v[k];
            for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
            if (lane == 0 && t) atomicAdd(dst[k], (unsigned long long)t);
        }
    }
}

} // namespace

hipError_t rt_launch_trace_exact(const RtSceneDev &scene, const rt_camera &cam, const RtFrameDev &frame,
                                 const RtSamples &samples, RtStatsDev *stats, hipStream_t stream)
{
    const int rows = frame.row_end - frame.row_begin;
    const int tiles = ((frame.width + 7) / 8) * ((rows + 7) / 8);
    if (tiles <= 0) return hipSuccess;
    const dim3 grid((tiles + kWavesPerBlock - 1) / kWavesPerBlock), block(kBlock);
    const bool chunked = frame.chunk < scene.n_spheres;
    const size_t lds = (size_t)(frame.chunk > 0 ? frame.chunk : 1) * sizeof(float4);
    if (chunked) {
        if (stats) hipLaunchKernelGGL((trace_exact_kernel<true, true>), grid, block, lds, stream, scene, cam, frame, samples, stats);
        else hipLaunchKernelGGL((trace_exact_kernel<true, false>), grid, block, lds, stream, scene, cam, frame, samples, stats);
    } else {
        if (stats) hipLaunchKernelGGL((trace_exact_kernel<false, true>), grid, block, lds, stream, scene, cam, frame, samples, stats);
        else hipLaunchKernelGGL((trace_exact_kernel<false, false>), grid, block, lds, stream, scene, cam, frame, samples, stats);
    }
    return hipGetLastError();
}
