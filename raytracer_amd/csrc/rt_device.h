// rt_device.h -- device-side scene layout shared by the C-ABI implementation (rt_api.hip) and the
// trace kernels (rt_kernels_*.hip).
//
// HBM layout of an uploaded scene (all arrays read-only during a render):
//   pairs  float4[2*ceil(n_spheres/2)]  spheres in LIST ORDER, two per 32-byte record:
//                              {cx0,cx1,cy0,cy1} {cz0,cz1,-r0*r0,-r1*r1}; the only array the
//                              intersection sweep streams (16 B per sphere, staged to LDS); an odd
//                              tail is padded with a sphere nothing can hit (-r*r = +FLT_MAX)
//   geom   float4[n_spheres]   {cx, cy, cz, r*r}: the same data unpaired (A/B kernel, hit lookups)
//   sph_id int32[n_spheres]    list index of sphere i (ties and light order are by list index)
//   planes RtPlaneDev[n_planes]
//   mats   rt_material[2*n]    [2*k] = object k's material (plane: white), [2*k+1] = plane black
//   emit   RtEmitterDev[n_emit] objects with a non-zero emissive material, in list order
//
// Grouped layout (RT_SWEEP_GROUPED): a second copy of pairs/sph_id in which the spheres are sorted
// along a Morton curve and cut into GROUPS of at most 8 (4 pair records each, short groups padded
// with unhittable spheres), plus
//   heads  float4[2*ceil(n_groups/2)]  bounds of the groups, two per 32-byte record:
//                              {gx0,gx1,gy0,gy1} {gz0,gz1,R0,R1} (an odd tail is zeros; the index check drops it).
//                              A bound is the ball {x : |diag(sigma)(x - g)| <= R}: an ellipsoid whose shape
//                              (RtSceneDev::metric) the upload picks per scene -- flattened along the axis the
//                              groups are thin in (spheres on a floor), round when they are not
//   supers float4[2*ceil(n_supers/2)]  the same one level up: bounding spheres of RT_SUPER_GROUPS
//                              consecutive groups
// A ray that passes a group's bounding sphere at a distance skips the group's 8 spheres.  The grouped
// arrays are read with wave-uniform scalar loads straight from HBM/L2: no LDS staging, no size limit.  Slot order is
// no longer list order there, so distance ties are broken by sph_id explicitly.
#ifndef RT_DEVICE_H
#define RT_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_api.h"

#define RT_MAX_LEVELS 16      // path segments kept per sample: max_depth <= RT_MAX_LEVELS - 1
#define RT_MAX_SPP 64
#define RT_MAX_PLANES 16
#define RT_MAX_EMITTERS 64
#define RT_GROUP_PAIRS 4      // pair records per group (8 spheres)
#define RT_SUPER_GROUPS 8      // groups per super-group (64 spheres)
#define RT_N_REGIONS 16
#ifndef RT_SORTED_CELL_BITS
#define RT_SORTED_CELL_BITS 15 // RT_SWEEP_SORTED: the scene is cut into 2^15 cells (measured: 2^18 no better, and its counters take 100 us to scan per level)
#endif
#define RT_QUEUE_BYTES (7 * 256 * 4)   // LDS request queue of the sweep kernel: 7 fields x 256 threads

struct RtPlaneDev {
    float ox, oy, oz;
    float nx, ny, nz;
    int32_t id;           // list index
    int32_t pad;
};

struct RtEmitterDev {
    float cx, cy, cz;     // Sphere::center / Plane::origin: the point GetRayFrom aims at
    int32_t id;           // list index
    int32_t is_plane;
    int32_t sphere_slot;  // index into geom[] when the emitter is a sphere, else -1
    int32_t order;        // emit[]: entry i names the emitter whose shadow ray is swept i-th (emitters that lie in similar
                          // directions share a sweep; light summation stays in list order).  emit_sweep[]: the emitters
                          // in that sweep order, `order` = the record's own index in emit[] (its bit in the lit mask)
    float radius;         // |radius| of a sphere emitter, +inf for a plane (grouped sweep: decided shadow rays)
};

struct RtSceneDev {
    const float4 *pairs;
    const float4 *geom;
    const int32_t *sph_id;
    const RtPlaneDev *planes;
    const rt_material *mats;
    const RtEmitterDev *emit;
    const RtEmitterDev *emit_sweep;   // the same records in sweep order (one scalar load per shadow ray instead of two dependent ones)
    int32_t n_spheres, n_planes, n_objects, n_emit;
    int32_t n_pairs;      // pair records in `pairs`: ceil(n_spheres/2) flat, 8*n_groups grouped
    const float4 *heads;  // grouped layout only: bounding spheres of the groups ...
    const float4 *supers; // ... and of the super-groups (RT_SUPER_GROUPS consecutive groups each)
    int32_t n_groups, n_supers;
    int32_t n_tail_pairs; // grouped layout: pair records after the last group's that EVERY sweep tests, bounds unasked -- the
                          // emissive spheres of a scene with at most 8 of them (every shadow ray is aimed at one)
    // the bounds are balls of the norm |x|_S = |diag(sigma) x| (ellipsoids, one shape per scene; sigma >= 1):
    float metric[3];      // sigma_a^2
    float slack_scale;    // max sigma^2: what a world-space allowance grows by in that norm
    float sigma[3];
    // the same bounds one per float4 {g, R}: the beam test puts one bound in each lane
    const float4 *heads_flat;
    const float4 *supers_flat;
    const float4 *supers2;    // paired records like `supers`, one level up: RT_SUPER_GROUPS consecutive super-groups each
    int32_t n_supers2;
    float beam_max_rho, beam_max_chord2;   // a wave's rays go through the beam test while their origins lie within this
                          // radius (in the bounds' norm) and each class's directions within this chord^2 of one another
    float far_bound;      // a lane whose filter allowance (2^-19 |pos|^2) reaches this is so far from the origin that the
                          // filter keeps spheres a good part of the scene's radius off its rays: its rays are tested
                          // against every sphere, one ray at a time across the wave (dense_sweep); +inf: never
    float box_lo[3], box_hi[3];   // the box around the spheres under bounds (centre -/+ |radius|; -inf/+inf: unknown): a ray that
                          // has left it has nothing more to hit among them (distance culling, rt_sweep_kernel.inc)
};

struct RtFrameDev {
    uint32_t *rgba8;      // band-relative: row (y - row_begin)
    float4 *f32x4;
    int32_t width, height;
    int32_t row_begin, row_end;
    int32_t max_depth, spp;
    float inv_spp;
    int32_t chunk;        // LDS capacity per pass: sphere PAIRS (sweep kernel) / spheres (simple kernel)
    int32_t lens_on;      // thin-lens ray generation (raytracer_win/camera.h:147-173), offsets in RtSamples::lens
    float focal_length;
    int32_t accumulate;   // running mean into f32x4 (raytracer_win/main.cpp:292-314)
    float k1, k2;         // currentSample and 1 / (currentSample + 1)
    // sample split: one wave per (tile, sample) writes that sample's radiance (w = its ray count) to
    // sample_buf[s * pixels + pixel]; rt_launch_resolve_samples then adds them up in sample order
    float4 *sample_buf;
    // samples in lanes (single-wave kernels, spp a power of two <= 64): a wave covers 64/spp pixels, one
    // lane per (pixel, sample); the samples of a pixel are summed in order across lanes.  tile_w x tile_h
    // is the pixel footprint (8x8 when off).
    int32_t lane_samples, tile_w, tile_h;
    int32_t strip_stride, strip_phase;   // rt_render_params.col_interleave/col_phase (stride 0 = off): `width` is the
                          // width of the output planes, whose column k is image column ((k / 8) * stride + phase) * 8 + k % 8
    int32_t first_tile_row;   // rt_render_params.first_row in tile rows of this band: launch order only
    int32_t cost_w;       // stats kernels: f32x4.w = work estimate instead of the ray count (RT_OUT_COST)
    // profiling builds only (-DRT_REGION_TIMERS, tools/region_profile.sh; RT_AMD_REGION_DUMP=1 makes the library pass
    // a buffer): RT_N_REGIONS sums of wave-cycles (s_memtime) by region of the trace kernel; NULL otherwise
    unsigned long long *region_cycles;
};

struct RtStatsDev {
    unsigned long long primary, reflection, shadow, hits, lit;
    unsigned long long sphere_tests_run, group_tests_run;   // (ray, sphere) filters / (ray, group) tests executed
    unsigned long long exact_tests_run;                     // (ray, sphere) candidates put through the exact test
};

struct RtSamples {
    float xy[2 * RT_MAX_SPP];
    float lens[2 * RT_MAX_SPP];   // lens-plane offsets (along right, along up) when RtFrameDev::lens_on
};

// RT_SWEEP_SORTED (rt_sweep_sorted.hip): the buffers one render's bounce levels pass through
struct RtSortedDev {
    float4 *rec;          // records of the path segments, 4 float4 each, level-major: (level * cap + index) * 4
    float4 *queue[2];     // entries queued for a level, 3 float4 each; queue[level & 1] feeds `level`
    unsigned *perm;       // sorted position -> entry of the level being swept
    unsigned *hist;       // [level][cell]: entries per cell, then (scan) where the cell's entries start
    float4 *farq;         // entries of the level being swept whose hit point is a far one (one wave each, see the kernels) ...
    float4 *farq0;        // ... level 0's apart: they are swept on a stream of their own while the levels go on
    unsigned *count;      // [level]: entries queued
    unsigned *count_far;  // [level]: of them, far ones (count + RT_MAX_LEVELS + 1)
    unsigned *count_kept; // [level]: segments followed by the wave that found them (records from the top of the level's)
    unsigned cap;         // paths of the frame = records per level at most
    int32_t far0_waves;   // waves of level 0's far launch (it shares the chip with the sorted levels)
    int32_t final_level;  // the last level that is sorted and launched by itself: its waves follow their paths to the end
    float cell_org[3], cell_inv[3];   // cell of a point p along axis a: (p[a] - cell_org[a]) * cell_inv[a], clamped to [0, 2^cell_bits[a])
    int32_t cell_bits[3];             // RT_SORTED_CELL_BITS shared between the axes so that cells come out roughly cubic
};
size_t rt_sorted_cells(void);
hipError_t rt_launch_trace_sorted(const RtSceneDev &scene, const rt_camera &cam, const RtFrameDev &frame, const RtSamples &samples,
                                  RtStatsDev *stats, bool fast, const RtSortedDev &sorted, hipStream_t stream, hipStream_t side,
                                  hipEvent_t ev_level0, hipEvent_t ev_side);

// kernel launchers
hipError_t rt_launch_trace_sweep(const RtSceneDev &scene, const rt_camera &cam, const RtFrameDev &frame,
                                 const RtSamples &samples, RtStatsDev *stats, bool fast, bool grouped,
                                 hipStream_t stream);
hipError_t rt_launch_resolve_samples(const RtFrameDev &frame, hipStream_t stream);
bool rt_sweep_can_put_samples_in_lanes(const RtSceneDev &scene, const RtFrameDev &frame, bool grouped);
bool rt_sweep_can_split_samples(const RtSceneDev &scene, const RtFrameDev &frame, bool grouped);
hipError_t rt_launch_trace_exact(const RtSceneDev &scene, const rt_camera &cam, const RtFrameDev &frame,
                                 const RtSamples &samples, RtStatsDev *stats, hipStream_t stream);

#endif
