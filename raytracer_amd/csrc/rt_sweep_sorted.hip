// rt_sweep_sorted.hip -- RT_SWEEP_SORTED: the grouped sweep with the bounce levels of TraceRay (main.cpp:84-188) run as
// separate launches, the hit points of each level re-sorted by where they lie before their rays are swept.
//
// Why.  In the one-launch kernel (rt_sweep_kernel.inc) a wave follows its 64 paths to the end.  After the first bounce
// off a scene of small spheres the wave's hit points are spread all over the scene: its rays form no beams, the bounds
// its lanes have to visit are the union over 64 unrelated origins, and most lanes have no path left -- on the
// 16k-sphere scene 70 % of the kernel's wave-cycles were such sweeps.  None of that is inherent: a ray's result does not
// depend on which lanes it shares a wave with (filter and bound tests are conservative, the exact test is the
// reference's arithmetic), so the segments of one level may be dealt to waves in any order.  Here they are dealt in
// order of the cell of a 2^15-cell grid over the scene (Morton order) their hit point lies in: a wave's 64 hit points then
// sit next to one another, every lane has work, the shadow rays towards an emitter form a beam again, and the bounds a
// wave visits are those of one neighbourhood.
//
// The pipeline, nothing read back by the host:
//   level 0   sorted_level_kernel<.., LEVEL0>: primary rays, tile by tile as in the one-launch kernel (camera rays are
//             coherent as they come), the hit point, the rays that leave it; writes the segment's RECORD (what the
//             shading pass needs: pos, normal, reflected direction, material, lit masks) and, when the reflection ray
//             hit something, queues an ENTRY for level 1 (the ray and what it hit) with the cell of its hit point and
//             its rank among that cell's entries (one atomic per wave and distinct cell on the cell's counter)
//   per level l = 1 .. max_depth:
//       sorted_scan_kernel    cell counters -> start of each cell's range (exclusive prefix sum, one workgroup)
//       sorted_perm_kernel    entry u goes to position start[cell] + rank: perm[position] = u
//       sorted_level_kernel   lane s takes entry perm[s]: hit point, record, sweeps, queue for level l + 1
//     (a fixed number of waves that stride over the sorted entries, bounded by the device-side count; nothing waits for
//     the host)
//   far segments (hit points so far out that every sphere is a candidate: dense_sweep) are not swept where they turn up
//             but queued, and sorted_level_kernel<.., FAR> gives each a wave of its own: level l's right behind level l's
//             launch; level 0's -- most of a frame's -- on a second stream beside the sorted levels, with a second launch
//             behind the last level for what is left (they take segments from a counter, and follow what their
//             reflection rays find themselves instead of queueing it)
//   sorted_shade_kernel       one thread per pixel: walks each sample's chain of records down, then shades it
//             deepest-first in the reference's operation order (main.cpp:109-186) -- or top-down with the throughput
//             accumulator (RT_MODE_FAST) -- adds the samples in order (main.cpp:203-210) and writes the pixel
//
// Same sweeps, same arithmetic per ray, same order of operations in the shading: the frame is the flat sweep's bit for
// bit (tests/test_gpu_cases.py::test_sorted_sweep_is_bit_identical).
//
// Memory (device, held by the context between renders of the same size): 64 B per record x paths x (max_depth + 1)
// levels, four entry buffers of 48 B x paths, 4 B x paths of permutation, 128 KB of cell counters per level; a 4K frame at
// 1 spp and MAX_DEPTH 5 takes 4.8 GB.  rt_render falls back to RT_SWEEP_GROUPED above RT_SORTED_MAX_BYTES.
#ifndef RT_DISTANCE_CULL
#define RT_DISTANCE_CULL 3      // a reflection ray swept on its own stops asking for bounds behind its hit, nearest bounds first (sweep_near_first())
#endif
#ifndef RT_DENSE_PRUNE
#define RT_DENSE_PRUNE 1
#endif
#ifndef RT_BOX_SKIP_WALK
#define RT_BOX_SKIP_WALK 3
#endif
#include "rt_sweep_kernel.inc"

namespace {

constexpr int kCellBitsTotal = RT_SORTED_CELL_BITS;        // shared between the axes (rt_api.hip: rt_upload_scene)
constexpr unsigned kCells = 1u << kCellBitsTotal;

struct RtSortedKernArgs {
    RtKernArgs base;                               // (first: fresh_args() reads it through the kernarg segment pointer)
    RtSortedDev so;
    int level;
};
typedef const RT_ARG RtSortedKernArgs *SortedArgsPtr;
typedef const RT_ARG RtSortedDev SortedArg;
__device__ __forceinline__ SortedArgsPtr fresh_sorted()
{
    SortedArgsPtr p = (SortedArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// record of one path segment: 4 float4
//   [0] pos.xyz, material index     [1] normal.xyz, child     [2] reflected direction.xyz, -     [3] lit (2 words), alt (2 words)
// child: the record of the segment the reflection ray found (an index into the next level's records), kNoRay when the
// segment sent no reflection ray (main.cpp:103), kRayMissed when it sent one that hit nothing (main.cpp:90-93).
// Level 0's record of a path whose primary ray missed has material kNoHit.
constexpr int kNoRay = -1, kRayMissed = -2, kNoHit = -1;

__device__ __forceinline__ unsigned cell_of(SortedArg &so, V3 p)
{
    // (min/max drop a NaN coordinate into cell 0: where an entry is sorted to changes nothing but its neighbours)
    const int bx = so.cell_bits[0], by = so.cell_bits[1], bz = so.cell_bits[2];
    const float fx = fminf(fmaxf((p.x - so.cell_org[0]) * so.cell_inv[0], 0.0f), (float)((1 << bx) - 1));
    const float fy = fminf(fmaxf((p.y - so.cell_org[1]) * so.cell_inv[1], 0.0f), (float)((1 << by) - 1));
    const float fz = fminf(fmaxf((p.z - so.cell_org[2]) * so.cell_inv[2], 0.0f), (float)((1 << bz) - 1));
    const unsigned ix = (unsigned)fx, iy = (unsigned)fy, iz = (unsigned)fz;
    // Morton order over axes of unequal depth: bit b of every axis that has one, lowest bits first
    unsigned m = 0;
    int at = 0;
#pragma unroll
    for (int b = 0; b < kCellBitsTotal; b++) {
        if (b < bx) m |= ((ix >> b) & 1u) << at++;
        if (b < bz) m |= ((iz >> b) & 1u) << at++;
        if (b < by) m |= ((iy >> b) & 1u) << at++;
    }
    return m & (kCells - 1u);
}

template <bool STATS>
__device__ __forceinline__ void flush_stats(bool counted, unsigned n_refl, unsigned n_shadow, unsigned n_hits, unsigned n_lit,
                                            const SweepCount &cnt)
{
    if (!STATS) return;
    const int lane = threadIdx.x & 63;
    RtStatsDev *const stats = fresh_args()->stats;
    unsigned v[4] = {n_refl, n_shadow, n_hits, n_lit};
    unsigned long long *dst[4] = {&stats->reflection, &stats->shadow, &stats->hits, &stats->lit};
    for (int k = 0; k < 4; k++) {
        unsigned t = counted ? v[k] : 0u;
        for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
        if (lane == 0 && t) atomicAdd(dst[k], (unsigned long long)t);
    }
    unsigned long long w[3] = {cnt.tests, cnt.gtests, cnt.cands};
    unsigned long long *wdst[3] = {&stats->sphere_tests_run, &stats->group_tests_run, &stats->exact_tests_run};
    for (int k = 0; k < 3; k++) {
        unsigned long long t = w[k];
        for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
        if (lane == 0 && t) atomicAdd(wdst[k], t);
    }
}

// One segment per lane: the hit point of `seg` (main.cpp:94-100), the K rays that leave it (the reflection ray and one
// shadow ray per emitter, swept together as in the one-launch kernel), the segment's record, and the next level's entry.
// `rec` is the index of this segment's record among its level's.  All lanes of the wave call this (ballots inside).
// FAR: the far kernel's call (one segment per wave); otherwise a segment whose hit point lies so far out that its rays have
// to be tested against every sphere (RtSceneDev::far_bound, dense_sweep) is not swept here but handed to that kernel.
template <bool STATS, int GROUPED, bool FAR>
// hand_over: far segments go to the far queue (else they are swept here, one at a time, as in the one-launch kernel).
// keep: the segment the reflection ray found is not queued but returned (next, next_rec, next_alive): the caller follows
// the path itself (the levels from RtSortedDev::final_level on).
__device__ __forceinline__ void sorted_segment(float4 *lds, bool alive, const Segment &seg, unsigned rec, int level,
                                               unsigned &n_refl, unsigned &n_shadow, unsigned &n_hits, SweepCount &cnt,
                                               bool hand_over, bool keep, Segment &next_seg, unsigned &next_rec, bool &next_alive)
{
    const int lane = threadIdx.x & 63;
    RT_TICK(RG_HITPOINT);
    V3 pos = v3(0, 0, 0), refl = v3(0, 0, 0), n = v3(0, 0, 0);
    bool want_refl = false;
    int mat = 0;
    ArgsPtr ad = fresh_args();                                               // this phase: hit point, then the segment's sweeps
    SceneArg &sc = ad->sc;
    const int n_emit = sc.n_emit;
    if (alive) pos = seg.o + seg.d * seg.dist;                               // main.cpp:94
    if (!FAR && hand_over) {
        // A wave whose 64 lanes are far runs 64 passes over the scene one after the other (the tail of every launch);
        // queued, each such segment gets a wave of its own in the launch that follows this one.
        const bool is_far = alive && filter_bound(pos) >= sc.far_bound;
        const unsigned long long fm = __ballot(is_far);
        if (fm) {
            SortedArg &so = fresh_sorted()->so;
            unsigned fbase = 0;
            if (lane == __builtin_ctzll(fm)) fbase = atomicAdd(&so.count_far[level], (unsigned)__builtin_popcountll(fm));
            fbase = (unsigned)__builtin_amdgcn_readlane((int)fbase, __builtin_ctzll(fm));
            if (is_far) {
                const unsigned f = min(fbase + (unsigned)__builtin_popcountll(fm & ((1ull << lane) - 1ull)), so.cap - 1u);
                float4 *const e4 = (level == 0 ? so.farq0 : so.farq) + (size_t)f * 3;
                e4[0] = make_float4(seg.o.x, seg.o.y, seg.o.z, seg.dist);
                e4[1] = make_float4(seg.d.x, seg.d.y, seg.d.z, __int_as_float(seg.slot));
                e4[2] = make_float4(__int_as_float(seg.id), __uint_as_float(rec), 0.0f, 0.0f);
                alive = false;
                pos = v3(0, 0, 0);
            }
        }
    }
    if (alive) {
        if (STATS) n_hits++;
        if (seg.slot >= 0) {
            const int r2 = safe_slot(sc, seg.slot) >> 1;
            const float4 A = sc.pairs[2 * r2], B = sc.pairs[2 * r2 + 1];
            const V3 c = (seg.slot & 1) ? v3(A.y, A.w, B.y) : v3(A.x, A.z, B.x);
            n = normalize(pos - c);                                          // sceneobjects.h:59-62
            mat = 2 * seg.id;
        } else {
            const RtPlaneDev p = sc.planes[min(-1 - seg.slot, sc.n_planes - 1)];
            n = v3(p.nx, p.ny, p.nz);                                        // sceneobjects.h:117-120
            mat = 2 * seg.id + (checker_white(pos) ? 0 : 1);                 // sceneobjects.h:106-115
        }
        refl = normalize(reflect(seg.d, n));                                 // main.cpp:100
        want_refl = level < ad->fr.max_depth && sc.mats[mat].reflectance > 0.0f;   // main.cpp:103
        if (want_refl) n_refl++;
        n_shadow += n_emit;
        // the part of the record that is known now (the sweeps need the registers)
        float4 *const r4 = fresh_sorted()->so.rec + ((size_t)level * fresh_sorted()->so.cap + rec) * 4;
        r4[0] = make_float4(pos.x, pos.y, pos.z, __int_as_float(mat));
        r4[2] = make_float4(refl.x, refl.y, refl.z, 0.0f);
        r4[1] = make_float4(n.x, n.y, n.z, __int_as_float(kNoRay));
    }
    unsigned long long lit = 0, alt = 0;
    Segment next;
    next.dist = 0.0f; next.slot = INT_MIN; next.id = -1;
    bool next_hit = false;
    {
        const bool any_refl = __any(want_refl) != 0;
        int e0 = 0;
        if (!__any(alive)) e0 = n_emit;
        else if (any_refl) {
            // The reflection ray is swept by itself, not with the first shadow rays as in the one-launch kernel: the reflection rays
            // of a wave fan out -- off small spheres even those of a tile's primary hits -- while its shadow rays form beams, and
            // together the fan costs the shadow classes their beam test (16k-sphere scene at 4K: sorted levels only 18.6 -> 17.4 ms,
            // level 0 as well -> 16.4 ms, bound tests 2.8e10 -> 1.5e10 -> 0.8e10).  The shadow rays then go kMaxK at a time (one
            // or two per sweep: 28.8 / 23.6 vs 21.3 ms at the time; 4 + 4 instead of 5 + 3 for eight emitters: the same).
            trace_group_k<true, false, GROUPED>(1, sc, lds, 0, alive, pos, refl, want_refl, 0, lit, alt, next, next_hit, cnt);
        }
        while (e0 < n_emit) {
            const int ne = min(n_emit - e0, kMaxK);
            trace_group_k<false, false, GROUPED>(ne, sc, lds, 0, alive, pos, refl, false, e0, lit, alt, next, next_hit, cnt);
            e0 += ne;
        }
    }
    RT_TICK(RG_AFTER);
    const bool push = alive && want_refl && next_hit && next.slot != INT_MIN;
    // queue the next level's entries: one atomic per wave for the slots, one per lane on its cell's counter
    SortedArg &so = fresh_sorted()->so;
    const unsigned long long pm = __ballot(push);
    // (a kept segment's record comes from the TOP of the next level's records, a queued one's -- which is also its entry's slot --
    // from the bottom: the two never meet, a level has at most `cap` segments, and the queue holds nothing but entries)
    unsigned base = 0;
    if (pm) {
        if (lane == __builtin_ctzll(pm)) base = atomicAdd(keep ? &so.count_kept[level + 1] : &so.count[level + 1], (unsigned)__builtin_popcountll(pm));
        base = (unsigned)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(pm));
    }
    // (the hit points of a wave fall into a handful of cells: one atomic per distinct cell, not 64 on the same counter)
    const V3 o = pos + refl * 0.001f;                                        // main.cpp:108
    // (measured: the octant of the reflection ray the queued segment will send as part of the key buys nothing, 16.8 vs 16.3 ms)
    const unsigned cell = (push && !keep) ? cell_of(so, o + refl * next.dist) : 0u;
    unsigned rank = 0;
    {
        unsigned long long todo = keep ? 0ull : pm;
        while (todo) {
            const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)cell, __builtin_ctzll(todo));
            const unsigned long long same = __ballot(push && cell == c) & todo;
            unsigned first = 0;
            if (lane == __builtin_ctzll(same)) first = atomicAdd(&so.hist[(size_t)(level + 1) * kCells + c], (unsigned)__builtin_popcountll(same));
            first = (unsigned)__builtin_amdgcn_readlane((int)first, __builtin_ctzll(same));
            if ((same >> lane) & 1ull) rank = first + (unsigned)__builtin_popcountll(same & ((1ull << lane) - 1ull));
            todo &= ~same;
        }
    }
    if (alive) {
        float4 *const r4 = so.rec + ((size_t)level * so.cap + rec) * 4;
        int child = want_refl ? kRayMissed : kNoRay;
        if (push) {
            const unsigned nth = base + (unsigned)__builtin_popcountll(pm & ((1ull << lane) - 1ull));
            const unsigned u = keep ? so.cap - 1u - min(nth, so.cap - 1u) : nth;
            child = (int)u;
            if (keep) {
                next_seg.o = o; next_seg.d = refl; next_seg.dist = next.dist; next_seg.slot = next.slot; next_seg.id = next.id;
                next_rec = min(u, so.cap - 1u);
            } else {
                float4 *const e4 = so.queue[(level + 1) & 1] + (size_t)u * 3;
                e4[0] = make_float4(o.x, o.y, o.z, next.dist);
                e4[1] = make_float4(refl.x, refl.y, refl.z, __int_as_float(next.slot));
                e4[2] = make_float4(__int_as_float(next.id), __uint_as_float(cell), __uint_as_float(rank), 0.0f);
            }
        }
        r4[1].w = __int_as_float(child);
        r4[3] = make_float4(__uint_as_float((unsigned)lit), __uint_as_float((unsigned)(lit >> 32)), __uint_as_float((unsigned)alt),
                            __uint_as_float((unsigned)(alt >> 32)));
    }
    next_alive = keep && push;
}

// A lane's segment, and -- from the final level on -- the segments its reflection rays find, to the end of the path.
template <bool STATS, int GROUPED, bool FAR>
__device__ __forceinline__ void sorted_path(float4 *lds, bool alive, Segment seg, unsigned rec, int level,
                                            unsigned &n_refl, unsigned &n_shadow, unsigned &n_hits, SweepCount &cnt)
{
    const int final_level = fresh_sorted()->so.final_level, max_depth = fresh_args()->fr.max_depth;
    for (int lvl = level;; lvl++) {
        // (level 0's far segments -- most of a frame's -- run beside the sorted levels on a stream of their own, so what their
        // reflection rays find cannot join level 1's queue: their waves follow it themselves)
        const bool keep = ((FAR && level == 0) || level >= final_level) && lvl < max_depth;
        Segment nseg;
        nseg.o = v3(0, 0, 0); nseg.d = v3(0, 0, 0); nseg.dist = 0.0f; nseg.slot = INT_MIN; nseg.id = -1;
        unsigned nrec = 0;
        bool nalive = false;
        sorted_segment<STATS, GROUPED, FAR>(lds, alive, seg, rec, lvl, n_refl, n_shadow, n_hits, cnt, lvl == level, keep, nseg, nrec, nalive);
        if (!keep || !__any(nalive)) break;
        seg = nseg; rec = nrec; alive = nalive;
    }
}

// Level 0 (LEVEL0): one lane per (pixel, sample), tiles as in the one-launch kernel; the path's index, pixel * spp +
// sample, is its record's.  Other levels: lane s of the launch takes entry perm[s].
// FAR: wave i of the launch takes far entry i, i + gridDim.x, ... (lane 0 holds the segment, all lanes sweep for it).
template <bool STATS, int GROUPED, bool LEVEL0, bool FAR = false>
__global__ __launch_bounds__(64, RT_WAVES_PER_SIMD_GROUPED) void sorted_level_kernel(RtSortedKernArgs by_value_only_for_the_layout)
{
    (void)by_value_only_for_the_layout;
    extern __shared__ float4 lds[];
    const int lane = threadIdx.x & 63;
    unsigned n_refl = 0, n_shadow = 0, n_hits = 0;
    SweepCount cnt = {0u, 0u, 0u};
#ifdef RT_REGION_TIMERS
    if (threadIdx.x < RT_N_REGIONS) rt_region_acc[threadIdx.x] = 0u;
    if (threadIdx.x == 0) { rt_region_last = __builtin_amdgcn_s_memtime(); rt_region_cur = RG_RAYGEN; }
    __syncthreads();
#endif
    if constexpr (LEVEL0) {
        ArgsPtr a0 = fresh_args();
        FrameArg &fr = a0->fr;
        const int rows = fr.row_end - fr.row_begin;
        const int tiles_x = (fr.width + fr.tile_w - 1) / fr.tile_w;
        const int tiles_y = (rows + fr.tile_h - 1) / fr.tile_h;
        const int ls = fr.lane_samples;                                      // 0 or spp (see the one-launch kernel)
        const int tile = blockIdx.x;
        const int lane_sample = ls ? lane & (ls - 1) : 0, lane_pixel = ls ? lane / ls : lane;
        const int s_begin = lane_sample, s_end = ls ? s_begin + 1 : fr.spp;
        const int tx = tile % tiles_x;
        int ty = tile / tiles_x + fr.first_tile_row;                         // launch order: hot rows first
        if (ty >= tiles_y && tile < tiles_x * tiles_y) ty -= tiles_y;
        const int x = tx * fr.tile_w + lane_pixel % fr.tile_w;
        const int yr = ty * fr.tile_h + lane_pixel / fr.tile_w;
        const int y = fr.row_begin + yr;
        const int x_img = fr.strip_stride ? (((x >> 3) * fr.strip_stride + fr.strip_phase) << 3) + (x & 7) : x;
        const bool in_image = (tile < tiles_x * tiles_y) && x < fr.width && yr < rows;
        const unsigned pixel = (unsigned)yr * (unsigned)fr.width + (unsigned)x;
        const unsigned spp = (unsigned)fr.spp;
        for (int s = s_begin; s < s_end; s++) {                              // main.cpp:203
            Segment seg;
            bool alive;
            {
                ArgsPtr ap = fresh_args();                                   // this phase: primary ray and its sweep
                CameraArg &cam = ap->cam;
                SamplesArg &smp = ap->smp;
                SceneArg &sc = ap->sc;
                const V3 cam_pos = v3(cam.position[0], cam.position[1], cam.position[2]);
                const V3 cam_view = v3(cam.view_direction[0], cam.view_direction[1], cam.view_direction[2]);
                const V3 cam_right = v3(cam.right[0], cam.right[1], cam.right[2]);
                const V3 cam_up = v3(cam.up[0], cam.up[1], cam.up[2]);
                const bool lens_on = ap->fr.lens_on != 0;
                // Camera::GetWorldRay, camera.h:48-62
                float sx = (float)x_img + smp.xy[2 * s], sy = (float)y + smp.xy[2 * s + 1];   // main.cpp:205
                float px = ((sx * 2.0f) / cam.image_width) - 1.0f;
                float py = ((sy * 2.0f) / cam.image_height) - 1.0f;
                V3 d = cam_view;
                d = d + cam_right * (cam.half_angle * cam.aspect_ratio * px);
                d = d - cam_up * (cam.half_angle * py);
                RaySet<1, GROUPED> r;
                r.attach(lds, lane);
                if (lens_on) {
                    // thin lens, raytracer_win/camera.h:160-171
                    const float ft = (ap->fr.focal_length - 1.0f) / sqrtf(dot(d, d));
                    const V3 focus = cam_pos + d * ft;
                    V3 lens = cam_pos;
                    lens = lens + cam_right * smp.lens[2 * s];
                    lens = lens + cam_up * smp.lens[2 * s + 1];
                    r.org = lens;
                    r.set_d(0, normalize(focus - lens));
                } else {
                    r.org = cam_pos;
                    r.set_d(0, normalize(d));
                }
                r.step = 0.0f;                                               // main.cpp:208: the ray starts at the camera itself
                set_dir_n(r, 0, normalize(r.d(0)));                          // sceneobjects.h:71
                r.act = in_image ? 1u : 0u;
                r.set_dec(0, -INFINITY);
                sweep<1, false, GROUPED>(sc, lds, 0, in_image, r, RG_PRIMARY);
                merge_planes(sc, in_image, r);
                if (in_image) { cnt.tests += 2u * r.pairs_run; cnt.gtests += 2u * r.heads_run; cnt.cands += r.cand_run; }
                cnt.gtests += r.beam_run;
                seg.o = r.org; seg.d = r.d(0); seg.dist = r.best(0); seg.slot = r.slot(0); seg.id = r.id[0];
                alive = in_image && r.slot(0) != INT_MIN;
            }
            const unsigned path = pixel * spp + (unsigned)s;
            if (in_image && !alive)                                          // main.cpp:90-93: the shading pass returns (0.1, 0.1, 0.1)
                fresh_sorted()->so.rec[(size_t)path * 4] = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(kNoHit));
            sorted_path<STATS, GROUPED, false>(lds, alive, seg, path, 0, n_refl, n_shadow, n_hits, cnt);
        }
        flush_stats<STATS>(in_image, n_refl, n_shadow, n_hits, 0u, cnt);
    } else if constexpr (FAR) {
        SortedArgsPtr sa = fresh_sorted();
        const int level = sa->level;
        const unsigned count = min(sa->so.count_far[level], sa->so.cap);
        // Level 0's far segments are taken from a counter, not by stride: the launch beside the sorted levels is a thin one (it
        // shares the chip with them), and a second one behind the last level takes whatever the first has not got to by then.
        unsigned *const next_far0 = sa->so.count_kept + (RT_MAX_LEVELS + 1);
        auto take = [&](unsigned prev) -> unsigned {
            if (level != 0) return prev == 0xffffffffu ? blockIdx.x : prev + gridDim.x;
            // (a look before the atomic: the catch-up launch's 8 192 waves mostly find the queue taken, and 8 192 atomics on one
            // word are 90 us of an otherwise empty launch; the counter only grows, so "taken" is final)
            unsigned i = *reinterpret_cast<volatile unsigned *>(next_far0);
            if (lane == 0 && i < count) i = atomicAdd(next_far0, 1u);
            return (unsigned)__builtin_amdgcn_readfirstlane((int)i);
        };
        for (unsigned i = take(0xffffffffu); i < count; i = take(i)) {
            const bool alive = lane == 0;
            Segment seg;
            seg.o = v3(0, 0, 0); seg.d = v3(0, 0, 0); seg.dist = 0.0f; seg.slot = INT_MIN; seg.id = -1;
            unsigned rec = 0;
            if (alive) {
                SortedArg &so = fresh_sorted()->so;
                const float4 *const e4 = (level == 0 ? so.farq0 : so.farq) + (size_t)i * 3;
                const float4 e0 = e4[0], e1 = e4[1], e2 = e4[2];
                seg.o = v3(e0.x, e0.y, e0.z); seg.dist = e0.w;
                seg.d = v3(e1.x, e1.y, e1.z); seg.slot = __float_as_int(e1.w);
                seg.id = __float_as_int(e2.x);
                rec = min(__float_as_uint(e2.y), so.cap - 1u);
            }
            sorted_path<STATS, GROUPED, true>(lds, alive, seg, rec, level, n_refl, n_shadow, n_hits, cnt);
        }
        flush_stats<STATS>(true, n_refl, n_shadow, n_hits, 0u, cnt);
    } else {
        SortedArgsPtr sa = fresh_sorted();
        const int level = sa->level;
        const unsigned count = min(sa->so.count[level], sa->so.cap);
        // a fixed number of waves, each taking every gridDim.x-th run of 64 sorted entries: the count is known on the
        // device only, and a launch sized for the worst case would be mostly waves that start and leave
        for (unsigned first = blockIdx.x * 64u; first < count; first += gridDim.x * 64u) {
            const unsigned s = first + (unsigned)lane;
            const bool alive = s < count;
            Segment seg;
            seg.o = v3(0, 0, 0); seg.d = v3(0, 0, 0); seg.dist = 0.0f; seg.slot = INT_MIN; seg.id = -1;
            unsigned u = 0;
            if (alive) {
                SortedArg &so = fresh_sorted()->so;
                u = min(so.perm[s], so.cap - 1u);                            // (safe_slot()'s reasoning)
                const float4 *const e4 = so.queue[level & 1] + (size_t)u * 3;
                const float4 e0 = e4[0], e1 = e4[1], e2 = e4[2];
                seg.o = v3(e0.x, e0.y, e0.z); seg.dist = e0.w;
                seg.d = v3(e1.x, e1.y, e1.z); seg.slot = __float_as_int(e1.w);
                seg.id = __float_as_int(e2.x);
            }
            sorted_path<STATS, GROUPED, false>(lds, alive, seg, u, level, n_refl, n_shadow, n_hits, cnt);
        }
        flush_stats<STATS>(true, n_refl, n_shadow, n_hits, 0u, cnt);
    }
#ifdef RT_REGION_TIMERS
    RT_TICK(RG_RAYGEN);
    __syncthreads();
    if (fresh_args()->fr.region_cycles && threadIdx.x < RT_N_REGIONS && rt_region_acc[threadIdx.x])
        atomicAdd(&fresh_args()->fr.region_cycles[threadIdx.x], (unsigned long long)rt_region_acc[threadIdx.x]);
#endif
}

// cell counters of one level -> where each cell's entries start (exclusive prefix sum); one workgroup
__global__ __launch_bounds__(1024) void sorted_scan_kernel(unsigned *hist)
{
    __shared__ unsigned part[1024];
    constexpr unsigned kPer = kCells / 1024u;
    const unsigned t = threadIdx.x;
    uint4 *const h = reinterpret_cast<uint4 *>(hist + t * kPer);
    unsigned sum = 0;
    for (unsigned i = 0; i < kPer / 4; i++) { const uint4 v = h[i]; sum += v.x + v.y + v.z + v.w; }
    part[t] = sum;
    __syncthreads();
    for (unsigned off = 1; off < 1024; off <<= 1) {
        const unsigned v = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    unsigned run = part[t] - sum;
    for (unsigned i = 0; i < kPer / 4; i++) {
        uint4 v = h[i];
        const unsigned a = v.x, b = v.y, c = v.z, d = v.w;
        v.x = run; v.y = run + a; v.z = run + a + b; v.w = run + a + b + c;
        run += a + b + c + d;
        h[i] = v;
    }
}

// entry u of this level's queue goes to position start[cell] + rank
__global__ __launch_bounds__(256) void sorted_perm_kernel(RtSortedDev so, int level)
{
    const unsigned count = min(so.count[level], so.cap);
    const float4 *const q = so.queue[level & 1];
    const unsigned *const start = so.hist + (size_t)level * kCells;
    for (unsigned u = blockIdx.x * 256u + threadIdx.x; u < count; u += gridDim.x * 256u) {
        const float4 e2 = q[(size_t)u * 3 + 2];
        const unsigned cell = min(__float_as_uint(e2.y), kCells - 1u);
        const unsigned pos = start[cell] + __float_as_uint(e2.z);
        if (pos < so.cap) so.perm[pos] = u;
    }
}

// The emitter loop and the recursion's arithmetic, per pixel: each sample's chain of records is walked down to its
// deepest segment, then shaded deepest-first (main.cpp:109-187); the samples are added in order (main.cpp:203-210).
template <bool STATS, bool FAST>
__global__ __launch_bounds__(256) void sorted_shade_kernel(RtSortedKernArgs by_value_only_for_the_layout)
{
    (void)by_value_only_for_the_layout;
    ArgsPtr a0 = fresh_args();
    FrameArg &fr = a0->fr;
    SceneArg &sc = a0->sc;
    SortedArg &so = fresh_sorted()->so;
    const unsigned n_px = (unsigned)fr.width * (unsigned)(fr.row_end - fr.row_begin);
    const unsigned pixel = blockIdx.x * 256u + threadIdx.x;
    const bool in_image = pixel < n_px;
    const int n_emit = sc.n_emit;
    V3 color = v3(0.0f, 0.0f, 0.0f);
    unsigned n_rays = 0, n_lit = 0;
    if (in_image) {
        for (int s = 0; s < fr.spp; s++) {                                   // main.cpp:203
            unsigned chain[RT_MAX_LEVELS];
            int nlev = 0;
            bool miss_tail = false;
            unsigned idx = pixel * (unsigned)fr.spp + (unsigned)s;
            n_rays += 1u;
            for (int level = 0; level <= fr.max_depth; level++) {
                const float4 *const r4 = so.rec + ((size_t)level * so.cap + idx) * 4;
                if (level == 0 && __float_as_int(r4[0].w) == kNoHit) { miss_tail = true; break; }
                chain[nlev++] = idx;
                n_rays += (unsigned)n_emit;
                const int child = __float_as_int(r4[1].w);
                if (child == kNoRay) break;
                n_rays += 1u;
                if (child == kRayMissed) { miss_tail = true; break; }
                idx = min((unsigned)child, so.cap - 1u);
            }
            V3 child = v3(0.1f, 0.1f, 0.1f);       // what a missing ray returns (main.cpp:92)
            if constexpr (FAST) {
                // RT_MODE_FAST: the same sum top-down with the throughput carried along (rt_sweep_kernel.inc)
                V3 acc = (nlev == 0 && miss_tail) ? v3(0.1f, 0.1f, 0.1f) : v3(0.0f, 0.0f, 0.0f);
                float tp = 1.0f;
                for (int lev = 0; lev < nlev; lev++) {
                    const float4 *const r4 = so.rec + ((size_t)lev * so.cap + chain[lev]) * 4;
                    const float4 q0 = r4[0], q1 = r4[1], q2 = r4[2], q3 = r4[3];
                    const rt_material m = sc.mats[__float_as_int(q0.w)];
                    const unsigned long long lit = (unsigned long long)__float_as_uint(q3.x) | ((unsigned long long)__float_as_uint(q3.y) << 32);
                    const unsigned long long alt = (unsigned long long)__float_as_uint(q3.z) | ((unsigned long long)__float_as_uint(q3.w) << 32);
                    V3 local = add_lights<STATS, true>(sc, v3(0.0f, 0.0f, 0.0f), v3(q0.x, q0.y, q0.z), v3(q1.x, q1.y, q1.z), v3(q2.x, q2.y, q2.z),
                                                       m, lit, alt, n_lit);
                    local = local * (1.f - m.reflectance) + v3(m.emissive[0], m.emissive[1], m.emissive[2]);
                    acc = acc + local * tp;
                    tp = tp * (m.reflectance * (1.f - m.reflectance));
                }
                if (nlev > 0 && miss_tail) acc = acc + v3(0.1f, 0.1f, 0.1f) * tp;
                child = acc;
            } else {
                bool has_child = miss_tail;
                for (int lev = nlev - 1; lev >= 0; lev--) {
                    const float4 *const r4 = so.rec + ((size_t)lev * so.cap + chain[lev]) * 4;
                    const float4 q0 = r4[0], q1 = r4[1], q2 = r4[2], q3 = r4[3];
                    const rt_material m = sc.mats[__float_as_int(q0.w)];
                    const unsigned long long lit = (unsigned long long)__float_as_uint(q3.x) | ((unsigned long long)__float_as_uint(q3.y) << 32);
                    const unsigned long long alt = (unsigned long long)__float_as_uint(q3.z) | ((unsigned long long)__float_as_uint(q3.w) << 32);
                    V3 out = v3(0.0f, 0.0f, 0.0f);
                    if (has_child) out = child * m.reflectance;              // main.cpp:109
                    out = add_lights<STATS, false>(sc, out, v3(q0.x, q0.y, q0.z), v3(q1.x, q1.y, q1.z), v3(q2.x, q2.y, q2.z), m, lit, alt, n_lit);
                    out = out * (1.f - m.reflectance);                       // main.cpp:185
                    out = out + v3(m.emissive[0], m.emissive[1], m.emissive[2]);   // main.cpp:186
                    child = out;
                    has_child = true;
                }
            }
            color = color + child;                                           // main.cpp:208
        }
        write_pixel(fr, (size_t)pixel, color, (float)n_rays);
    }
    if (STATS) {
        unsigned t = in_image ? n_lit : 0u;
        for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
        if ((threadIdx.x & 63) == 0 && t) atomicAdd(&fresh_args()->stats->lit, (unsigned long long)t);
    }
}

// `side`: a second stream for level 0's far segments, which depend on level 0's launch alone and on which only the shading
// pass depends (ev_level0: recorded on `stream` behind level 0; ev_side: recorded on `side` behind the far launch).
template <bool STATS, int GROUPED>
hipError_t launch_sorted_levels(const RtSortedKernArgs &proto, bool fast, hipStream_t stream, hipStream_t side, hipEvent_t ev_level0,
                                hipEvent_t ev_side)
{
    RtSortedKernArgs args = proto;
    const RtFrameDev &fr = args.base.fr;
    const int rows = fr.row_end - fr.row_begin;
    const long long tiles = (long long)((fr.width + fr.tile_w - 1) / fr.tile_w) * ((rows + fr.tile_h - 1) / fr.tile_h);
    const unsigned long long paths = (unsigned long long)fr.width * (unsigned long long)rows * (unsigned long long)fr.spp;
    if (tiles <= 0) return hipSuccess;
    if (tiles > 0x7fffffffll || paths > 0x7fffffffull || paths > args.so.cap) return hipErrorInvalidValue;
    const int levels = fr.max_depth + 1;
    // the counters of this frame: entries queued per level, entries per cell and level
    hipError_t e = hipMemsetAsync(args.so.count, 0, sizeof(unsigned) * (3 * (RT_MAX_LEVELS + 1) + 1), stream);    // count | count_far | count_kept | next far segment of level 0
    if (e != hipSuccess) return e;
    if (levels > 1) {
        e = hipMemsetAsync(args.so.hist + kCells, 0, sizeof(unsigned) * (size_t)kCells * (size_t)(levels - 1), stream);
        if (e != hipSuccess) return e;
    }
    args.level = 0;
    constexpr unsigned kFarWaves = 8192;            // (twice what the chip holds at 4 waves per SIMD)
    const bool with_far = args.base.sc.far_bound < INFINITY;
    hipLaunchKernelGGL((sorted_level_kernel<STATS, GROUPED, true>), dim3((unsigned)tiles), dim3(64), kGroupedLdsBytes, stream, args);
    if (with_far) {
        if ((e = hipEventRecord(ev_level0, stream)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(side, ev_level0, 0)) != hipSuccess) return e;
        hipLaunchKernelGGL((sorted_level_kernel<STATS, GROUPED, false, true>), dim3(args.so.far0_waves > 0 ? (unsigned)args.so.far0_waves : 1024u), dim3(64),
                           kGroupedLdsBytes, side, args);
        if ((e = hipEventRecord(ev_side, side)) != hipSuccess) return e;
    }
    const unsigned waves = (unsigned)std::min<unsigned long long>((paths + 63) / 64, 16384), perm_blocks = (unsigned)std::min<unsigned long long>((paths + 255) / 256, 4096);
    // (levels past the final one are followed by the final level's waves themselves)
    if (args.so.final_level < 1) args.so.final_level = 1;
    for (int level = 1; level < levels && level <= args.so.final_level; level++) {
        args.level = level;
        hipLaunchKernelGGL(sorted_scan_kernel, dim3(1), dim3(1024), 0, stream, args.so.hist + (size_t)level * kCells);
        hipLaunchKernelGGL(sorted_perm_kernel, dim3(perm_blocks), dim3(256), 0, stream, args.so, level);
        hipLaunchKernelGGL((sorted_level_kernel<STATS, GROUPED, false>), dim3(waves), dim3(64), kGroupedLdsBytes, stream, args);
        if (with_far) hipLaunchKernelGGL((sorted_level_kernel<STATS, GROUPED, false, true>), dim3(kFarWaves), dim3(64), kGroupedLdsBytes, stream, args);
    }
    const unsigned n_px = (unsigned)((unsigned long long)fr.width * (unsigned long long)rows);
    args.level = 0;
    if (with_far) {
        // the levels are through: whatever is left of level 0's far segments gets the whole chip
        hipLaunchKernelGGL((sorted_level_kernel<STATS, GROUPED, false, true>), dim3(kFarWaves), dim3(64), kGroupedLdsBytes, stream, args);
        if ((e = hipStreamWaitEvent(stream, ev_side, 0)) != hipSuccess) return e;
    }
    if (fast) hipLaunchKernelGGL((sorted_shade_kernel<STATS, true>), dim3((n_px + 255) / 256), dim3(256), 0, stream, args);
    else hipLaunchKernelGGL((sorted_shade_kernel<STATS, false>), dim3((n_px + 255) / 256), dim3(256), 0, stream, args);
    return hipGetLastError();
}

} // namespace

size_t rt_sorted_cells(void) { return kCells; }

hipError_t rt_launch_trace_sorted(const RtSceneDev &scene, const rt_camera &cam, const RtFrameDev &frame, const RtSamples &samples,
                                  RtStatsDev *stats, bool fast, const RtSortedDev &sorted, hipStream_t stream, hipStream_t side,
                                  hipEvent_t ev_level0, hipEvent_t ev_side)
{
    RtSortedKernArgs args;
    args.base.sc = scene; args.base.cam = cam; args.base.fr = frame; args.base.smp = samples; args.base.stats = stats;
    args.so = sorted;
    args.level = 0;
    const bool three = scene.n_supers > 64;        // (the ray-by-ray bound tests start one level up, as in rt_launch_trace_sweep)
    if (stats) return three ? launch_sorted_levels<true, 2>(args, fast, stream, side, ev_level0, ev_side) : launch_sorted_levels<true, 1>(args, fast, stream, side, ev_level0, ev_side);
    return three ? launch_sorted_levels<false, 2>(args, fast, stream, side, ev_level0, ev_side) : launch_sorted_levels<false, 1>(args, fast, stream, side, ev_level0, ev_side);
}
