// rt_sweep_small.hip -- the sweep dispatcher, the sample-split resolve launch, and the flat sweep of scenes of a few
// spheres (single-wave workgroups, no compaction).
#include "rt_sweep_kernel.inc"

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

hipError_t rt_launch_sweep_grouped(const RtSceneDev &, const rt_camera &, const RtFrameDev &, const RtSamples &, RtStatsDev *, bool,
                                   dim3, dim3, size_t, hipStream_t);
hipError_t rt_launch_sweep_grouped3(const RtSceneDev &, const rt_camera &, const RtFrameDev &, const RtSamples &, RtStatsDev *, bool,
                                    dim3, dim3, size_t, hipStream_t);
hipError_t rt_launch_sweep_chunked(const RtSceneDev &, const rt_camera &, const RtFrameDev &, const RtSamples &, RtStatsDev *, bool,
                                   dim3, dim3, size_t, hipStream_t);
hipError_t rt_launch_sweep_compact(const RtSceneDev &, const rt_camera &, const RtFrameDev &, const RtSamples &, RtStatsDev *, bool,
                                   dim3, dim3, size_t, hipStream_t);

hipError_t rt_launch_trace_sweep(const RtSceneDev &scene, const rt_camera &cam, const RtFrameDev &frame,
                                 const RtSamples &samples, RtStatsDev *stats, bool fast, bool grouped,
                                 hipStream_t stream)
{
    const int rows = frame.row_end - frame.row_begin;
    const long long tiles64 = (long long)((frame.width + frame.tile_w - 1) / frame.tile_w) * ((rows + frame.tile_h - 1) / frame.tile_h);
    if (tiles64 <= 0) return hipSuccess;
    const long long units64 = frame.sample_buf ? tiles64 * frame.spp : tiles64;   // sample split: one wave per (tile, sample)
    if (units64 > 0x7fffffffll) return hipErrorInvalidValue;                      // the kernel indexes units with an int
    const int units = (int)units64;
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
    const size_t lds_bytes = grouped ? kGroupedLdsBytes : (size_t)(2 * cap + second) * sizeof(float4);   // grouped: beams + ray state columns
    const dim3 grid(one_wave ? units : (units + kWavesPerBlock - 1) / kWavesPerBlock), block(one_wave ? 64 : kBlock);
    if (chunked) return rt_launch_sweep_chunked(scene, cam, frame, samples, stats, fast, grid, block, lds_bytes, stream);
    if (grouped && scene.n_supers > 64) return rt_launch_sweep_grouped3(scene, cam, frame, samples, stats, fast, grid, block, lds_bytes, stream);
    if (grouped) return rt_launch_sweep_grouped(scene, cam, frame, samples, stats, fast, grid, block, lds_bytes, stream);
    if (compact) return rt_launch_sweep_compact(scene, cam, frame, samples, stats, fast, grid, block, lds_bytes, stream);
    return launch_sweep_class<false, false, 0>(scene, cam, frame, samples, stats, fast, grid, block, lds_bytes, stream);
}
