// rt_sweep_grouped.hip -- the grouped sweep (groups of 8 spheres under two levels of bounds, scalar loads, single-wave workgroups),
// RT_MODE_EXACT; RT_MODE_FAST's instantiations are in rt_sweep_grouped_fast.hip (they want a different scheduling strategy).
#define RT_SWEEP_ONLY_EXACT
#include "rt_sweep_kernel.inc"

hipError_t rt_launch_sweep_grouped_fast(const RtSceneDev &, const rt_camera &, const RtFrameDev &, const RtSamples &, RtStatsDev *, bool,
                                        dim3, dim3, size_t, hipStream_t);

hipError_t rt_launch_sweep_grouped(const RtSceneDev &scene, const rt_camera &cam, const RtFrameDev &frame, const RtSamples &samples,
                              RtStatsDev *stats, bool fast, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t stream)
{
    if (fast) return rt_launch_sweep_grouped_fast(scene, cam, frame, samples, stats, fast, grid, block, lds_bytes, stream);
    return launch_sweep_class<false, false, 1>(scene, cam, frame, samples, stats, fast, grid, block, lds_bytes, stream);
}
