// rt_sweep_grouped.hip -- the grouped sweep (groups of 8 spheres under two levels of bounds, scalar loads, single-wave workgroups).
#include "rt_sweep_kernel.inc"

hipError_t rt_launch_sweep_grouped(const RtSceneDev &scene, const rt_camera &cam, const RtFrameDev &frame, const RtSamples &samples,
                              RtStatsDev *stats, bool fast, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t stream)
{
    return launch_sweep_class<false, false, 1>(scene, cam, frame, samples, stats, fast, grid, block, lds_bytes, stream);
}
