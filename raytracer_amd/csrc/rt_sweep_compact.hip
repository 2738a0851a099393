// rt_sweep_compact.hip -- the flat sweep of LDS-resident scenes with ray compaction.
#include "rt_sweep_kernel.inc"

hipError_t rt_launch_sweep_compact(const RtSceneDev &scene, const rt_camera &cam, const RtFrameDev &frame, const RtSamples &samples,
                              RtStatsDev *stats, bool fast, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t stream)
{
    return launch_sweep_class<false, true, 0>(scene, cam, frame, samples, stats, fast, grid, block, lds_bytes, stream);
}
