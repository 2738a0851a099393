// rt_sweep_grouped_fast.hip -- the grouped sweep's RT_MODE_FAST instantiations (see rt_sweep_grouped.hip).
#define RT_SWEEP_ONLY_FAST
#include "rt_sweep_kernel.inc"

hipError_t rt_launch_sweep_grouped_fast(const RtSceneDev &scene, const rt_camera &cam, const RtFrameDev &frame, const RtSamples &samples,
                                        RtStatsDev *stats, bool fast, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t stream)
{
    return launch_sweep_class<false, false, 1>(scene, cam, frame, samples, stats, fast, grid, block, lds_bytes, stream);
}
