// rt_sweep_chunked.hip -- the flat sweep of scenes that do not fit LDS (streamed in chunks, ray compaction).
#include "rt_sweep_kernel.inc"

hipError_t rt_launch_sweep_chunked(const RtSceneDev &scene, const rt_camera &cam, const RtFrameDev &frame, const RtSamples &samples,
                              RtStatsDev *stats, bool fast, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t stream)
{
    return launch_sweep_class<true, true, 0>(scene, cam, frame, samples, stats, fast, grid, block, lds_bytes, stream);
}
