"""rt_render_noalloc (include/rt_api.h): the entry point for callers that cannot allocate.

Checked through the C ABI against rt_render, the existing entry point whose kernels it shares, bit for
bit (radiance words, per-pixel ray counts, 8-bit image, ray counters): two calls in a row on one context,
the second with a smaller input, both through the path that needs the scratch array (flat sweep of a
1028-sphere scene at 4 spp: sample-split plane + counters), then a render that needs no plane, and the
size limit.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reference_frames(capi, ctx, objs, cam, w, h, spp, sweep):
    ctx.upload_scene(objs, cam)
    p = capi.render_params(w, h, 5, spp, outputs=capi.RT_OUT_RGBA8 | capi.RT_OUT_F32X4, stats=True, sweep=sweep)
    ctx.render(p)
    return ctx.download(capi.RT_FMT_RGBA8, w, h), ctx.download(capi.RT_FMT_F32X4, w, h), ctx.stats()


def test_render_noalloc_matches_rt_render(capi):
    import torch
    counters = ("primary_rays", "reflection_rays", "shadow_rays", "hits", "lit", "sphere_tests_run", "group_tests_run", "exact_tests_run")
    ref_ctx, ctx = capi.Context(0), capi.Context(0)
    try:
        # caller-owned output planes, allocated once and up front (poisoned so stale data would show)
        rgba = torch.full((640 * 360,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
        f32 = torch.full((640 * 360 * 4,), float("nan"), dtype=torch.float32, device="cuda")
        cases = [(640, 360, 4, capi.RT_SWEEP_FLAT),      # sample split: 14.7 MB of the scratch array
                 (203, 117, 4, capi.RT_SWEEP_FLAT),      # smaller input, same array: nothing of the first call may leak in
                 (333, 211, 4, capi.RT_SWEEP_GROUPED),   # samples in lanes: counters only
                 (203, 117, 3, capi.RT_SWEEP_GROUPED)]   # spp not a power of two: sample split in the grouped sweep
        off3 = np.asarray([[.1, .2], [.6, .5], [.8, .7]], dtype=np.float32)
        for w, h, spp, sweep in cases:
            objs, cam = capi.make_scene(capi.RT_SCENE_GRID1K, w, h)
            offsets = off3 if spp == 3 else None
            ref_ctx.upload_scene(objs, cam)
            p = capi.render_params(w, h, 5, spp, offsets=offsets, outputs=capi.RT_OUT_RGBA8 | capi.RT_OUT_F32X4, stats=True, sweep=sweep)
            ref_ctx.render(p)
            want_rgba, want_f32, want_st = ref_ctx.download(capi.RT_FMT_RGBA8, w, h), ref_ctx.download(capi.RT_FMT_F32X4, w, h), ref_ctx.stats()

            ctx.upload_scene(objs, cam)                  # allocates: outside the no-allocation window
            ctx.render_noalloc(p, rgba.data_ptr(), w * h * 4, f32.data_ptr(), w * h * 16)
            ctx.sync()
            got_rgba = rgba[: w * h].cpu().numpy().view(np.uint8).reshape(h, w, 4)
            got_f32 = f32[: w * h * 4].cpu().numpy().reshape(h, w, 4)
            assert np.array_equal(got_rgba, want_rgba), (w, h, spp, sweep)
            assert np.array_equal(got_f32.view(np.uint32), want_f32.view(np.uint32)), (w, h, spp, sweep)
            st = ctx.stats()
            assert {k: st[k] for k in counters} == {k: want_st[k] for k in counters}, (w, h, spp, sweep)

        # beyond the fixed array: a clear error, nothing enqueued, the planes untouched
        w, h = 1024, 768                                 # 4 spp x 786 432 px x 16 B = 48 MiB > 32 MiB
        objs, cam = capi.make_scene(capi.RT_SCENE_GRID1K, w, h)
        ctx.upload_scene(objs, cam)
        big = torch.zeros((w * h,), dtype=torch.int32, device="cuda")
        with pytest.raises(capi.RtError) as e:
            ctx.render_noalloc(capi.render_params(w, h, 5, 4, sweep=capi.RT_SWEEP_FLAT), big.data_ptr(), w * h * 4)
        assert e.value.code == capi.RT_ERR_LIMIT and "scratch" in str(e.value)
        assert int(big.abs().sum().item()) == 0
        # ... the same frame in bands that fit
        want = _reference_frames(capi, ref_ctx, objs, cam, w, h, 4, capi.RT_SWEEP_FLAT)[0]
        for y0 in range(0, h, 256):
            p = capi.render_params(w, h, 5, 4, rows=(y0, y0 + 256), sweep=capi.RT_SWEEP_FLAT)
            ctx.render_noalloc(p, big.data_ptr() + y0 * w * 4, 256 * w * 4)
        ctx.sync()
        assert np.array_equal(big.cpu().numpy().view(np.uint8).reshape(h, w, 4), want)
        # bad arguments
        with pytest.raises(capi.RtError) as e:
            ctx.render_noalloc(capi.render_params(64, 64, 5, 1), 0, 0)
        assert e.value.code == capi.RT_ERR_INVALID_ARG
        with pytest.raises(capi.RtError) as e:
            ctx.render_noalloc(capi.render_params(64, 64, 5, 1), big.data_ptr(), 100)
        assert e.value.code == capi.RT_ERR_BUFFER_TOO_SMALL
    finally:
        ctx.close()
        ref_ctx.close()


def test_render_noalloc_is_capturable(capi):
    """What the entry point exists for: it can run where nothing may allocate or synchronise.  One call is recorded
    under HIP stream capture (torch.cuda.graph on the stream the context renders on: a hipMalloc, hipFree, event
    creation or stream synchronisation inside the capture fails it), nothing executes while recording, and the
    replayed graph writes the frame rt_render gives, twice in a row."""
    import torch
    w, h, spp = 320, 180, 4
    objs, cam = capi.make_scene(capi.RT_SCENE_GRID1K, w, h)
    ref_ctx, ctx = capi.Context(0), capi.Context(0)
    try:
        for sweep in (capi.RT_SWEEP_FLAT, capi.RT_SWEEP_GROUPED):      # with and without the sample-split plane in the scratch array
            p = capi.render_params(w, h, 5, spp, outputs=capi.RT_OUT_RGBA8 | capi.RT_OUT_F32X4, sweep=sweep)
            ref_ctx.upload_scene(objs, cam)
            ref_ctx.render(p)
            want_rgba, want_f32 = ref_ctx.download(capi.RT_FMT_RGBA8, w, h), ref_ctx.download(capi.RT_FMT_F32X4, w, h)
            ctx.upload_scene(objs, cam)
            stream = torch.cuda.Stream()
            rgba = torch.zeros((w * h,), dtype=torch.int32, device="cuda")
            f32 = torch.zeros((w * h * 4,), dtype=torch.float32, device="cuda")
            ctx.set_stream(stream.cuda_stream)
            try:
                ctx.render_noalloc(p, rgba.data_ptr(), w * h * 4, f32.data_ptr(), w * h * 16)    # (first launch: code objects load outside the capture)
                stream.synchronize()
                rgba.zero_()
                f32.zero_()
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=stream):
                    ctx.render_noalloc(p, rgba.data_ptr(), w * h * 4, f32.data_ptr(), w * h * 16)
                torch.cuda.synchronize()
                assert int(rgba.abs().sum().item()) == 0                 # recorded, not run
                for _ in range(2):
                    rgba.zero_()
                    graph.replay()
                    torch.cuda.synchronize()
                    assert np.array_equal(rgba.cpu().numpy().view(np.uint8).reshape(h, w, 4), want_rgba), sweep
                    assert np.array_equal(f32.cpu().numpy().reshape(h, w, 4).view(np.uint32), want_f32.view(np.uint32)), sweep
                del graph
            finally:
                ctx.set_stream(0)
    finally:
        ctx.close()
        ref_ctx.close()
