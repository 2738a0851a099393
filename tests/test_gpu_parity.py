"""GPU parity: the HIP trace kernel, called through the C ABI, against the CPU oracle.

RT_MODE_EXACT must be bit-identical (float radiance and 8-bit pixels) to the oracle, which is itself
pinned to the reference's outputs (tests/test_oracle_golden.py).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _render_gpu(capi, ctx, objs, cam, w, h, depth, spp, offsets=None, rows=None, mode=0, stats=False):
    ctx.upload_scene(objs, cam)
    p = capi.render_params(w, h, depth, spp, offsets=offsets, rows=rows,
                           outputs=capi.RT_OUT_RGBA8 | capi.RT_OUT_F32X4, mode=mode, stats=stats)
    ctx.render(p)
    nrows = h if rows is None else rows[1] - rows[0]
    return ctx.download(capi.RT_FMT_RGB8, w, nrows), ctx.download(capi.RT_FMT_F32X4, w, nrows)


@pytest.mark.parametrize("spp", [1, 4])
def test_default_scene_exact(capi, oracle, gpu_ctx, spp):
    w, h = 1024, 768
    objs, cam = capi.make_scene(capi.RT_SCENE_DEFAULT, w, h)
    p = capi.render_params(w, h, 5, spp)
    rgb_o, f32_o, _ = oracle.render(objs, cam, p, threads=8)
    rgb_g, f32_g = _render_gpu(capi, gpu_ctx, objs, cam, w, h, 5, spp)
    assert np.array_equal(f32_o.view(np.uint32), f32_g.view(np.uint32)), \
        f"{np.count_nonzero(f32_o.view(np.uint32) != f32_g.view(np.uint32))} radiance words differ"
    assert np.array_equal(rgb_o, rgb_g)


def _assert_bits_equal(f32_o, f32_g, rgb_o, rgb_g):
    bad = np.count_nonzero(f32_o.view(np.uint32) != f32_g.view(np.uint32))
    assert bad == 0, f"{bad} radiance/ray-count words differ"
    assert np.array_equal(rgb_o, rgb_g)


@pytest.mark.parametrize("kind,w,h,depth", [(1, 640, 360, 5), (1, 320, 200, 8), (2, 256, 144, 5)])
def test_grid_scenes_exact(capi, oracle, gpu_ctx, kind, w, h, depth):
    """grid-1024 (LDS-resident) and grid-16k (LDS chunk streaming), odd tile edges included."""
    objs, cam = capi.make_scene(kind, w, h, frame=17)
    p = capi.render_params(w, h, depth, 1)
    rgb_o, f32_o, _ = oracle.render(objs, cam, p, threads=16)
    rgb_g, f32_g = _render_gpu(capi, gpu_ctx, objs, cam, w, h, depth, 1)
    _assert_bits_equal(f32_o, f32_g, rgb_o, rgb_g)
