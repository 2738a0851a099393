"""Work-in-progress check of RT_SWEEP_SORTED against RT_SWEEP_GROUPED (run by naming the file)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi


def render(ctx, objs, cam, w, h, depth, spp, sweep, mode=0, stats=True, **kw):
    ctx.upload_scene(objs, cam)
    p = capi.render_params(w, h, depth, spp, outputs=capi.RT_OUT_RGBA8 | capi.RT_OUT_F32X4, sweep=sweep, mode=mode, stats=stats, **kw)
    ctx.render(p)
    rgb = ctx.download(capi.RT_FMT_RGB8, w, h)
    f32 = ctx.download(capi.RT_FMT_F32X4, w, h)
    return rgb, f32, (ctx.stats() if stats else None)


def main():
    ctx = capi.Context(0)
    bad = 0
    for kind, w, h, depth, spp in ((capi.RT_SCENE_GRID16K, 640, 360, 5, 1), (capi.RT_SCENE_GRID1K, 480, 270, 8, 4),
                                   (capi.RT_SCENE_GRID1K, 333, 211, 5, 1), (capi.RT_SCENE_GRID16K, 203, 117, 3, 4),
                                   (capi.RT_SCENE_GRID1K, 320, 200, 0, 1)):
        objs, cam = capi.make_scene(kind, w, h)
        for mode in (capi.RT_MODE_EXACT, capi.RT_MODE_FAST):
            a = render(ctx, objs, cam, w, h, depth, spp, capi.RT_SWEEP_GROUPED, mode)
            b = render(ctx, objs, cam, w, h, depth, spp, capi.RT_SWEEP_SORTED, mode)
            d = int(np.count_nonzero(a[1].view(np.uint32) != b[1].view(np.uint32)))
            d8 = int(np.count_nonzero(a[0] != b[0]))
            keys = ("reflection_rays", "shadow_rays", "hits", "lit")
            same = all(a[2][k] == b[2][k] for k in keys)
            print(kind, w, h, depth, spp, "mode", mode, "f32 words differing", d, "rgb", d8, "census same", same,
                  {k: (a[2][k], b[2][k]) for k in keys} if not same else "", flush=True)
            bad += (d != 0) + (d8 != 0) + (not same)
    print("BAD" if bad else "OK", flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
