import sys, os
sys.path.insert(0, '/root/repo')
import torch
from raytracer_amd import capi
W, H = 3840, 2160
objs, cam = capi.make_scene(capi.RT_SCENE_GRID16K, W, H)
ctx = capi.Context(0)
ctx.upload_scene(objs, cam)
for rows in ((456, 464), (464, 472), (472, 480), (1000, 1008), (0, 2160)):
    p = capi.render_params(W, H, 5, 1, rows=rows, stats=True)
    ctx.render(p)
    st = ctx.stats()
    p2 = capi.render_params(W, H, 5, 1, rows=rows)
    ctx.render(p2); ctx.render(p2)
    ms = ctx.stats()["kernel_ms"]
    print(rows, "ms %.2f" % ms, {k: st[k] for k in ("primary_rays", "reflection_rays", "shadow_rays", "hits", "sphere_tests_run", "group_tests_run", "exact_tests_run")})
