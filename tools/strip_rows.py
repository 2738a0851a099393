#!/usr/bin/env python3
"""Which 8x8 tile of rows 464..471 of the 16k scene's frame holds up its launch?  (one 8-column strip per render)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from raytracer_amd import capi
W, H = 3840, 2160
y0 = int(sys.argv[1]) if len(sys.argv) > 1 else 464
objs, cam = capi.make_scene(capi.RT_SCENE_GRID16K, W, H)
ctx = capi.Context(0)
ctx.upload_scene(objs, cam)
res = []
for s in range(W // 8):
    p = capi.render_params(W, H, 5, 1, interleave=(W // 8, s), rows=(y0, y0 + 8), sweep=capi.RT_SWEEP_SORTED)
    for _ in range(2):
        ctx.render(p); ctx.sync()
    res.append((ctx.stats()["kernel_ms"], s))
res.sort(reverse=True)
print("slowest strips (ms, strip):", res[:8], "median", res[len(res) // 2])
