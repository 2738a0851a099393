#!/usr/bin/env python3
"""Renders one band of rows of a scene three times (for rocprofv3 --pmc around it):  python tools/band_pmc.py grid16k 456 464 [spp]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from raytracer_amd import capi  # noqa: E402

scene, y0, y1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 1
W, H = 3840, 2160
kind = {"default": capi.RT_SCENE_DEFAULT, "grid1k": capi.RT_SCENE_GRID1K, "grid16k": capi.RT_SCENE_GRID16K}[scene]
objs, cam = capi.make_scene(kind, W, H)
ctx = capi.Context(0)
ctx.upload_scene(objs, cam)
p = capi.render_params(W, H, 5, spp, rows=(y0, y1))
for _ in range(3):
    ctx.render(p)
print("kernel ms", ctx.stats()["kernel_ms"])
