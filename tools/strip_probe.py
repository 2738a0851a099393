#!/usr/bin/env python3
"""Renders one rank's share of 8-column strips of the 16k scene a few times (for rocprofv3 --kernel-trace around it):
    python tools/strip_probe.py <n_ranks> <phase> [repeats [row_begin row_end [depth]]]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from raytracer_amd import capi  # noqa: E402

n, phase = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rows = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else None
depth = int(sys.argv[6]) if len(sys.argv) > 6 else 5
W, H = 3840, 2160
objs, cam = capi.make_scene(capi.RT_SCENE_GRID16K, W, H)
ctx = capi.Context(0)
ctx.upload_scene(objs, cam)
p = capi.render_params(W, H, depth, 1, interleave=(n, phase), rows=rows, sweep=capi.RT_SWEEP_SORTED)
for _ in range(reps):
    ctx.render(p)
    ctx.sync()
print("strips", n, phase, "kernel ms", ctx.stats()["kernel_ms"], "sweep", ctx.stats()["sweep"])
