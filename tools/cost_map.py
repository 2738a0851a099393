#!/usr/bin/env python3
"""Where in the frame does the trace kernel's work sit?  Renders the cost plane (RT_OUT_COST: tests executed per
pixel, weighted as in rt_sweep_kernel.inc) and prints its distribution by row band and by pixel percentile.

    python tools/cost_map.py [--scene grid16k] [--width 960 --height 540 --depth 5]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="grid16k", choices=["default", "grid1k", "grid16k"])
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--height", type=int, default=540)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--bands", type=int, default=27)
    args = ap.parse_args()
    import torch  # noqa: F401
    from raytracer_amd import capi
    kind = {"default": capi.RT_SCENE_DEFAULT, "grid1k": capi.RT_SCENE_GRID1K, "grid16k": capi.RT_SCENE_GRID16K}[args.scene]
    W, H = args.width, args.height
    objs, cam = capi.make_scene(kind, W, H)
    ctx = capi.Context(0)
    ctx.upload_scene(objs, cam)
    ctx.render(capi.render_params(W, H, args.depth, 1, outputs=capi.RT_OUT_F32X4 | capi.RT_OUT_COST, stats=True))
    cost = ctx.download(capi.RT_FMT_F32X4, W, H)[..., 3].astype(np.float64)
    ctx.render(capi.render_params(W, H, args.depth, 1, outputs=capi.RT_OUT_F32X4))
    rays = ctx.download(capi.RT_FMT_F32X4, W, H)[..., 3].astype(np.float64)
    ctx.close()
    total = cost.sum()
    print(f"{args.scene} {W}x{H} depth {args.depth}: cost {total:.4g}, rays {rays.sum():.4g}, cost per ray {total / rays.sum():.1f}")
    step = -(-H // args.bands)
    print("rows        share of cost   rays/px   cost/ray")
    for y0 in range(0, H, step):
        c, r = cost[y0:y0 + step], rays[y0:y0 + step]
        print(f"{y0:5d}-{min(H, y0 + step):5d}   {100 * c.sum() / total:6.2f} %      {r.mean():6.2f}   {c.sum() / max(r.sum(), 1):8.1f}")
    per_ray = cost / np.maximum(rays, 1)
    qs = [50, 90, 99, 99.9, 100]
    print("cost/ray over pixels, percentiles", qs, [float(f"{v:.4g}") for v in np.percentile(per_ray, qs)])
    order = np.sort(cost.ravel())[::-1]
    cum = np.cumsum(order) / total
    for f in (0.001, 0.01, 0.1, 0.5):
        print(f"most expensive {100 * f:.1f} % of pixels hold {100 * cum[int(f * order.size) - 1]:.1f} % of the cost")


if __name__ == "__main__":
    main()
