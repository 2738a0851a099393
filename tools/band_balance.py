#!/usr/bin/env python3
"""How well do the N-way splits of a frame balance in time?  Renders each part of a split on one GPU
(cost-balanced bands, equal bands, round-robin strips) and prints the kernel time per part (the slowest
part sets the N-GPU frame time).

    python tools/band_balance.py [--gpus 8] [--scene grid1k] [--width 3840 --height 2160 --spp 4]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=8)
    ap.add_argument("--scene", default="grid1k", choices=["default", "grid1k", "grid16k"])
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--spp", type=int, default=4)
    ap.add_argument("--sweep", default="auto", choices=["auto", "flat", "grouped", "sorted"])
    args = ap.parse_args()
    import torch  # noqa: F401
    from raytracer_amd import capi, multigpu
    kind = {"default": capi.RT_SCENE_DEFAULT, "grid1k": capi.RT_SCENE_GRID1K, "grid16k": capi.RT_SCENE_GRID16K}[args.scene]
    W, H = args.width, args.height
    objs, cam = capi.make_scene(kind, W, H)
    ctx = capi.Context(0)
    ctx.upload_scene(objs, cam)
    cost = multigpu.prepass_row_cost(ctx, objs, cam, W, H, args.depth)
    hot = multigpu.hot_row(cost)
    print(f"hot row {hot}")
    sweep = {"auto": capi.RT_SWEEP_AUTO, "flat": capi.RT_SWEEP_FLAT, "grouped": capi.RT_SWEEP_GROUPED, "sorted": capi.RT_SWEEP_SORTED}[args.sweep]

    def timed(**kw):
        p = capi.render_params(W, H, args.depth, args.spp, sweep=sweep, **kw)
        ctx.render(p)
        ctx.render(p)
        return ctx.stats()["kernel_ms"]

    def line(what, ms):
        print(f"  ms per {what} " + " ".join(f"{m:.2f}" for m in ms) + f" | max {max(ms):.2f}, mean {sum(ms) / len(ms):.2f}, "
              f"max/mean {max(ms) / (sum(ms) / len(ms)):.3f}")

    for first in (0, hot):
        print(f"first_row {first}: whole frame {timed(first_row=first):.2f} ms")
        for name, bounds in (("cost-balanced", multigpu.plan_bands(cost, H, args.gpus)), ("equal rows", capi.split_bands(H, args.gpus, None))):
            print(f" {name}: bounds {bounds}")
            line("band", [timed(rows=(bounds[r], bounds[r + 1]), first_row=first) for r in range(args.gpus)])
        print(" 8-column strips dealt round-robin (col_interleave):")
        line("rank", [timed(interleave=(args.gpus, r), first_row=first) for r in range(args.gpus)])
    ctx.close()


if __name__ == "__main__":
    main()
