#!/usr/bin/env python3
"""Config 5 of BASELINE.json: the 16k-sphere scene, a 240-frame camera orbit at 4K.

Frame-parallel over ranks (frame f on rank f mod N), pipelined inside each rank
(raytracer_amd.sweep).  Examples:
    python tools/camera_sweep.py --frames 8                       # one GPU, no files
    python -m torch.distributed.run --nproc-per-node 8 tools/camera_sweep.py --frames 240 --out /tmp/sweep
Prints one JSON line per rank.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=240)
    ap.add_argument("--total-frames", type=int, default=240, help="frames of the full orbit (sets the angle step)")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--scene", default="grid16k", choices=["grid16k", "grid1k", "default"])
    ap.add_argument("--out", default=None, help="directory for frame_NNNN.bmp (omit: frames are only downloaded)")
    args = ap.parse_args()

    import torch
    from raytracer_amd import capi, sweep

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())   # (ranks may share a GPU: rehearsals, tests)
    torch.cuda.set_device(local)
    kind = {"default": capi.RT_SCENE_DEFAULT, "grid1k": capi.RT_SCENE_GRID1K, "grid16k": capi.RT_SCENE_GRID16K}[args.scene]
    mine = list(range(rank, args.frames, world))
    objs, cam0 = capi.make_scene(kind, args.width, args.height, 0, args.total_frames)
    cams = [capi.make_scene(kind, args.width, args.height, f, args.total_frames)[1] for f in mine]
    ctx = capi.Context(local)
    ctx.upload_scene(objs, cam0)
    res = sweep.render_sequence(ctx, cams, args.width, args.height, args.depth, 1,
                                on_frame=sweep.bmp_writer(args.out, args.width, args.height) if args.out else None,
                                frame_ids=mine, bmp_body=bool(args.out))
    ctx.close()
    res.update({"rank": rank, "world": world, "scene": args.scene, "size": [args.width, args.height]})
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
