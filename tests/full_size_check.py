#!/usr/bin/env python3
"""One-off, full-size parity runs that are too slow for the test suite (minutes of CPU): whole
frames of BASELINE.json's synthetic configs on the GPU (both sweeps) against the CPU oracle, every
bit of radiance, ray count and 8-bit colour.

    python tests/full_size_check.py grid1k 7680 4320 8 1      # config 4's frame
    python tests/full_size_check.py grid16k 1920 1080 5 1     # config 5's scene, frame 0

Prints a progress line while the oracle works (the GPU box kills silent jobs).
"""
import sys
import threading
import time
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    scene, w, h, depth, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    import torch  # noqa: F401
    from raytracer_amd import capi
    from oracle import oracle_lib
    kind = {"default": capi.RT_SCENE_DEFAULT, "grid1k": capi.RT_SCENE_GRID1K, "grid16k": capi.RT_SCENE_GRID16K}[scene]
    objs, cam = capi.make_scene(kind, w, h)
    ctx = capi.Context(0)
    ctx.upload_scene(objs, cam)
    gpu = {}
    for name, sweep in (("flat", capi.RT_SWEEP_FLAT), ("grouped", capi.RT_SWEEP_GROUPED), ("sorted", capi.RT_SWEEP_SORTED)):
        ctx.render(capi.render_params(w, h, depth, spp, outputs=capi.RT_OUT_RGBA8 | capi.RT_OUT_F32X4, sweep=sweep))
        gpu[name] = (ctx.download(capi.RT_FMT_RGB8, w, h), ctx.download(capi.RT_FMT_F32X4, w, h), ctx.stats()["kernel_ms"])
        print(f"gpu {name}: {gpu[name][2]:.2f} ms", flush=True)
    done = threading.Event()

    def ticker():
        t0 = time.time()
        while not done.wait(30):
            print(f"  oracle running, {time.time() - t0:.0f} s", flush=True)

    threading.Thread(target=ticker, daemon=True).start()
    t0 = time.time()
    rgb_o, f32_o, _ = oracle_lib.render(objs, cam, capi.render_params(w, h, depth, spp), threads=min(16, os.cpu_count() or 1))
    done.set()
    print(f"oracle: {time.time() - t0:.1f} s", flush=True)
    bad = 0
    for name, (rgb, f32, _) in gpu.items():
        d = int(np.count_nonzero(f32.view(np.uint32) != f32_o.view(np.uint32))) + int(np.count_nonzero(rgb != rgb_o))
        print(f"{scene} {w}x{h} depth {depth} spp {spp}, {name} sweep vs oracle: {d} differing words of {f32.size + rgb.size}", flush=True)
        bad += d
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
