#!/usr/bin/env python3
"""Generates tests/golden/lens_rays.json from the REFERENCE'S OWN raytracer_win/camera.h.

Runs only in the build container: it loads oracle/_ref/libref_win.so, which oracle/Makefile compiles
from that header where it lies (oracle/ref_win_harness.cpp).  Each row: a camera (position, focal
point, film size), a pixel sample, the lens offset glm::circularRand(0.14f) produced for the row's
rand() seed, and the ray Camera::GetWorldRay (camera.h:147-173) returned for it -- all as IEEE-754
bit patterns.  Data only; nothing of the reference travels to the GPU box.

Usage: make -C oracle && python tests/gen_golden_lens.py
"""
import ctypes as C
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F3 = C.c_float * 3


def bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32).tolist()


def main():
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_win.so"))
    rng = np.random.default_rng(0x1E25)
    cams = [((0, 6, 8), (0, 0, 0), 1024, 768), ((3, 2.5, -7), (0.5, 1, 0), 1920, 1080), ((-40, 12, 33), (4, 0, -9), 640, 480),
            ((0.1, 30, 0.2), (0, 0, 0), 333, 211), ((5, 1.2, 5), (5.5, 1.0, 4.2), 3840, 2160)]
    rows = []
    for pos, focal, w, h in cams:
        for _ in range(40):
            seed = int(rng.integers(1, 2 ** 31 - 1))
            sx, sy = np.float32(rng.random() * w), np.float32(rng.random() * h)
            cam, lens, o, d = (C.c_float * 17)(), (C.c_float * 2)(), F3(), F3()
            ref.ref_win_lens_ray(F3(*pos), F3(*focal), C.c_float(w), C.c_float(h), C.c_uint(seed), C.c_float(sx), C.c_float(sy),
                                 cam, lens, o, d)
            rows.append({"seed": seed, "s": bits([sx, sy]), "camera": bits(list(cam)), "lens": bits(list(lens)),
                         "o": bits(list(o)), "d": bits(list(d))})
    json.dump({"source": "raytracer_win Camera::SetPositionAndFocalPoint/SetFilmSize/PreRender/GetWorldRay via oracle/_ref/libref_win.so; "
                         "camera = position(3) view(3) right(3) up(3) aspect half film_w film_h |focalPoint - position|",
               "rows": rows}, open(os.path.join(ROOT, "tests", "golden", "lens_rays.json"), "w"))
    print(len(rows), "rows")


if __name__ == "__main__":
    main()
