#!/usr/bin/env python3
"""Generates tests/golden/*.json from the REFERENCE'S OWN code.

Runs only in the build container: it loads oracle/_ref/libref.so, which oracle/Makefile compiles
from the reference's headers where they lie (raytracer_cpp/sceneobjects.h, camera.h, writebitmap.h
and the vendored GLM) -- see oracle/ref_harness.cpp.  The fixtures are data only: inputs and the
reference's outputs as IEEE-754 bit patterns.  Nothing of the reference travels to the GPU box.

Also records the reference outputs the survey captured from the real program: BMP
md5s, ten radiance values, camera basis and ray counters -- those pin the trace loop, which cannot be
compiled here (raytracer_cpp/main.cpp needs the un-vendored CmdParser submodule).

Usage: make -C oracle && python tests/gen_golden.py
"""
import ctypes as C
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_lib  # noqa: E402
from raytracer_amd import capi  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
F3 = C.c_float * 3


def bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32).tolist()


def f3(v):
    return F3(*[float(x) for x in v])


def main():
    ref = oracle_lib.load_ref()
    if ref is None:
        raise SystemExit("oracle/_ref/libref.so missing: run `make -C oracle` where /root/reference exists")
    rng = np.random.default_rng(0x60D1)
    os.makedirs(OUT, exist_ok=True)

    # ---- Sphere::Intersects (sceneobjects.h:69-73 -> gtx/intersect.inl:105-123) ----
    rows = []
    def sphere_case(c, r, o, d):
        dist = C.c_float(0.0)
        hit = ref.ref_sphere_intersects(f3(c), C.c_float(r), f3(o), f3(d), C.byref(dist))
        rows.append({"c": bits(c), "r": bits([r])[0], "o": bits(o), "d": bits(d), "hit": int(hit),
                     "dist": bits([dist.value])[0] if hit else None})
    for _ in range(400):
        c = rng.uniform(-20, 20, 3).astype(np.float32)
        r = np.float32(rng.uniform(0.1, 3.0))
        o = rng.uniform(-30, 30, 3).astype(np.float32)
        aim = c + rng.normal(0, float(r) * 0.9, 3).astype(np.float32)
        d = (aim - o).astype(np.float32)
        if rng.random() < 0.5:
            d = d / np.float32(np.linalg.norm(d))
        sphere_case(c, r, o, d.astype(np.float32))
    for _ in range(100):   # origin inside / on the surface / behind
        c = rng.uniform(-5, 5, 3).astype(np.float32)
        r = np.float32(rng.uniform(0.5, 2.0))
        u = rng.normal(0, 1, 3); u /= np.linalg.norm(u)
        scale = rng.choice([0.0, 0.5, 1.0, 1.0000001, 1.001, 3.0])
        o = (c + np.float32(r * scale) * u).astype(np.float32)
        d = rng.normal(0, 1, 3).astype(np.float32)
        sphere_case(c, r, o, d)
    for d in ([0, 0, 0], [1e-30, 0, 0], [1e20, 1e20, 0], [np.inf, 0, 0], [np.nan, 0, 1]):
        sphere_case([0, 0, -5], 1.0, [0, 0, 0], d)
    json.dump({"source": "Sphere::Intersects via oracle/_ref", "rows": rows}, open(os.path.join(OUT, "sphere_intersects.json"), "w"))

    # ---- TiledPlane::Intersects (sceneobjects.h:127-130 -> gtx/intersect.inl:7-24) ----
    rows = []
    def plane_case(po, pn, o, d):
        dist = C.c_float(0.0)
        hit = ref.ref_plane_intersects(f3(po), f3(pn), f3(o), f3(d), C.byref(dist))
        rows.append({"po": bits(po), "pn": bits(pn), "o": bits(o), "d": bits(d), "hit": int(hit),
                     "dist": bits([dist.value])[0] if hit else None})
    for _ in range(300):
        o = rng.uniform(-30, 30, 3).astype(np.float32)
        d = rng.normal(0, 1, 3).astype(np.float32)
        if rng.random() < 0.3:
            d[1] = np.float32(rng.choice([0.0, 1e-8, -1e-8, 1.1920929e-07, 1.19e-07, -0.0]))
        pn = [0, 1, 0] if rng.random() < 0.7 else (lambda v: (v / np.linalg.norm(v)).astype(np.float32))(rng.normal(0, 1, 3))
        plane_case([0, 0, 0] if rng.random() < 0.7 else rng.uniform(-2, 2, 3).astype(np.float32), pn, o, d)
    json.dump({"source": "TiledPlane::Intersects via oracle/_ref", "rows": rows}, open(os.path.join(OUT, "plane_intersects.json"), "w"))

    # ---- TiledPlane::GetMaterial checker (sceneobjects.h:106-115) ----
    rows = []
    vals = list(rng.uniform(-50, 50, 300)) + [0.0, -0.0, 0.5, -0.5, 1.0, -1.0, 2147483520.0, 2147483648.0, -2147483648.0,
                                              -2147483904.0, 4294967296.0, 1e10, -1e10, 3e38, -3e38, 1e-40, 8388607.5, 16777216.0,
                                              float("inf"), float("-inf"), float("nan")]
    for i in range(len(vals)):
        x = np.float32(vals[i]); z = np.float32(vals[(i * 7 + 3) % len(vals)])
        pos = [x, np.float32(0.0), z]
        rows.append({"pos": bits(pos), "white": int(ref.ref_checker_white(f3(pos)))})
    white, black = capi.Material(), capi.Material()
    ref.ref_plane_materials(C.byref(white), C.byref(black))
    mats = {"white": bits(list(white.albedo) + list(white.specular) + [white.reflectance]),
            "black": bits(list(black.albedo) + list(black.specular) + [black.reflectance])}
    json.dump({"source": "TiledPlane::GetMaterial via oracle/_ref", "rows": rows, "materials": mats},
              open(os.path.join(OUT, "checker.json"), "w"))

    # ---- normals, GetRayFrom, normalize(reflect()) ----
    rows = []
    for _ in range(200):
        c = rng.uniform(-10, 10, 3).astype(np.float32)
        r = np.float32(rng.uniform(0.2, 2.0))
        p = rng.uniform(-12, 12, 3).astype(np.float32)
        n, rf, rd = F3(), F3(), F3()
        ref.ref_sphere_normal(f3(c), C.c_float(r), f3(p), n)
        ref.ref_sphere_ray_from(f3(c), C.c_float(r), f3(p), rf)
        i = rng.normal(0, 1, 3).astype(np.float32)
        ref.ref_reflect_dir(f3(i), n, rd)
        rows.append({"c": bits(c), "p": bits(p), "normal": bits(list(n)), "ray_from": bits(list(rf)),
                     "i": bits(i), "reflect": bits(list(rd))})
    json.dump({"source": "Sphere::GetSurfaceNormal/GetRayFrom, glm::normalize(glm::reflect) via oracle/_ref", "rows": rows},
              open(os.path.join(OUT, "vectors.json"), "w"))

    # ---- Camera (camera.h:18-62) ----
    rows = []
    cams = [([0, 6, 8], [0, -.8, -1], 60.0, 1024, 768), ([0, 6, 8], [0, -.8, -1], 60.0, 1920, 1080),
            ([0, 10, 12], [0, -.45, -1], 60.0, 3840, 2160), ([3, 2, -7], [-.3, -.2, 1], 45.0, 641, 479),
            ([120, 40, -91.25], [-120, -40, 0], 60.0, 3840, 2160), ([1, 1, 1], [1, -1, .5], 90.0, 333, 777)]
    for pos, d, fov, w, h in cams:
        cam = capi.Camera()
        ref.ref_camera_init(C.byref(cam), f3(pos), f3(d), C.c_float(fov), w, h)
        rays = []
        for _ in range(24):
            sx, sy = np.float32(rng.uniform(0, w)), np.float32(rng.uniform(0, h))
            out = F3()
            ref.ref_world_ray(f3(pos), f3(d), C.c_float(fov), w, h, C.c_float(sx), C.c_float(sy), out)
            rays.append({"s": bits([sx, sy]), "dir": bits(list(out))})
        rows.append({"pos": bits(pos), "dir": bits(d), "fov": bits([fov])[0], "w": w, "h": h,
                     "camera": bits(list(cam.position) + list(cam.view_direction) + list(cam.right) + list(cam.up) +
                                    [cam.aspect_ratio, cam.half_angle, cam.image_width, cam.image_height]),
                     "rays": rays})
    sizes = (C.c_int * 4)()
    ref.ref_sizes(sizes)
    json.dump({"source": "Camera ctor + GetWorldRay via oracle/_ref", "rows": rows,
               "sizeof": {"Material": sizes[0], "Sphere": sizes[1], "TiledPlane": sizes[2], "Camera": sizes[3]}},
              open(os.path.join(OUT, "camera.json"), "w"))

    # ---- WriteBitmap (writebitmap.h:88-182): md5 of the file for seeded images ----
    rows = []
    for w, h in [(1, 1), (2, 3), (3, 2), (5, 4), (7, 7), (64, 48), (1024, 768)]:
        img = np.random.default_rng(w * 1000 + h).integers(0, 256, (h, w, 3), dtype=np.uint8)
        with tempfile.NamedTemporaryFile(suffix=".bmp") as tf:
            ref.ref_write_bitmap(os.fsencode(tf.name), w, h, img.ctypes.data_as(C.c_void_p))
            data = open(tf.name, "rb").read()
        rows.append({"w": w, "h": h, "seed": w * 1000 + h, "bytes": len(data), "md5": hashlib.md5(data).hexdigest(),
                     "head": data[:54].hex()})
    json.dump({"source": "WriteBitmap via oracle/_ref; image = default_rng(seed).integers(0,256,(h,w,3),uint8)", "rows": rows},
              open(os.path.join(OUT, "bmp.json"), "w"))

    # ---- reference program outputs recorded by the survey ----
    survey = {
        "source": "SURVEY.md sections 6 and 8c: outputs of the verbatim raytracer_cpp program, default scene 1024x768 depth 5",
        "bmp_md5": {"a0": "14fde7512af4d3ff28fca8df037d2cde", "a1": "1ca90b11700b443f2b4c52af5823bbfb"},
        "bmp_header": {"bfSize": 2359350, "offBits": 54, "biSize": 40, "w": 1024, "h": 768, "bpp": 24, "biSizeImage": 2359296},
        "camera": {"up": [0.0, float.fromhex("0x1.8fce08p-1"), float.fromhex("-0x1.3fd806p-1")],
                   "view": [0.0, float.fromhex("-0x1.3fd806p-1"), float.fromhex("-0x1.8fce08p-1")],
                   "half": float.fromhex("0x1.279a74p-1"), "aspect": float.fromhex("0x1.555556p+0")},
        "radiance_a0": {
            "0,0": ["0x1.89374cp-6"] * 3, "1023,767": ["0x1.89374cp-6"] * 3, "100,700": ["0x1.89374cp-6"] * 3,
            "660,180": ["0x1.89374cp-6"] * 3,
            "512,384": ["0x1.e136f2p-4", "0x1.1db498p-6", "0x1.5aa50cp-7"],
            "445,232": ["0x1.03fc8ep+0", "0x1.c6e97cp-2", "0x1.616632p-2"],
            "300,400": ["0x1.ee193cp-3", "0x1.a0488ap-6", "0x1.a0abd0p-3"],
            "750,410": ["0x1.d043a8p+0", "0x1.d043a8p+0", "0x1.a08752p-1"],
            "510,490": ["0x0p+0", "0x1.b44948p-3", "0x1.84706ep-1"],
            "512,300": ["0x1.baebe8p-3", "0x1.2733e0p-5", "0x1.2733e0p-5"]},
        "primary_dir_512_384": ["0x1.3b3bb8p-13", "-0x1.3ff6cep-1", "-0x1.8fb56cp-1"],
        "counters_a1": {"calls": 6752898, "hits": 3791785, "by_depth": [3145728, 3058819, 353060, 121998, 47328, 25965],
                        "nearest_tests": 40517388, "shadow_tests_verbatim": 136504260, "shadow_tests_pruned": 45501420,
                        "shadow_rays_verbatim": 22750710, "shadow_rays_pruned": 7583570, "lit": 5472647},
        "counters_a0": {"calls": 1688209, "hits": 947931, "nearest_tests": 10129254, "shadow_tests_verbatim": 34125516},
    }
    json.dump(survey, open(os.path.join(OUT, "survey_pins.json"), "w"), indent=1)

    # ---- scene generator hashes + sparse radiance of the synthetic scenes (from the pinned oracle) ----
    scenes = {}
    for name, kind, w, h, depth in (("default", 0, 1024, 768, 5), ("grid1k", 1, 3840, 2160, 5), ("grid1k_8k", 1, 7680, 4320, 8),
                                    ("grid16k", 2, 3840, 2160, 5)):
        objs, cam = capi.make_scene(kind, w, h, frame=17)
        blob = bytes(objs)
        # pixel-aligned samples at the reference's single-sample offset (.1,.2), main.cpp:202-205
        px = np.stack([rng.integers(0, w, 192), rng.integers(0, h, 192)], axis=1)
        sxy = (px.astype(np.float32) + np.array([0.1, 0.2], dtype=np.float32)).astype(np.float32)
        rad = oracle_lib.trace_samples(objs, cam, depth, sxy)
        scenes[name] = {"kind": kind, "w": w, "h": h, "depth": depth, "frame": 17, "n": len(objs),
                        "objects_sha256": hashlib.sha256(blob).hexdigest(),
                        "camera": bits(list(cam.position) + list(cam.view_direction) + list(cam.right) + list(cam.up) +
                                       [cam.aspect_ratio, cam.half_angle, cam.image_width, cam.image_height]),
                        "pixels": px.reshape(-1).tolist(),
                        "samples": bits(sxy.reshape(-1)), "radiance": bits(rad.reshape(-1))}
    json.dump({"source": "rt_make_scene + the pinned oracle (orc_trace_samples); full-resolution sparse-pixel pins",
               "scenes": scenes}, open(os.path.join(OUT, "scenes.json"), "w"))
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
