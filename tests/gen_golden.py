#!/usr/bin/env python3
"""Generates tests/golden/*.json from the REFERENCE'S OWN code.

Runs only in the build container: it loads oracle/_ref/libref.so, which oracle/Makefile compiles
from the reference's headers where they lie (raytracer_cpp/sceneobjects.h, camera.h, writebitmap.h
and the vendored GLM) -- see oracle/ref_harness.cpp.  The fixtures are data only: inputs and the
reference's outputs as IEEE-754 bit patterns.  Nothing of the reference travels to the GPU box.

The trace loop itself -- InitScene, FindNearestObject, TraceRay, DrawScene of raytracer_cpp/main.cpp --
is compiled from the reference's file as well (oracle/_ref/libref_trace_d<MAX_DEPTH>.so, see
oracle/ref_trace_harness.cpp and oracle/Makefile), and every whole-path pin below is COMPUTED from it:
image.bmp md5s, radiance, call counters, the synthetic scenes at depth 5 and 8, the edge scenes at
depth 0/1/5/15, the 1080p and 4K-16spp frames of BASELINE's configs 2 and 3.

Usage: make -C oracle && python tests/gen_golden.py        (`... gen_golden.py headline`: only headline_4spp.json)
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


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def cam_bits(cam):
    return bits(list(cam.position) + list(cam.view_direction) + list(cam.right) + list(cam.up) +
                [cam.aspect_ratio, cam.half_angle, cam.image_width, cam.image_height])


REF_PATTERN = np.array([0.1, 0.2, 0.6, 0.5, 0.8, 0.7, 0.2, 0.8], dtype=np.float32)      # main.cpp:202
SURVEY_PIXELS = [(0, 0), (1023, 767), (100, 700), (660, 180), (512, 384), (445, 232), (300, 400), (750, 410), (510, 490), (512, 300)]


def ref_trace(depth):
    r = oracle_lib.load_ref_trace(depth)
    if r is None:
        raise SystemExit("oracle/_ref/libref_trace_d%d.so missing: run `make -C oracle` where /root/reference exists" % depth)
    return r


def gen_trace_pins(rng):
    """The reference PROGRAM's path on its own scene: InitScene + DrawScene + WriteBitmap verbatim (what
    main() does, main.cpp:240-255), radiance through the verbatim TraceRay, and the calls TraceRay makes
    into the scene objects (counting subclasses of the reference's Sphere/TiledPlane)."""
    r5 = ref_trace(5)
    r5.init_scene()
    objs, n, cam = r5.get_scene(capi.Object, capi.Camera)
    pins = {"source": "raytracer_cpp/main.cpp compiled in place (oracle/_ref/libref_trace_d*.so): InitScene, DrawScene, "
                      "WriteBitmap, TraceRay verbatim; default scene, ImageWidth x ImageHeight, MAX_DEPTH 5",
            "image": [r5.lib.ref_trace_image_width(), r5.lib.ref_trace_image_height()],
            "init_scene": {"n": n, "objects_sha256": hashlib.sha256(bytes(objs)[: n * C.sizeof(capi.Object)]).hexdigest(),
                           "camera": cam_bits(cam)}}
    w, h = pins["image"]
    md5, rgb_sha, header = {}, {}, None
    for a in (0, 1):
        with tempfile.NamedTemporaryFile(suffix=".bmp") as tf:
            rgb_parts = [r5.draw_scene(tf.name, parts, a) for parts in ((2, 1, 8) if a == 0 else (2,))]   # -p default 2
            data = open(tf.name, "rb").read()
        assert all(np.array_equal(rgb_parts[0], x) for x in rgb_parts), "the image depends on -p"
        md5["a%d" % a] = hashlib.md5(data).hexdigest()
        rgb_sha["a%d" % a] = sha(rgb_parts[0])
        header = data[:54]
        # the harness's own pixel loop (used below for other sizes) must be DrawScene's
        rgb_f, _ = r5.frame(w, h, 4 if a else 1, REF_PATTERN[: 8 if a else 2])
        assert np.array_equal(rgb_f, rgb_parts[0]), "ref_trace_frame differs from the verbatim DrawScene"
    pins["bmp_md5"] = md5
    pins["rgb8_sha256"] = rgb_sha
    import struct
    pins["bmp_header"] = {"bfSize": struct.unpack_from("<I", header, 2)[0], "offBits": struct.unpack_from("<I", header, 10)[0],
                          "biSize": struct.unpack_from("<I", header, 14)[0], "w": struct.unpack_from("<i", header, 18)[0],
                          "h": struct.unpack_from("<i", header, 22)[0], "bpp": struct.unpack_from("<H", header, 28)[0],
                          "biSizeImage": struct.unpack_from("<I", header, 34)[0]}
    # radiance of single samples at the -a 0 offset (.1, .2): the survey's ten pixels and 1024 random ones
    px = np.array(SURVEY_PIXELS + [(int(rng.integers(0, w)), int(rng.integers(0, h))) for _ in range(1024)])
    sxy = (px.astype(np.float32) + REF_PATTERN[:2]).astype(np.float32)
    rad = r5.trace_samples(sxy)
    pins["radiance_a0"] = {"pixels": px.reshape(-1).tolist(), "samples": bits(sxy.reshape(-1)), "radiance": bits(rad.reshape(-1))}
    _, f32 = r5.frame(w, h, 1, REF_PATTERN[:2])
    pins["f32_rgb_sha256_a0"] = sha(f32[..., :3])
    _, f32 = r5.frame(w, h, 4, REF_PATTERN)
    pins["f32_rgb_sha256_a1"] = sha(f32[..., :3])
    # counters: calls the verbatim loop makes.  Intersects = N * calls + N^2 * hits (main.cpp:71-79 and
    # :112-151), GetSurfaceNormal = hits (:95), GetRayFrom = N * hits (:114).  Calls by recursion depth
    # from libraries built with MAX_DEPTH 0..5: a path's first k segments do not depend on MAX_DEPTH.
    dobjs, dcam = capi.make_scene(capi.RT_SCENE_DEFAULT, w, h)
    counters = {}
    for a in (0, 1):
        by_limit = []
        for d in range(6):
            rd = ref_trace(d)
            rd.set_scene(dobjs, dcam, counting=True)
            rd.reset_counters()
            rd.draw_scene(None, 1, a)
            c = rd.counters()
            hits = c["normals"]
            assert c["ray_from"] == n * hits
            assert (c["intersects"] - n * n * hits) % n == 0
            by_limit.append({"calls": (c["intersects"] - n * n * hits) // n, "hits": hits, "intersects": c["intersects"],
                             "materials": c["materials"]})
        full = by_limit[5]
        counters["a%d" % a] = {"calls": full["calls"], "hits": full["hits"], "nearest_tests": n * full["calls"],
                               "shadow_tests_verbatim": n * n * full["hits"], "shadow_rays_verbatim": n * full["hits"],
                               "get_material_calls": full["materials"],
                               "by_depth": [by_limit[0]["calls"]] + [by_limit[d]["calls"] - by_limit[d - 1]["calls"] for d in range(1, 6)]}
    pins["counters"] = counters
    ray = F3()
    ref = oracle_lib.load_ref()
    ref.ref_world_ray(f3([0, 6, 8]), f3([0, -.8, -1]), C.c_float(60.0), w, h, C.c_float(512.1), C.c_float(384.2), ray)
    pins["primary_dir_512_384"] = bits(list(ray))
    json.dump(pins, open(os.path.join(OUT, "trace_pins.json"), "w"))


def gen_scene_pins(rng):
    """Sparse-pixel radiance of the synthetic scenes at BASELINE's full sizes through the verbatim
    TraceRay (O(N^2) per hit as written: ~20 ms per hit at N = 1029, ~4 s at N = 16393)."""
    scenes = {}
    for name, kind, w, h, depth, n_px in (("default", 0, 1024, 768, 5, 192), ("grid1k", 1, 3840, 2160, 5, 192),
                                          ("grid1k_8k", 1, 7680, 4320, 8, 192), ("grid16k", 2, 3840, 2160, 5, 24)):
        objs, cam = capi.make_scene(kind, w, h, frame=17)
        r = ref_trace(depth)
        r.set_scene(objs, cam)
        # pixel-aligned samples at the reference's single-sample offset (.1,.2), main.cpp:202-205
        px = np.stack([rng.integers(0, w, n_px), rng.integers(0, h, n_px)], axis=1)
        sxy = (px.astype(np.float32) + REF_PATTERN[:2]).astype(np.float32)
        rad = r.trace_samples(sxy, threads=8)
        scenes[name] = {"kind": kind, "w": w, "h": h, "depth": depth, "frame": 17, "n": len(objs),
                        "objects_sha256": hashlib.sha256(bytes(objs)).hexdigest(), "camera": cam_bits(cam),
                        "pixels": px.reshape(-1).tolist(), "samples": bits(sxy.reshape(-1)), "radiance": bits(rad.reshape(-1))}
        print("scene pins:", name, flush=True)
    json.dump({"source": "rt_make_scene + the reference's own TraceRay (oracle/_ref/libref_trace_d5/d8.so); "
                         "full-resolution sparse-pixel pins", "scenes": scenes}, open(os.path.join(OUT, "scenes.json"), "w"))


def gen_headline_pins():
    """bench.py's own launch configuration (grid-1024, frame 0, 3840x2160, MAX_DEPTH 5, 4 spp: the samples-in-lanes
    layout of the grouped sweep): 192 pixels x the four offsets of main.cpp:202 through the verbatim TraceRay, summed
    in sample order and scaled by 1/4 as main.cpp:203-210 does (float adds in that order).  Own file, own RNG, so
    that adding it moved no other fixture: `python tests/gen_golden.py headline` writes only this one."""
    w, h, depth, spp, n_px = 3840, 2160, 5, 4, 192
    rng = np.random.default_rng(0x4EAD11)
    objs, cam = capi.make_scene(1, w, h)
    r = ref_trace(depth)
    r.set_scene(objs, cam)
    # a third of the pixels from the rows under the horizon (the far-lane path), the rest anywhere
    px = np.stack([rng.integers(0, w, n_px), rng.integers(0, h, n_px)], axis=1)
    px[: n_px // 3, 1] = rng.integers(0, h // 4, n_px // 3)
    sxy = (px[:, None, :].astype(np.float32) + REF_PATTERN.reshape(4, 2)[None, :spp, :]).astype(np.float32).reshape(-1, 2)
    rad = r.trace_samples(sxy, threads=8).reshape(n_px, spp, 3).astype(np.float32)
    color = np.zeros((n_px, 3), dtype=np.float32)
    for s in range(spp):
        color = (color + rad[:, s, :]).astype(np.float32)              # main.cpp:208
    color = (color * np.float32(1.0 / spp)).astype(np.float32)          # main.cpp:210
    json.dump({"source": "rt_make_scene(grid1k, frame 0) + the reference's own TraceRay (oracle/_ref/libref_trace_d5.so), "
                         "4 samples per pixel summed and scaled as main.cpp:203-210",
               "kind": 1, "w": w, "h": h, "depth": depth, "spp": spp, "frame": 0, "n": len(objs),
               "objects_sha256": hashlib.sha256(bytes(objs)).hexdigest(), "camera": cam_bits(cam),
               "pixels": px.reshape(-1).tolist(), "sample_radiance": bits(rad.reshape(-1)), "radiance": bits(color.reshape(-1))},
              open(os.path.join(OUT, "headline_4spp.json"), "w"))
    print("headline pins: %d pixels x %d samples" % (n_px, spp), flush=True)


def gen_edge_pins():
    """Whole small frames of tests/edge_scenes.py through the reference's TraceRay at MAX_DEPTH 0/1/5/15,
    spp 1 and 4: sha256 of the radiance bits and of the 8-bit image, plus a few radiance values to look at."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import edge_scenes
    out = {}
    for depth, (w, h) in edge_scenes.CASES:
        r = ref_trace(depth)
        cam = capi.camera_init(*edge_scenes.CAMERA, w, h)
        for name, items in edge_scenes.edge_scenes(capi).items():
            objs = capi.objects(items)
            r.set_scene(objs, cam)
            for spp in (1, 4):
                rgb, f32 = r.frame(w, h, spp, REF_PATTERN[: 2 * spp])
                out["%s/d%d/%dx%d/spp%d" % (name, depth, w, h, spp)] = {
                    "objects_sha256": hashlib.sha256(bytes(objs)).hexdigest(), "f32_rgb_sha256": sha(f32[..., :3]),
                    "rgb8_sha256": sha(rgb), "first_row": bits(f32[0, : min(w, 8), :3].reshape(-1))}
    json.dump({"source": "tests/edge_scenes.py rendered by the reference's own TraceRay (oracle/_ref/libref_trace_d*.so) inside "
                         "DrawScene's pixel loop; camera = InitScene's", "frames": out}, open(os.path.join(OUT, "edge_scenes.json"), "w"))


def gen_config_pins():
    """BASELINE configs 2 and 3 on the default scene through the reference's TraceRay: the whole 1920x1080
    frame at -a 0 and -a 1, and rows 1200-1296 of the 3840x2160 frame with config 3's 16 stratified samples."""
    r = ref_trace(5)
    out = {}
    w, h = 1920, 1080
    objs, cam = capi.make_scene(capi.RT_SCENE_DEFAULT, w, h)
    r.set_scene(objs, cam)
    for spp in (1, 4):
        rgb, f32 = r.frame(w, h, spp, REF_PATTERN[: 2 * spp])
        out["config2/spp%d" % spp] = {"w": w, "h": h, "depth": 5, "spp": spp, "rows": [0, h], "f32_rgb_sha256": sha(f32[..., :3]),
                                      "rgb8_sha256": sha(rgb)}
    w, h, rows = 3840, 2160, (1200, 1296)
    off = capi.stratified_offsets(16)
    objs, cam = capi.make_scene(capi.RT_SCENE_DEFAULT, w, h)
    r.set_scene(objs, cam)
    rgb, f32 = r.frame(w, h, 16, off, rows=rows)
    out["config3/band"] = {"w": w, "h": h, "depth": 5, "spp": 16, "rows": list(rows), "offsets": bits(np.asarray(off).reshape(-1)),
                           "f32_rgb_sha256": sha(f32[rows[0]:rows[1], :, :3]), "rgb8_sha256": sha(rgb[rows[0]:rows[1]])}
    json.dump({"source": "default scene through the reference's own TraceRay (oracle/_ref/libref_trace_d5.so) inside DrawScene's "
                         "pixel loop", "frames": out}, open(os.path.join(OUT, "config_frames.json"), "w"))


def main():
    if sys.argv[1:] == ["headline"]:
        os.makedirs(OUT, exist_ok=True)
        gen_headline_pins()
        return
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

    gen_trace_pins(rng)
    gen_scene_pins(rng)
    gen_edge_pins()
    gen_config_pins()
    gen_headline_pins()
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
