"""CPU tests: the oracle (oracle/rt_oracle.c) against the reference's own outputs.

* tests/golden/trace_pins.json, scenes.json, edge_scenes.json, config_frames.json -- outputs of the
  reference's own InitScene / TraceRay / DrawScene / WriteBitmap, compiled from raytracer_cpp/main.cpp
  where it lies (oracle/_ref/libref_trace_d*.so, oracle/ref_trace_harness.cpp) and recorded by
  tests/gen_golden.py: BMP md5s, radiance, call counters, the synthetic scenes at depth 5 and 8, the
  edge scenes at depth 0/1/5/15, BASELINE's configs 2 and 3.  They pin the whole trace loop.
* the other fixtures -- outputs of the reference's own headers (oracle/_ref/libref.so): pin every
  primitive bit for bit.
* where the built reference libraries are present (this container; they travel to the GPU box with the
  tree, the reference's sources never do), the oracle is also compared with them live on random scenes.
"""
import ctypes as C
import hashlib
import json
import os

import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
F3 = C.c_float * 3


def load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def fl(bits):
    return np.asarray(bits, dtype=np.uint32).view(np.float32)


def f3(bits):
    return F3(*[float(x) for x in fl(bits)])


def same_bits(a, b):
    a = np.asarray(a, dtype=np.float32).view(np.uint32)
    b = np.asarray(b, dtype=np.float32).view(np.uint32)
    return np.array_equal(a, b)


def test_sphere_intersects_kat(oracle):
    lib = oracle.lib()
    rows = load("sphere_intersects.json")["rows"]
    assert len(rows) > 400
    hits = 0
    for r in rows:
        dist = C.c_float(0.0)
        hit = lib.orc_sphere_intersects(f3(r["c"]), C.c_float(float(fl([r["r"]])[0])), f3(r["o"]), f3(r["d"]), C.byref(dist))
        assert hit == r["hit"], r
        if hit:
            hits += 1
            assert same_bits([dist.value], fl([r["dist"]])), r
    assert 50 < hits < len(rows) - 50          # the table exercises both outcomes


def test_plane_intersects_kat(oracle):
    lib = oracle.lib()
    for r in load("plane_intersects.json")["rows"]:
        dist = C.c_float(0.0)
        hit = lib.orc_plane_intersects(f3(r["po"]), f3(r["pn"]), f3(r["o"]), f3(r["d"]), C.byref(dist))
        assert hit == r["hit"], r
        if hit:
            assert same_bits([dist.value], fl([r["dist"]])), r


def test_checker_kat(oracle, capi):
    lib = oracle.lib()
    g = load("checker.json")
    seen = set()
    for r in g["rows"]:
        w = lib.orc_checker_white(f3(r["pos"]))
        assert w == r["white"], (r, fl(r["pos"]))
        seen.add(w)
    assert seen == {0, 1}
    # TiledPlane's two materials as rt_make_scene emits them
    objs, _ = capi.make_scene(capi.RT_SCENE_DEFAULT, 64, 48)
    plane = objs[len(objs) - 1]
    assert plane.type == capi.RT_OBJ_TILED_PLANE
    for name, m in (("white", plane.mat), ("black", plane.mat2)):
        got = list(m.albedo) + list(m.specular) + [m.reflectance]
        assert same_bits(got, fl(g["materials"][name]))
        assert list(m.emissive) == [0.0, 0.0, 0.0]


def test_vector_kat(oracle):
    lib = oracle.lib()
    for r in load("vectors.json")["rows"]:
        n, rf, rd = F3(), F3(), F3()
        lib.orc_sphere_normal(f3(r["c"]), f3(r["p"]), n)
        lib.orc_ray_from(f3(r["c"]), f3(r["p"]), rf)
        lib.orc_reflect_dir(f3(r["i"]), n, rd)
        assert same_bits(list(n), fl(r["normal"]))
        assert same_bits(list(rf), fl(r["ray_from"]))
        assert same_bits(list(rd), fl(r["reflect"]))


def _cam_list(cam):
    return (list(cam.position) + list(cam.view_direction) + list(cam.right) + list(cam.up) +
            [cam.aspect_ratio, cam.half_angle, cam.image_width, cam.image_height])


def test_camera_kat(oracle, capi):
    lib = oracle.lib()
    g = load("camera.json")
    assert g["sizeof"] == {"Material": 40, "Sphere": 64, "TiledPlane": 112, "Camera": 64}
    assert C.sizeof(capi.Material) == 40 and C.sizeof(capi.Camera) == 64 and C.sizeof(capi.Object) == 112
    for r in g["rows"]:
        cam = capi.Camera()
        lib.orc_camera_init(C.byref(cam), f3(r["pos"]), f3(r["dir"]), C.c_float(float(fl([r["fov"]])[0])), r["w"], r["h"])
        assert same_bits(_cam_list(cam), fl(r["camera"]))
        # the product's host helper (rt_camera_init) is the same constructor
        cam2 = capi.camera_init(fl(r["pos"]), fl(r["dir"]), float(fl([r["fov"]])[0]), r["w"], r["h"])
        assert same_bits(_cam_list(cam2), fl(r["camera"]))
        for ray in r["rays"]:
            s = fl(ray["s"])
            out = F3()
            lib.orc_world_ray(C.byref(cam), C.c_float(float(s[0])), C.c_float(float(s[1])), out)
            assert same_bits(list(out), fl(ray["dir"]))


def test_bmp_writer_kat(oracle, capi, tmp_path):
    for r in load("bmp.json")["rows"]:
        img = np.random.default_rng(r["seed"]).integers(0, 256, (r["h"], r["w"], 3), dtype=np.uint8)
        for name, writer in (("oracle", oracle.write_bmp), ("product", capi.write_bmp)):
            path = str(tmp_path / f"{name}_{r['w']}x{r['h']}.bmp")
            writer(path, img)
            data = open(path, "rb").read()
            assert len(data) == r["bytes"], name
            assert data[:54].hex() == r["head"], name
            assert hashlib.md5(data).hexdigest() == r["md5"], name


@pytest.fixture(scope="module")
def default_frames(oracle, capi):
    """The two renders the reference program can produce: -a 0 and -a 1, 1024x768, depth 5."""
    objs, cam = capi.make_scene(capi.RT_SCENE_DEFAULT, 1024, 768)
    out = {}
    for a, spp in ((0, 1), (1, 4)):
        p = capi.render_params(1024, 768, 5, spp)
        out[a] = oracle.render(objs, cam, p, threads=8, want_stats=True)
    return objs, cam, out


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_reference_program_md5(oracle, default_frames, tmp_path):
    """image.bmp of the reference program (InitScene + DrawScene + WriteBitmap verbatim): the end-to-end
    pin of the trace loop."""
    pins = load("trace_pins.json")
    assert pins["bmp_md5"] == {"a0": "14fde7512af4d3ff28fca8df037d2cde", "a1": "1ca90b11700b443f2b4c52af5823bbfb"}   # SURVEY.md 8c agrees
    _, _, frames = default_frames
    for a in (0, 1):
        path = str(tmp_path / f"a{a}.bmp")
        oracle.write_bmp(path, frames[a][0])
        data = open(path, "rb").read()
        assert len(data) == pins["bmp_header"]["bfSize"]
        assert hashlib.md5(data).hexdigest() == pins["bmp_md5"][f"a{a}"]
        assert sha(frames[a][0]) == pins["rgb8_sha256"][f"a{a}"]
        assert sha(frames[a][1][..., :3]) == pins[f"f32_rgb_sha256_a{a}"]     # every radiance bit of the frame


def test_init_scene_is_the_references(capi):
    """rt_make_scene(RT_SCENE_DEFAULT) against the objects and camera InitScene (main.cpp:21-63) built."""
    pins = load("trace_pins.json")
    objs, cam = capi.make_scene(capi.RT_SCENE_DEFAULT, *pins["image"])
    assert len(objs) == pins["init_scene"]["n"]
    assert hashlib.sha256(bytes(objs)).hexdigest() == pins["init_scene"]["objects_sha256"]
    assert same_bits(_cam_list(cam), fl(pins["init_scene"]["camera"]))


def test_reference_program_radiance_and_camera(oracle, capi, default_frames):
    pins = load("trace_pins.json")
    objs, cam, frames = default_frames
    f32 = frames[0][1]
    r = pins["radiance_a0"]
    px = np.asarray(r["pixels"]).reshape(-1, 2)
    assert len(px) > 1000
    assert same_bits(f32[px[:, 1], px[:, 0], :3], fl(r["radiance"]).reshape(-1, 3))
    rad = oracle.trace_samples(objs, cam, 5, fl(r["samples"]).reshape(-1, 2))
    assert same_bits(rad, fl(r["radiance"]).reshape(-1, 3))
    d = F3()
    oracle.lib().orc_world_ray(C.byref(cam), C.c_float(512.1), C.c_float(384.2), d)
    assert same_bits(list(d), fl(pins["primary_dir_512_384"]))


def test_reference_program_counters(default_frames):
    """Calls the verbatim loop makes into the scene objects (counting subclasses of the reference's
    Sphere/TiledPlane): TraceRay calls in total and by recursion depth, hits, object tests."""
    pins = load("trace_pins.json")["counters"]
    _, _, frames = default_frames
    for a in (0, 1):
        st, c = frames[a][2], pins[f"a{a}"]
        assert st["calls"] == c["calls"] and st["hits"] == c["hits"]
        assert st["calls_by_depth"][:6] == c["by_depth"]
        assert st["nearest_tests"] == c["nearest_tests"]
        assert st["shadow_rays_verbatim"] == c["shadow_rays_verbatim"]
        # the pruned emitter loop runs E = 2 of the N = 6 shadow sweeps the loop as written runs
        assert st["shadow_tests"] * 6 == c["shadow_tests_verbatim"] * 2
        assert st["shadow_rays_pruned"] * 6 == c["shadow_rays_verbatim"] * 2


def test_edge_scenes_against_reference_frames(oracle, capi):
    """tests/edge_scenes.py: whole frames at MAX_DEPTH 0/1/5/15, spp 1 and 4, against the hashes of the
    reference's own TraceRay output (plane emitters and the alternate tile material, distance ties,
    several planes, negative plane distances, emitters that are the hit object ...)."""
    import edge_scenes
    pins = load("edge_scenes.json")["frames"]
    seen = 0
    for depth, (w, h) in edge_scenes.CASES:
        cam = capi.camera_init(*edge_scenes.CAMERA, w, h)
        for name, items in edge_scenes.edge_scenes(capi).items():
            objs = capi.objects(items)
            for spp in (1, 4):
                pin = pins[f"{name}/d{depth}/{w}x{h}/spp{spp}"]
                assert hashlib.sha256(bytes(objs)).hexdigest() == pin["objects_sha256"]
                for verbatim in (False, True):
                    rgb, f32, _ = oracle.render(objs, cam, capi.render_params(w, h, depth, spp), verbatim=verbatim, threads=8)
                    assert sha(f32[..., :3]) == pin["f32_rgb_sha256"], (name, depth, spp, verbatim)
                    assert sha(rgb) == pin["rgb8_sha256"], (name, depth, spp, verbatim)
                seen += 1
    assert seen == len(pins)


def test_config_frames_against_reference(oracle, capi):
    """BASELINE configs 2 and 3 on the CPU: the oracle's 1920x1080 frames (-a 0, -a 1) and a band of the
    3840x2160 16-sample frame against the hashes of the reference's own TraceRay output."""
    pins = load("config_frames.json")["frames"]
    for key, pin in pins.items():
        w, h, rows = pin["w"], pin["h"], pin["rows"]
        if key == "config2/spp4":
            continue            # 8 M paths: left to the GPU suite, which checks the same pin
        objs, cam = capi.make_scene(capi.RT_SCENE_DEFAULT, w, h)
        off = fl(pin["offsets"]).reshape(-1, 2) if "offsets" in pin else None
        if key.startswith("config3"):
            rows = [rows[0], rows[0] + 16]      # a slice of the band here; the GPU suite checks all of it
        p = capi.render_params(w, h, pin["depth"], pin["spp"], offsets=off, rows=tuple(rows))
        rgb, f32, _ = oracle.render(objs, cam, p, threads=8)
        if key.startswith("config3"):
            ref = oracle.load_ref_trace(5)
            if ref is None:
                continue
            ref.set_scene(objs, cam)
            rgb_r, f32_r = ref.frame(w, h, pin["spp"], off, rows=tuple(rows))
            assert np.array_equal(rgb[rows[0]:rows[1]], rgb_r[rows[0]:rows[1]])
            assert same_bits(f32[rows[0]:rows[1], :, :3], f32_r[rows[0]:rows[1], :, :3])
        else:
            assert sha(f32[rows[0]:rows[1], :, :3]) == pin["f32_rgb_sha256"], key
            assert sha(rgb[rows[0]:rows[1]]) == pin["rgb8_sha256"], key


def test_oracle_equals_reference_live(oracle, capi):
    """Random scenes (tests/fuzz_sweeps.py: slabs, blobs, shells, lines; 0-3 planes; shuffled order) and
    random image samples through the reference's own TraceRay (oracle/_ref, where built) and through
    the oracle, verbatim and pruned emitter loop: every radiance bit equal."""
    import fuzz_sweeps
    rng = np.random.default_rng(0xC0FFEE)
    checked = 0
    refs = {}
    for _ in range(60):
        objs, cam, w, h, depth, _spp = fuzz_sweeps.random_scene(capi, rng)
        if depth == 9:
            depth = 8
        if depth not in refs:
            refs[depth] = oracle.load_ref_trace(depth)
        ref = refs[depth]
        if ref is None:
            pytest.skip("oracle/_ref/libref_trace_d*.so not built (no /root/reference here)")
        n = 96 if len(objs) < 300 else 12          # the loop as written is O(N^2) per hit
        sxy = (rng.random((n, 2)) * (w, h)).astype(np.float32)
        ref.set_scene(objs, cam)
        want = ref.trace_samples(sxy, threads=8)
        got = oracle.trace_samples(objs, cam, depth, sxy)
        assert same_bits(got, want), (len(objs), depth)
        if len(objs) < 300:
            got_v = oracle.trace_samples(objs, cam, depth, sxy, verbatim=True)
            assert same_bits(got_v, want), (len(objs), depth)
        checked += n
    assert checked > 2000


def test_pruned_equals_verbatim(oracle, capi):
    """looping only over emissive objects is bit-identical to the loop as written."""
    w, h = 256, 192
    objs, cam = capi.make_scene(capi.RT_SCENE_DEFAULT, w, h)
    p = capi.render_params(w, h, 5, 4)
    rgb_v, f32_v, st_v = oracle.render(objs, cam, p, verbatim=True, threads=8, want_stats=True)
    rgb_p, f32_p, st_p = oracle.render(objs, cam, p, verbatim=False, threads=8, want_stats=True)
    assert np.array_equal(rgb_v, rgb_p)
    assert np.array_equal(f32_v[..., :3].view(np.uint32), f32_p[..., :3].view(np.uint32))
    assert st_v["shadow_tests"] == 36 * st_v["hits"] and st_p["shadow_tests"] == 12 * st_p["hits"]


def test_partition_independence(oracle, capi):
    """main.cpp:193-196: the image does not depend on the number of threads."""
    w, h = 160, 120
    objs, cam = capi.make_scene(capi.RT_SCENE_DEFAULT, w, h)
    p = capi.render_params(w, h, 5, 1)
    a = oracle.render(objs, cam, p, threads=1)[0]
    for t in (2, 3, 8):
        assert np.array_equal(a, oracle.render(objs, cam, p, threads=t)[0])


def test_scene_generators_and_sparse_radiance(oracle, capi):
    """rt_make_scene is deterministic (sha256 of the object array) and the oracle reproduces the
    full-resolution sparse-pixel radiance the reference's own TraceRay gave for every synthetic scene
    (grid-1024 at 4K depth 5 and at 8K depth 8, grid-16k at 4K depth 5)."""
    g = load("scenes.json")["scenes"]
    for name, s in g.items():
        objs, cam = capi.make_scene(s["kind"], s["w"], s["h"], frame=s["frame"])
        assert len(objs) == s["n"], name
        assert hashlib.sha256(bytes(objs)).hexdigest() == s["objects_sha256"], name
        assert same_bits(_cam_list(cam), fl(s["camera"])), name
        sxy = fl(s["samples"]).reshape(-1, 2)
        rad = oracle.trace_samples(objs, cam, s["depth"], sxy)
        assert same_bits(rad.reshape(-1), fl(s["radiance"])), name


def test_headline_workload_4spp_pins(oracle, capi):
    """bench.py's own workload (grid-1024 at 3840x2160, MAX_DEPTH 5, 4 spp): the oracle's per-sample radiance and the
    pixel value after main.cpp:203-210's sum and scale, for 192 pixels, against the reference's own TraceRay
    (tests/golden/headline_4spp.json); a third of the pixels lie in the rows under the horizon."""
    g = load("headline_4spp.json")
    objs, cam = capi.make_scene(g["kind"], g["w"], g["h"], frame=g["frame"])
    assert len(objs) == g["n"] and hashlib.sha256(bytes(objs)).hexdigest() == g["objects_sha256"]
    assert same_bits(_cam_list(cam), fl(g["camera"]))
    px = np.asarray(g["pixels"]).reshape(-1, 2)
    pattern = np.asarray([0.1, 0.2, 0.6, 0.5, 0.8, 0.7, 0.2, 0.8], dtype=np.float32).reshape(4, 2)
    sxy = (px[:, None, :].astype(np.float32) + pattern[None, : g["spp"], :]).astype(np.float32).reshape(-1, 2)
    rad = oracle.trace_samples(objs, cam, g["depth"], sxy)
    assert same_bits(rad.reshape(-1), fl(g["sample_radiance"]))
    # the oracle's own pixel loop on the rows that hold those pixels: same sum, same scale
    for y in sorted(set(px[:8, 1].tolist())):
        _, f32, _ = oracle.render(objs, cam, capi.render_params(g["w"], g["h"], g["depth"], g["spp"], rows=(y, y + 1)), threads=8, want_rgb8=False)
        row = f32[y] if f32.shape[0] == g["h"] else f32[0]
        for i in np.nonzero(px[:, 1] == y)[0]:
            assert same_bits(row[px[i, 0], :3], fl(g["radiance"]).reshape(-1, 3)[i]), (y, i)


def test_thin_lens_ray_kat(oracle, capi):
    """orc_lens_ray against raytracer_win's Camera::GetWorldRay (camera.h:147-173): 200 rays of the
    reference header itself, with the lens offset its glm::circularRand(0.14f) drew for each."""
    rows = load("lens_rays.json")["rows"]
    assert len(rows) == 200
    for r in rows:
        c = fl(r["camera"])
        cam = capi.Camera()
        cam.position[:], cam.view_direction[:], cam.right[:], cam.up[:] = c[0:3], c[3:6], c[6:9], c[9:12]
        cam.aspect_ratio, cam.half_angle, cam.image_width, cam.image_height = c[12], c[13], c[14], c[15]
        o, d = oracle.lens_ray(cam, float(c[16]), [float(v) for v in fl(r["lens"])], [float(v) for v in fl(r["s"])])
        assert same_bits(o, fl(r["o"])) and same_bits(d, fl(r["d"])), r["seed"]
        assert abs(np.hypot(*fl(r["lens"])) - 0.14) < 1e-6


def test_accumulation_is_a_running_mean(oracle, capi):
    """raytracer_win/main.cpp:292-314 on the default scene: three accumulated frames with different
    sub-pixel samples equal the float-by-float recurrence applied to the three single frames."""
    w, h = 64, 48
    objs, cam = capi.make_scene(capi.RT_SCENE_DEFAULT, w, h)
    rng = np.random.default_rng(3)
    acc = np.zeros((h, w, 4), dtype=np.float32)
    want = np.zeros((h, w, 4), dtype=np.float32)
    for k in range(3):
        off = rng.random((1, 2)).astype(np.float32)
        _, single, _ = oracle.render(objs, cam, capi.render_params(w, h, 5, 1, offsets=off), threads=4)
        single[..., 3] = 1.0
        k1 = np.float32(k)
        want = ((want * k1) + single) * (np.float32(1.0) / (k1 + np.float32(1.0)))
        rgb, acc, _ = oracle.render(objs, cam, capi.render_params(w, h, 5, 1, offsets=off, accumulate=k), threads=4, f32_in=acc)
    assert same_bits(acc, want)
    assert np.array_equal(rgb, np.clip(want[..., :3] * np.float32(255.0), 0, 255).astype(np.uint8))
