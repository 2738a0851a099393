"""The host C++ API (raytracer_amd/host): the reference's class surface on top of the C ABI.

CPU: its classes reproduce the golden fixtures bit for bit; the host program fails loudly without
a device.  GPU: the host program writes the reference program's image.bmp (md5 pins)."""
import ctypes as C
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
MAIN = os.path.join(ROOT, "raytracer_amd", "host", "rt_main")
F3 = C.c_float * 3


def fl(bits):
    return np.asarray(bits, dtype=np.uint32).view(np.float32)


def f3(bits):
    return F3(*[float(x) for x in fl(bits)])


def same_bits(a, b):
    return np.array_equal(np.asarray(a, dtype=np.float32).view(np.uint32), np.asarray(b, dtype=np.float32).view(np.uint32))


@pytest.fixture(scope="module")
def shim(capi):
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libhostshim.so")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "raytracer_amd", "host"),
                           "-o", so, os.path.join(ROOT, "tests", "host_shim.cpp"),
                           "-L" + os.path.join(ROOT, "raytracer_amd"), "-lrt_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "raytracer_amd")])
    return C.CDLL(so)


def test_host_classes_match_reference_fixtures(shim, capi):
    for r in json.load(open(os.path.join(GOLD, "sphere_intersects.json")))["rows"]:
        dist = C.c_float(0.0)
        hit = shim.host_sphere_intersects(f3(r["c"]), C.c_float(float(fl([r["r"]])[0])), f3(r["o"]), f3(r["d"]), C.byref(dist))
        assert hit == r["hit"] and (not hit or same_bits([dist.value], fl([r["dist"]])))
    for r in json.load(open(os.path.join(GOLD, "plane_intersects.json")))["rows"]:
        dist = C.c_float(0.0)
        hit = shim.host_plane_intersects(f3(r["po"]), f3(r["pn"]), f3(r["o"]), f3(r["d"]), C.byref(dist))
        assert hit == r["hit"] and (not hit or same_bits([dist.value], fl([r["dist"]])))
    for r in json.load(open(os.path.join(GOLD, "checker.json")))["rows"]:
        assert shim.host_checker_white(f3(r["pos"])) == r["white"]
    for r in json.load(open(os.path.join(GOLD, "vectors.json")))["rows"]:
        n, rf, rd = F3(), F3(), F3()
        shim.host_vectors(f3(r["c"]), f3(r["p"]), f3(r["i"]), n, rf, rd)
        assert same_bits(list(n), fl(r["normal"])) and same_bits(list(rf), fl(r["ray_from"])) and same_bits(list(rd), fl(r["reflect"]))
    for r in json.load(open(os.path.join(GOLD, "camera.json")))["rows"]:
        for ray in r["rays"]:
            cam, out, s = capi.Camera(), F3(), fl(ray["s"])
            shim.host_camera(f3(r["pos"]), f3(r["dir"]), C.c_float(float(fl([r["fov"]])[0])), r["w"], r["h"], C.byref(cam),
                             C.c_float(float(s[0])), C.c_float(float(s[1])), out)
            got = (list(cam.position) + list(cam.view_direction) + list(cam.right) + list(cam.up) +
                   [cam.aspect_ratio, cam.half_angle, cam.image_width, cam.image_height])
            assert same_bits(got, fl(r["camera"])) and same_bits(list(out), fl(ray["dir"]))


def test_host_flatten(shim, capi):
    s, p = capi.Object(), capi.Object()
    shim.host_flatten(C.byref(s), C.byref(p))
    assert s.type == capi.RT_OBJ_SPHERE and list(s.a) == [1.0, 2.0, 3.0] and s.radius == 4.0
    assert same_bits(list(s.mat.albedo) + list(s.mat.specular) + [s.mat.reflectance] + list(s.mat.emissive),
                     [.1, .2, .3, .4, .5, .6, .7, .8, .9, 1.0])
    ref_plane = capi.make_scene(capi.RT_SCENE_DEFAULT, 8, 8)[0][5]
    assert bytes(p) == bytes(ref_plane)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_host_program_fails_loudly_without_gpu(tmp_path):
    r = subprocess.run([MAIN, "-a", "0"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "no CPU path" in r.stderr
    assert not (tmp_path / "image.bmp").exists()
    r = subprocess.run([MAIN, "--bogus"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 2 and "partitions" in r.stderr and "antialiased" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("args,key", [(["-a", "0"], "a0"), ([], "a1"), (["-p", "8", "--antialiased=1"], "a1"),
                                      (["-g", "0,0,0"], "a1")])       # the frame shared three ways (rt_create_multi)
def test_host_program_writes_reference_image(tmp_path, args, key):
    """raytracer_cpp's CLI contract: defaults -p 2 -a 1, image.bmp in the working directory, whose
    md5 is the one the verbatim program produced."""
    pins = json.load(open(os.path.join(GOLD, "trace_pins.json")))
    r = subprocess.run([MAIN] + args, cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("Time: ")
    data = open(tmp_path / "image.bmp", "rb").read()
    assert hashlib.md5(data).hexdigest() == pins["bmp_md5"][key]


SCENE_TXT = """# three balls, a light and a tinted floor
camera      0 5 9   0 -0.55 -1   55
sphere      0 1.5 0 1.5     0.7 0.1 0.1   0.9 0.1 0.1   0.5
sphere      -3 1 1.5 1      0.1 0.6 0.2   0.3 0.9 0.3   0.25
tiledplane  0 0 0   0 2 0     1 0.9 0.8  1 1 1  0.4     0.1 0.1 0.2  0 0 0  0.7
sphere      2.5 0.6 2 0.6   1 1 1   0 0 0   0    1 0.8 0.6     # emits
sphere      -6 9 6 0.5      0 0 0   0 0 0   0    1 1 1
"""


def test_scene_file_parsers_agree_and_reject_garbage(capi, tmp_path):
    path = tmp_path / "scene.txt"
    path.write_text(SCENE_TXT)
    objs, cam = capi.load_scene_file(str(path), 320, 200)
    assert len(objs) == 5 and [o.type for o in objs] == [0, 0, 1, 0, 0]
    assert list(objs[2].b) == [0.0, 1.0, 0.0] and abs(objs[2].mat2.reflectance - 0.7) < 1e-6
    assert list(objs[3].mat.emissive) == [1.0, np.float32(0.8), np.float32(0.6)]
    (tmp_path / "bad.txt").write_text("camera 0 0 0 0 0 -1 60\nsphere 1 2 3\n")
    with pytest.raises(ValueError):
        capi.load_scene_file(str(tmp_path / "bad.txt"), 8, 8)
    r = subprocess.run([MAIN, "-s", str(tmp_path / "bad.txt")], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 2 and "bad.txt:2" in r.stderr
    r = subprocess.run([MAIN, "-s", str(tmp_path / "missing.txt")], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 2 and "cannot open" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["-w", "1"], ["-w", "2"]])
def test_host_program_renders_scene_file(capi, oracle, tmp_path, extra):
    """rt_main -s FILE: the scene goes through the host's Sphere/TiledPlane/Camera classes; the BMP
    must be the one the oracle renders from the Python parse of the same file."""
    path = tmp_path / "scene.txt"
    path.write_text(SCENE_TXT)
    w, h = 333, 211
    r = subprocess.run([MAIN, "-s", str(path), "-W", str(w), "-H", str(h), "-a", "1", "-o", "out.bmp"] + extra,
                       cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    objs, cam = capi.load_scene_file(str(path), w, h)
    rgb, _, _ = oracle.render(objs, cam, capi.render_params(w, h, 5, 4), threads=8)
    want = tmp_path / "want.bmp"
    oracle.write_bmp(str(want), rgb)
    assert open(tmp_path / "out.bmp", "rb").read() == open(want, "rb").read()


def _lcg_unit(state):
    state[0] = (state[0] * 1664525 + 1013904223) & 0xFFFFFFFF
    return np.float32(state[0] >> 8) * np.float32(1.0 / 16777216.0)


@pytest.mark.gpu
def test_host_program_progressive_depth_of_field(capi, oracle, tmp_path):
    """rt_main --frames 5 --lens 0.14 --focus 10: raytracer_win's progressive mode (jittered single
    samples, a lens point per frame, running mean) on the default scene; the BMP equals the oracle's
    for the same documented LCG sequence."""
    import math
    w, h, frames = 320, 200, 5
    r = subprocess.run([MAIN, "-W", str(w), "-H", str(h), "-f", str(frames), "-l", "0.14", "-F", "10", "-o", "dof.bmp"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    objs, cam = capi.make_scene(capi.RT_SCENE_DEFAULT, w, h)
    st, acc = [0x5EED1E25], np.zeros((h, w, 4), dtype=np.float32)
    for k in range(frames):
        jit = [_lcg_unit(st), _lcg_unit(st)]
        a = 6.283185307179586 * float(_lcg_unit(st))
        lens = [np.float32(math.cos(a)) * np.float32(0.14), np.float32(math.sin(a)) * np.float32(0.14)]
        p = capi.render_params(w, h, 5, 1, offsets=[jit], lens=[lens], focal_length=10.0, accumulate=k,
                               outputs=capi.RT_OUT_RGBA8 | capi.RT_OUT_F32X4)
        rgb, acc, _ = oracle.render(objs, cam, p, threads=8, f32_in=acc)
    want = tmp_path / "want.bmp"
    oracle.write_bmp(str(want), rgb)
    assert open(tmp_path / "dof.bmp", "rb").read() == open(want, "rb").read()
