"""CPU tests of the C-ABI library itself: it loads, exports every symbol include/rt_api.h declares,
its host-side helpers work without a device, and device entry points fail loudly (no CPU path)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "rt_api.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", text)))


def test_exports_match_header(capi):
    declared = _declared_symbols()
    assert len(declared) >= 17
    for name in declared:
        assert hasattr(capi.lib, name), f"librt_amd.so does not export {name}"
    assert sorted(capi.EXPORTS) == declared
    assert capi.lib.rt_api_version() == 3


def test_no_oracle_in_product():
    """The product must not reach into oracle/ (the checker) in any form."""
    for base, _, files in os.walk(os.path.join(ROOT, "raytracer_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp", "Makefile")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle" not in text.lower().replace("# oracle", ""), f"{f} mentions the oracle"


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_device_fails_loudly(capi):
    with pytest.raises(capi.RtError) as e:
        capi.Context(0)
    assert e.value.code == capi.RT_ERR_NO_DEVICE
    assert "no CPU path" in str(e.value)


def test_null_arguments(capi):
    assert capi.lib.rt_create(None, 0) == capi.RT_ERR_INVALID_ARG
    assert capi.lib.rt_upload_scene(None, None, 0, None) == capi.RT_ERR_INVALID_ARG
    assert capi.lib.rt_render(None, None) == capi.RT_ERR_INVALID_ARG
    assert capi.lib.rt_download_framebuffer(None, None, 0, 0) == capi.RT_ERR_INVALID_ARG
    capi.lib.rt_destroy(None)     # must be a no-op


def test_default_scene_is_initscene(capi):
    """InitScene, raytracer_cpp/main.cpp:21-63."""
    objs, cam = capi.make_scene(capi.RT_SCENE_DEFAULT, 1024, 768)
    assert len(objs) == 6
    assert [o.type for o in objs] == [0, 0, 0, 0, 0, 1]
    assert [round(o.radius, 3) for o in objs[:5]] == [2.0, 1.0, 0.5, 0.8, 0.4]
    assert list(objs[3].mat.emissive) == [1.0, 1.0, 0.0] and list(objs[4].mat.emissive) == [1.0, 1.0, 1.0]
    assert all(list(o.mat.emissive) == [0.0, 0.0, 0.0] for o in objs[:3])
    assert list(objs[4].a) == [np.float32(-10.8), np.float32(8.4), 10.0]
    assert list(cam.position) == [0.0, 6.0, 8.0] and cam.image_width == 1024.0 and cam.image_height == 768.0


def test_grid_scene_shapes(capi):
    for kind, n, ne in ((capi.RT_SCENE_GRID1K, 1029, 4), (capi.RT_SCENE_GRID16K, 16393, 8)):
        objs, _ = capi.make_scene(kind, 640, 360, frame=3)
        assert len(objs) == n
        assert objs[n - 1].type == capi.RT_OBJ_TILED_PLANE
        assert sum(1 for o in objs if any(o.mat.emissive)) == ne
    assert capi.lib.rt_make_scene(99, 8, 8, 0, 1, None, 0, None) == capi.RT_ERR_INVALID_ARG


def test_split_bands_properties(capi):
    rng = np.random.default_rng(5)
    for height, n in ((2160, 8), (2160, 1), (7, 8), (8, 8), (1080, 3), (33, 4)):
        cost = rng.uniform(0.0, 5.0, height).astype(np.float32)
        cost[: height // 3] *= 10
        for c in (None, cost):
            b = capi.split_bands(height, n, c)
            assert b[0] == 0 and b[-1] == height and len(b) == n + 1
            assert all(b[i] <= b[i + 1] for i in range(n))
            if height >= n:
                assert all(b[i] < b[i + 1] for i in range(n)), (height, n, b)
        b = capi.split_bands(height, n, cost)
        if height >= 20 * n:
            band = [float(cost[b[i]:b[i + 1]].sum()) for i in range(n)]
            assert max(band) <= float(cost.sum()) / n + 2 * float(cost.max()) + 1e-3
    assert capi.split_bands(0, 2, None) == [0, 0, 0]
    with pytest.raises(capi.RtError):
        capi.split_bands(4, 0, None)


def test_ctypes_mirror_matches_the_header(capi, tmp_path):
    """raytracer_amd/capi.py restates the structs of include/rt_api.h by hand: sizes and field offsets
    must be what a C compiler makes of the header."""
    import ctypes as C
    import subprocess
    structs = {"rt_material": capi.Material, "rt_object": capi.Object, "rt_camera": capi.Camera,
               "rt_render_params": capi.RenderParams, "rt_stats": capi.Stats}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "rt_api.h"', 'int main(void) {']
    for cname, cls in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['return 0; }']
    src = tmp_path / "abi.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(root, "include"), "-o", str(exe), str(src)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


def _grouping_invariants(capi, objs):
    gb, sb, of, sigma = capi.plan_groups(objs, with_metric=True)
    gb64, sb64 = gb.astype(np.float64), sb.astype(np.float64)
    assert sigma.min() == 1.0 and set(sigma) <= {1.0, 2.0, 4.0}
    kappa = sigma.max()
    dirs = np.random.default_rng(5).normal(0, 1, (120, 3))
    dirs = np.concatenate([dirs / np.linalg.norm(dirs, axis=1, keepdims=True), np.eye(3), -np.eye(3)])
    sph = [k for k, o in enumerate(objs) if o.type == capi.RT_OBJ_SPHERE]
    # emissive spheres are kept apart: up to 8 of them form the tail that every sweep tests (no group, no bound);
    # more than 8 are grouped among themselves
    finite = lambda o: np.isfinite(list(o.a)).all() and np.isfinite(o.radius)
    lights = [k for k in sph if any(objs[k].mat.emissive) and finite(objs[k])]
    if len(lights) <= 8:
        assert all(of[k] == -1 for k in lights)
        sph = [k for k in sph if k not in set(lights)]
    else:
        for k in lights:
            assert all(any(objs[j].mat.emissive) for j in np.nonzero(of == of[k])[0])
    assert all(of[k] == -1 for k in range(len(objs)) if k not in set(sph))
    counts = np.bincount(of[sph], minlength=len(gb))
    assert counts.min() >= 1 and counts.max() <= 8 and len(sb) == (len(gb) + 7) // 8
    for k in sph:                        # every sphere lies inside its group's bound {x : |sigma (x - g)| <= R} ...
        o, g = objs[k], of[k]
        c, r = np.array(list(o.a), dtype=np.float64), abs(float(o.radius))
        if not (np.isfinite(c).all() and np.isfinite(r)):
            assert counts[g] == 1        # a sphere nothing can hit sits in a group of its own
            continue
        pts = c + r * dirs               # points of the sphere: random ones and the axis extremes
        s = g // 8                       # ... and inside its super-group's
        for b in (gb64[g], sb64[s]):
            reach = np.linalg.norm(sigma * (pts - b[:3]), axis=1).max()
            assert reach <= b[3] * (1 + 1e-6) or np.isinf(b[3])
            assert b[3] <= (np.linalg.norm(sigma * (c - b[:3])) + kappa * r) * (1 + 1e-5) or counts[g] > 1 or b is not gb64[g]
    return gb, sb, of, sigma


def test_grouping_invariants(capi):
    """rt_plan_groups (the grouping behind RT_SWEEP_GROUPED, host only): every sphere in exactly one
    group of at most 8, bounds that contain their members at both levels, outliers and unhittable
    spheres on their own, planes in no group; and the grouping actually is local."""
    objs, _ = capi.make_scene(capi.RT_SCENE_GRID1K, 64, 64)
    gb, sb, of, sigma = _grouping_invariants(capi, objs)
    assert list(sigma) == [1.0, 4.0, 1.0]          # spheres on a floor: bounds flattened along y
    assert 128 <= len(gb) <= 160 and np.median(gb[:, 3]) < 4.5    # 2x4 blocks of a 1.5-pitch grid
    lights = [k for k, o in enumerate(objs) if any(o.mat.emissive)]
    assert len(lights) == 4 and all(of[k] == -1 for k in lights)  # the 4 lights: the tail, in no group
    objs16, _ = capi.make_scene(capi.RT_SCENE_GRID16K, 64, 64)
    gb16, _, _, sigma16 = _grouping_invariants(capi, objs16)
    assert sigma16[1] == 4.0 and sigma16[2] == 1.0     # (its groups are 2 x 4 blocks, long in z: x may get 2)
    assert len(gb16) < 2400
    rng = np.random.default_rng(9)
    M, S, P = capi.material, capi.sphere, capi.tiled_plane
    m = M((.5, .5, .5), (.5, .5, .5), 0.2)
    odd = [S(m, tuple(rng.normal(0, 5, 3)), float(rng.uniform(.1, 1))) for _ in range(333)]
    odd += [S(m, (1e6, 0, 0), 5e5), S(m, (float("nan"), 0, 0), 1.0), S(m, (0, 0, 0), float("inf")), P((0, 0, 0), (0, 1, 0)),
            S(m, (0, 1, 0), 1.0)] + [S(m, (2, 2, 2), 1.0) for _ in range(40)]
    _grouping_invariants(capi, capi.objects(odd))          # (whatever shape the bounds get, they must contain their members)
    glow = M((1, 1, 1), (0, 0, 0), 0.0, (1, 1, 1))
    many = odd[:100] + [S(glow, tuple(rng.normal(0, 30, 3)), 0.5) for _ in range(19)]
    _grouping_invariants(capi, capi.objects(many))         # 19 emissive spheres: groups of their own, under bounds
    few = odd[:100] + [S(glow, tuple(rng.normal(0, 30, 3)), 0.5) for _ in range(8)] + [S(glow, (float("nan"), 0, 0), 1.0)]
    _grouping_invariants(capi, capi.objects(few))          # 8 (and an unhittable ninth): the tail
    one = capi.objects([P((0, 0, 0), (0, 1, 0))])
    gb0, sb0, of0 = capi.plan_groups(one)
    assert len(gb0) == 0 and len(sb0) == 0 and list(of0) == [-1]
