#!/usr/bin/env python3
"""Differential fuzzing of the sweeps on random scenes: for every scene the flat sweep, the grouped
sweep and the straight-transcription kernel (RT_AMD_KERNEL=simple, no filter, no grouping: the
reference's loop as written) must produce the same bits -- radiance, per-pixel ray counts, 8-bit
image.  A dropped candidate of the conservative filter or of a bounding-sphere test would show up
as a differing pixel.

    python tests/fuzz_sweeps.py --scenes 50 --seed 1 [--cpu]

--cpu adds the CPU restatement of the reference (oracle/, test infrastructure) as a fourth opinion.

The transcription kernel is selected per context through the environment, so it runs in a child
process (this script with --child) that dumps its frames for the parent to compare.
"""
import argparse
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def random_scene(capi, rng):
    """Spheres in a slab, a blob, a shell or a line (clustered, overlapping, tiny and huge radii),
    0-3 planes, 1-6 emitters, list order shuffled."""
    M, S, P = capi.material, capi.sphere, capi.tiled_plane
    n = int(rng.choice([3, 17, 64, 65, 200, 700, 2500, 5500]))      # (5500: the large-scene instantiation of the grouped sweep, > 64 super-groups)
    layout = rng.choice(["slab", "blob", "shell", "line"])
    scale = float(rng.choice([1.0, 10.0, 300.0]))
    items = []
    for k in range(n):
        if layout == "slab":
            c = rng.uniform(-30, 30, 3) * (1, .03, 1)
        elif layout == "blob":
            c = rng.normal(0, 6, 3)
        elif layout == "shell":
            v = rng.normal(0, 1, 3)
            c = 25 * v / np.linalg.norm(v)
        else:
            c = np.array([k * 0.7 - n * 0.35, np.sin(k * .3), np.cos(k * .2)])
        r = float(rng.choice([.05, .3, .8, 3.0, 12.0], p=[.1, .5, .3, .08, .02]))
        refl = float(rng.choice([0, 0, .3, .9]))
        em = tuple(rng.random(3)) if rng.random() < min(0.5, 4.0 / n) else (0, 0, 0)
        items.append(S(M(tuple(rng.random(3)), tuple(rng.random(3)), refl, em), tuple(c * scale + (0, 5 * scale, -20 * scale)), r * scale))
    if not any(any(it.mat.emissive) for it in items):
        items.append(S(M((1, 1, 1), (0, 0, 0), 0.0, (1, 1, 1)), (0, 40 * scale, 0), 2 * scale))
    for _ in range(int(rng.integers(0, 4))):
        nrm = rng.normal(0, 1, 3)
        items.append(P(tuple(rng.uniform(-5, 5, 3) * scale), tuple(nrm / np.linalg.norm(nrm))))
    order = rng.permutation(len(items))
    items = [items[i] for i in order]
    w, h = int(rng.integers(40, 200)), int(rng.integers(30, 140))
    eye = rng.normal(0, 1, 3) * 30 * scale + (0, 10 * scale, 10 * scale)
    look = np.array([0, 5 * scale, -20 * scale]) - eye
    cam = capi.camera_init(tuple(eye), tuple(look), float(rng.uniform(30, 100)), w, h)
    return capi.objects(items), cam, w, h, int(rng.choice([0, 2, 5, 9])), int(rng.choice([1, 4]))


def render_all(capi, ctx, seed, n_scenes, sweeps, with_cpu_checker=False):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_scenes):
        objs, cam, w, h, depth, spp = random_scene(capi, rng)
        ctx.upload_scene(objs, cam)
        frames = []
        if with_cpu_checker:       # test infrastructure: the CPU restatement of the reference
            from oracle import oracle_lib
            p = capi.render_params(w, h, depth, spp, outputs=capi.RT_OUT_RGBA8 | capi.RT_OUT_F32X4)
            rgb, f32, _ = oracle_lib.render(objs, cam, p, threads=16)
            frames.append(hashlib.sha256(rgb.tobytes() + f32.tobytes()).hexdigest())
        for sweep in sweeps:
            ctx.render(capi.render_params(w, h, depth, spp, outputs=capi.RT_OUT_RGBA8 | capi.RT_OUT_F32X4, sweep=sweep))
            frames.append(hashlib.sha256(ctx.download(capi.RT_FMT_RGB8, w, h).tobytes() +
                                         ctx.download(capi.RT_FMT_F32X4, w, h).tobytes()).hexdigest())
        out.append((len(objs), w, h, depth, spp, frames))
    return out


def check_fast_mode(capi, ctx, seed, n_scenes):
    """RT_MODE_FAST against RT_MODE_EXACT on the same random scenes: same ray counts, radiance within
    1e-4 relative, 8-bit within +-1 LSB on >= 99.99 % of the pixels (north_star's tolerance)."""
    rng = np.random.default_rng(seed)
    bad = 0
    for i in range(n_scenes):
        objs, cam, w, h, depth, spp = random_scene(capi, rng)
        ctx.upload_scene(objs, cam)
        out = []
        for mode in (capi.RT_MODE_EXACT, capi.RT_MODE_FAST):
            ctx.render(capi.render_params(w, h, depth, spp, outputs=capi.RT_OUT_RGBA8 | capi.RT_OUT_F32X4, mode=mode))
            out.append((ctx.download(capi.RT_FMT_RGB8, w, h), ctx.download(capi.RT_FMT_F32X4, w, h)))
        (rgb_e, f_e), (rgb_f, f_f) = out
        a, b = f_e[..., :3].astype(np.float64), f_f[..., :3].astype(np.float64)
        fin = np.isfinite(a) & np.isfinite(b)
        rel = np.max(np.abs(a - b)[fin] / np.maximum(np.abs(a)[fin], 1e-6)) if fin.any() else 0.0
        same_nan = np.array_equal(np.isfinite(a), np.isfinite(b))
        lsb = (np.abs(rgb_e.astype(np.int16) - rgb_f.astype(np.int16)).max(axis=-1) <= 1).mean()
        ok = rel <= 1e-4 and lsb >= 0.9999 and same_nan and np.array_equal(f_e[..., 3], f_f[..., 3])
        bad += not ok
        print(f"scene {i}: {len(objs)} objects {w}x{h} depth {depth} spp {spp}: max rel {rel:.2e}, within 1 LSB {lsb:.6f}: "
              f"{'ok' if ok else 'OUT OF TOLERANCE'}", flush=True)
    print(f"{n_scenes - bad}/{n_scenes} scenes: fast mode within tolerance of exact mode")
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu", action="store_true", help="also compare with the CPU checker under oracle/ (slow on big scenes)")
    ap.add_argument("--fast", action="store_true", help="instead: RT_MODE_FAST against RT_MODE_EXACT at the stated tolerance")
    ap.add_argument("--child", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    import torch  # noqa: F401  (first, so that one HIP runtime serves the process)
    from raytracer_amd import capi
    ctx = capi.Context(0)
    if args.fast:
        return 1 if check_fast_mode(capi, ctx, args.seed, args.scenes) else 0
    if args.child:
        res = render_all(capi, ctx, args.seed, args.scenes, [capi.RT_SWEEP_FLAT])
        with open(args.child, "w") as f:
            for r in res:
                f.write(r[5][0] + "\n")
        return 0
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "simple.txt")
        env = dict(os.environ, RT_AMD_KERNEL="simple", RT_AMD_DEBUG="1")
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--scenes", str(args.scenes), "--seed", str(args.seed),
                               "--child", path], env=env)
        simple = open(path).read().split()
    res = render_all(capi, ctx, args.seed, args.scenes, [capi.RT_SWEEP_FLAT, capi.RT_SWEEP_GROUPED, capi.RT_SWEEP_SORTED], args.cpu)
    bad = 0
    for i, (n, w, h, depth, spp, frames) in enumerate(res):
        ok = all(f == simple[i] for f in frames)
        bad += not ok
        print(f"scene {i}: {n} objects {w}x{h} depth {depth} spp {spp}: {'ok' if ok else 'MISMATCH ' + str([frames[0][:8], frames[1][:8], simple[i][:8]])}", flush=True)
    print(f"{len(res) - bad}/{len(res)} scenes identical across flat, grouped, sorted and transcription kernels"
          + (" and the CPU checker" if args.cpu else ""))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
