#!/usr/bin/env python3
"""bench.py -- Mrays/s of the trace loop on MI355X, with roofline and a CPU baseline.

A step = one frame of the workload: the synthetic "grid-1024" scene (1024 jittered spheres + 4
emissive spheres + tiled plane, SURVEY.md 8d config 4's scene) at 3840x2160, MAX_DEPTH 5 (the
metric's "4K depth 5"), antialiased with the reference's 4 fixed samples per pixel (its CLI default,
-a 1, raytracer_cpp/main.cpp:234).  With N ranks the frame is cut into N cost-balanced row bands
(one per GPU) and the RGBA8 bands are gathered to rank 0 over RCCL: strong scaling.

The sweep is the library default (RT_SWEEP_AUTO: the grouped sweep for this scene).  The roofline
object prices the FLOPs of the tests that kernel executed; "roofline_flat" is the same frame through
the flat every-ray-tests-every-sphere sweep, measured in the same run (N = 1).

With N > 1 each rank keeps two frames in flight ("lanes": two contexts, streams and band buffers used
alternately) so that the gather and assembly of frame k run under the render of frame k + 1; every
frame is still rendered, gathered and assembled inside the timed region.  N = 1 runs one lane, which
keeps the per-launch kernel time of the roofline free of overlap (`--lanes 2` there gives ~7 % more
frames per second by filling one frame's tail with the next frame's waves).

Renders are enqueued on a torch stream (rt_set_stream) so that kernel, gather and assembly order
on the device without host round trips; the trace kernel's duration is measured per launch with HIP
events on that stream over the timed region (rt_stats.kernel_ms_total).

Rays = primary + reflection + shadow (one per hit x emissive object), counted once by a
stats-enabled render outside the timed region.  Inputs (scene, camera) are resident in HBM before
the timed region; the framebuffer stays on the device.

Launch: python bench.py [--gpus N --steps K --warmup W]; for N > 1 under torch.distributed.run.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_VALU_TFLOPS = 157.3   # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
FLOP_PER_SPHERE_TEST = 15       # 3 sub + 5 (dot) + 5 (dot) + mul + sub to the reject
FLOP_PER_PLANE_TEST = 5


def profiled_counters(path, kernel):
    """{counter: value per launch}, scratch bytes of the profiled kernel, from a committed rocprofv3 PMC summary
    (tools/pmc_profile.sh / tools/pmc_instmix.sh) -- ({}, None) when the file or the kernel is not there."""
    rows, on, scratch = {}, False, None
    try:
        for line in open(path):
            if line.startswith('"# ') or line.startswith("# "):
                if "vgpr,agpr" in line:
                    if on:
                        scratch = int(line.split("=")[1].split(",")[4])
                else:
                    on = kernel in line
            elif on and "," in line:
                k, v = line.split(",")[:2]
                try:
                    rows[k] = float(v)
                except ValueError:
                    pass
    except (OSError, ValueError, IndexError):
        return {}, None
    return rows, scratch


def profiled_traffic(path, kernel):
    """(bytes per launch, scratch bytes of the profiled kernel, VALU busy fraction) from a committed rocprofv3 PMC summary of
    this same command (tools/pmc_profile.sh), plus the VALU pipes' busy fraction: FETCH_SIZE and WRITE_SIZE are in KiB; MI355X_MICROARCH.md's
    gfx950 correction doubles FETCH_SIZE for wide streaming reads -- here most of the traffic is 4-byte
    private-memory (spill/record) accesses, a width the guide calls uncalibrated, so the uncorrected
    sum is reported and the note says so.  (None, None) when the summary is not there."""
    try:
        rows, on, scratch = {}, False, None
        for line in open(path):
            if line.startswith('"# ') or line.startswith("# "):
                if "vgpr,agpr" in line:
                    if on:
                        scratch = int(line.split("=")[1].split(",")[4])
                else:
                    on = kernel in line
            elif on and "," in line:
                k, v = line.split(",")[:2]
                rows[k] = v
        busy = None
        if "SQ_ACTIVE_INST_VALU" in rows and "GRBM_GUI_ACTIVE" in rows:
            # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the 1024 SIMDs, GRBM_GUI_ACTIVE cycles summed over the 8 XCDs
            busy = 4.0 * float(rows["SQ_ACTIVE_INST_VALU"]) / 1024.0 / (float(rows["GRBM_GUI_ACTIVE"]) / 8.0)
        return (float(rows["FETCH_SIZE"]) + float(rows["WRITE_SIZE"])) * 1024.0, scratch, busy
    except (OSError, KeyError, ValueError, IndexError):
        return None, None, None


def built_scratch_bytes(unit, mangled_args):
    """Private-segment bytes of a kernel of the library as built (raytracer_amd/kernel_resources.json, written
    by the Makefile from the code objects' metadata notes)."""
    try:
        for k in json.load(open(os.path.join(ROOT, "raytracer_amd", "kernel_resources.json")))[unit]:
            if mangled_args in k["name"]:
                return int(k["scratch"])
    except (OSError, KeyError, ValueError):
        pass
    return None


def reference_as_written(capi, width, height, depth, spp, kind, threads):
    """The reference's OWN object code (oracle/_ref/libref_trace_d<depth>.so: raytracer_cpp/main.cpp compiled in
    the build container, see oracle/Makefile) timed on this host: (1) the program's own case -- InitScene +
    DrawScene at 1024x768, -a 1, -p 2 (its defaults), timed like main.cpp:246-253; (2) this bench's workload
    through its TraceRay as written (every object a candidate emitter: O(N^2) tests per hit, by-value
    shared_ptr copies in FindNearestObject) on a bounded sample of pixels.  Rays are counted as the loop as
    written casts them: primary + reflection + one shadow ray per (hit, object).  When the library is absent
    (no /root/reference at build time) the figures SURVEY.md section 6 measured in the build container stand in."""
    import numpy as np
    from oracle import oracle_lib
    ref = oracle_lib.load_ref_trace(depth)
    if ref is None:
        return {"kind": "survey", "note": "oracle/_ref not built; SURVEY.md section 6, build container (8 vCPU): default scene 1024x768 "
                "-a 1: 2581 ms at -p 1, 3199 ms at -p 2 (the program's default), 3583 ms at -p 8 = 9.1 Mrays/s as written at -p 2"}
    out = {"kind": "reference", "library": "oracle/_ref/libref_trace_d%d.so" % depth}
    if depth == 5:
        ref.init_scene()
        t0 = time.perf_counter()
        ref.draw_scene(None, 2, 1)
        dt = time.perf_counter() - t0
        # rays of that frame as written (tests/golden/trace_pins.json: 6 752 898 TraceRay calls, 3 791 785 hits x 6 objects)
        rays = 6752898 + 3791785 * 6
        out["program_default"] = {"ms": dt * 1e3, "value": rays / dt / 1e6, "unit": "Mrays/s", "cores": 2,
                                  "sample": "InitScene + DrawScene verbatim, 1024x768, -a 1, -p 2 (main.cpp:232-248): %d rays as written" % rays}
    # this workload through TraceRay as written, on random pixels, `threads` threads
    objs, cam = capi.make_scene(kind, width, height)
    ref.set_scene(objs, cam, counting=True)
    rng = np.random.default_rng(1)
    n = 16
    total_t, total_rays, total_px = 0.0, 0, 0
    while total_t < 8.0 and total_px < 65536:
        px = np.stack([rng.integers(0, width, n), rng.integers(0, height, n)], axis=1).astype(np.float32)
        sxy = (px[:, None, :] + np.asarray([0.1, 0.2, 0.6, 0.5, 0.8, 0.7, 0.2, 0.8], dtype=np.float32).reshape(4, 2)[None, :spp, :]).reshape(-1, 2)
        ref.reset_counters()
        t0 = time.perf_counter()
        ref.trace_samples(sxy, threads=threads)
        total_t += time.perf_counter() - t0
        c = ref.counters()
        n_obj = len(objs)
        calls = (c["intersects"] - n_obj * n_obj * c["normals"]) // n_obj
        total_rays += calls + c["ray_from"]           # TraceRay calls + one shadow ray per (hit, object)
        total_px += n
        n = min(4096, n * 2)
    out["workload_sample"] = {"value": total_rays / total_t / 1e6, "unit": "Mrays/s", "cores": threads,
                              "sample": "%d random pixels x %d spp of this workload through TraceRay as written: %d rays in %.2f s"
                                        % (total_px, spp, total_rays, total_t)}
    return out


def cpu_baseline(capi, width, height, depth, spp, kind, offsets=None):
    """Times the CPU oracle (a port of the reference loop) on a bounded sample of the workload:
    the same scene and camera at 1/2 resolution per axis."""
    from oracle import oracle_lib
    lw, lh = width // 2, height // 2
    objs, cam = capi.make_scene(kind, lw, lh)
    threads = max(1, min(16, os.cpu_count() or 1))
    p = capi.render_params(lw, lh, depth, spp, offsets=offsets)
    t0 = time.perf_counter()
    _, _, st = oracle_lib.render(objs, cam, p, threads=threads, want_rgb8=True, want_f32=False, want_stats=True)
    dt = time.perf_counter() - t0
    rays = st["primary_rays"] + st["reflection_rays"] + st["shadow_rays_pruned"]
    # and on one core (SURVEY 8d asks for both), on a sample 1/16 the size
    sw, sh = max(8, lw // 4), max(8, lh // 4)
    objs1, cam1 = capi.make_scene(kind, sw, sh)
    p1 = capi.render_params(sw, sh, depth, spp, offsets=offsets)
    t1 = time.perf_counter()
    _, _, st1 = oracle_lib.render(objs1, cam1, p1, threads=1, want_rgb8=True, want_f32=False, want_stats=True)
    dt1 = time.perf_counter() - t1
    rays1 = st1["primary_rays"] + st1["reflection_rays"] + st1["shadow_rays_pruned"]
    as_written = reference_as_written(capi, width, height, depth, spp, kind, threads)
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "reference_as_written": as_written,
            "sample": f"same scene/camera at {lw}x{lh} (1/4 of the pixels), depth {depth}, {spp} spp, "
                      f"{rays} rays in {dt:.2f} s, emitter loop pruned to emissive objects",
            "one_core": {"value": rays1 / dt1 / 1e6, "unit": "Mrays/s", "cores": 1,
                         "sample": f"{sw}x{sh}, {rays1} rays in {dt1:.2f} s"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--spp", type=int, default=4)
    ap.add_argument("--scene", default="grid1k", choices=["default", "grid1k", "grid16k"])
    ap.add_argument("--mode", default="exact", choices=["exact", "fast"])
    ap.add_argument("--sweep", default="auto", choices=["auto", "flat", "grouped", "sorted"],
                    help="flat: every ray tests every sphere; grouped: bounding-sphere groups of 8 are skipped when "
                         "missed; auto (the library default): grouped from 64 spheres up")
    ap.add_argument("--lanes", type=int, default=0,
                    help="frames in flight per rank (each lane: its own context, stream and band buffer); 0 = 1 for one "
                         "GPU, 2 for N > 1 so that the gather of frame k runs under the render of frame k + 1")
    ap.add_argument("--plan", default="auto", choices=["auto", "bands", "strips"],
                    help="N > 1: 8-column strips dealt round-robin to the ranks (what auto picks when the width is a multiple "
                         "of 8 x gpus: every rank gets the same share of every row, slowest of 8 parts 1.05x the mean), or "
                         "contiguous cost-balanced row bands (1.12x; auto's fallback)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N > 1 path with all ranks sharing GPU 0 (bands staged through the host)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from raytracer_amd import capi, multigpu

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the trace loop has no CPU fallback")
    rehearsal = args.backend == "gloo"
    if rehearsal:
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    kind = {"default": capi.RT_SCENE_DEFAULT, "grid1k": capi.RT_SCENE_GRID1K, "grid16k": capi.RT_SCENE_GRID16K}[args.scene]
    mode = capi.RT_MODE_EXACT if args.mode == "exact" else capi.RT_MODE_FAST
    sweep = {"auto": capi.RT_SWEEP_AUTO, "flat": capi.RT_SWEEP_FLAT, "grouped": capi.RT_SWEEP_GROUPED, "sorted": capi.RT_SWEEP_SORTED}[args.sweep]
    W, H = args.width, args.height
    objs, cam = capi.make_scene(kind, W, H)
    ctx = capi.Context(local)
    ctx.upload_scene(objs, cam)

    # the plan is identical on every rank without communication: contiguous row bands balanced by a
    # deterministic 1/16-resolution pre-pass, or strips of 8 columns dealt round-robin (every rank gets
    # the same share of every row; no pre-pass, but measured no better: DESIGN.md section 5)
    strips = world > 1 and (args.plan == "strips" or (args.plan == "auto" and args.width % (8 * world) == 0))
    out_w = W
    if strips:
        bounds = None
        part = dict(interleave=(world, rank))
        max_rows = H
        out_w = capi.interleaved_count(W, world, 0)          # the widest share: equal counts for the gather
        if capi.interleaved_count(W, world, rank) != out_w:
            raise SystemExit("bench.py --plan strips wants equal shares: width must be a multiple of 8 x gpus")
    else:
        if world > 1:
            cost = multigpu.prepass_row_cost(ctx, objs, cam, W, H, args.depth)
            bounds = multigpu.plan_bands(cost, H, world)
        else:
            bounds = [0, H]
        part = dict(rows=(bounds[rank], bounds[rank + 1]))
        max_rows = max(bounds[i + 1] - bounds[i] for i in range(world))
    strip_index = {}
    band = torch.zeros((max_rows, out_w), dtype=torch.int32, device=dev)   # RGBA8 words
    ctx.bind_framebuffer(capi.RT_OUT_RGBA8, band.data_ptr(), band.numel() * 4)
    frame = torch.zeros((H, W), dtype=torch.int32, device=dev) if rank == 0 else None

    # 1 and 4 spp use the reference's fixed pattern; other counts the stratified pattern of config 3
    offsets = None if args.spp in (1, 4) else capi.stratified_offsets(args.spp)

    # ray census of this rank's band (untimed)
    ctx.render(capi.render_params(W, H, args.depth, args.spp, offsets=offsets, mode=mode, stats=True, sweep=sweep, **part))
    st = ctx.stats()
    n_sph = sum(1 for o in objs if o.type == capi.RT_OBJ_SPHERE)
    n_pl = len(objs) - n_sph
    my_rays = st["primary_rays"] + st["reflection_rays"] + st["shadow_rays"]
    # FLOPs of the tests the kernel EXECUTED: for the flat sweep that is rays x spheres (SURVEY 8d's
    # lower bound); the grouped sweep skips most of them and pays bounding-sphere tests instead (the
    # same 15-FLOP line-distance expression), so its figure is what ran, never the brute-force count
    my_flop = (st["sphere_tests_run"] + st["group_tests_run"]) * FLOP_PER_SPHERE_TEST + my_rays * n_pl * FLOP_PER_PLANE_TEST
    brute_flop = my_rays * (n_sph * FLOP_PER_SPHERE_TEST + n_pl * FLOP_PER_PLANE_TEST)
    grouped = st["group_tests_run"] > 0
    sorted_levels = st.get("sweep") == capi.RT_SWEEP_SORTED      # RT_SWEEP_AUTO resolved (rt_stats.sweep)

    params = capi.render_params(W, H, args.depth, args.spp, offsets=offsets, mode=mode, sweep=sweep, **part)

    # Lanes: frames in flight on this rank.  A lane owns a context (a context's renders share scratch and
    # must follow one another on one stream), a stream, a band buffer and, on rank 0, a frame.  With two
    # lanes the gather and assembly of frame k overlap the render of frame k + 1; the collectives are
    # still issued in the same order on every rank.
    n_lanes = args.lanes if args.lanes > 0 else (2 if world > 1 else 1)
    lanes = []
    for i in range(n_lanes):
        c = ctx
        b, f = band, frame
        if i > 0:
            c = capi.Context(local)
            c.upload_scene(objs, cam)
            b = torch.zeros_like(band)
            c.bind_framebuffer(capi.RT_OUT_RGBA8, b.data_ptr(), b.numel() * 4)
            f = torch.zeros_like(frame) if rank == 0 else None
            c.render(params)                                # untimed: the lane has rendered once (stats, scratch)
        s_ = torch.cuda.Stream(device=dev)
        c.set_stream(s_.cuda_stream)
        lanes.append((c, s_, b, f))
    issued = [0]
    if strips and rank == 0 and not rehearsal:
        for _, _, b_, _ in lanes:                           # the gather's receive buffers exist before anything is timed
            multigpu.strip_receive_buffer(strip_index, b_, world)

    def step():
        c, stream, b, f = lanes[issued[0] % n_lanes]
        issued[0] += 1
        with torch.cuda.stream(stream):
            c.render(params, wait=False)                    # enqueue only
            if world == 1:
                f.copy_(b, non_blocking=True)               # the N = 1 "assembly"
            if world > 1 and not rehearsal:
                if strips:
                    multigpu.gather_strips(b, W, rank, world, out=f, index_cache=strip_index)
                else:
                    multigpu.gather_bands(b, bounds, rank, world, out=f)
        if world > 1 and rehearsal:
            stream.synchronize()
            if strips:
                host = multigpu.gather_strips(b.cpu(), W, rank, world, index_cache=strip_index)
            else:
                host = multigpu.gather_bands(b.cpu(), bounds, rank, world)
            if rank == 0:
                f.copy_(host)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def lane_stats():
        sts = [c.stats() for c, _, _, _ in lanes]
        return sum(x["kernels_timed"] for x in sts), sum(x["kernel_ms_total"] for x in sts)

    for _ in range(args.warmup):
        step()
    fence()
    before = lane_stats()
    fence()
    allocs0 = torch.cuda.memory_stats(dev).get("allocation.all.allocated", 0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    timed_allocs = torch.cuda.memory_stats(dev).get("allocation.all.allocated", 0) - allocs0
    after = lane_stats()
    assert after[0] - before[0] == args.steps
    kernel_ms = after[1] - before[1]
    frame = lanes[(issued[0] - 1) % n_lanes][3]             # the last frame assembled

    tot = torch.tensor([elapsed, float(my_rays), float(my_flop), kernel_ms / max(1, args.steps)],
                       dtype=torch.float64, device="cpu" if rehearsal else dev)
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, rays, flop, k_ms = mx[0].item(), sm[1].item(), sm[2].item(), mx[3].item()
    else:
        elapsed, rays, flop, k_ms = tot[0].item(), tot[1].item(), tot[2].item(), tot[3].item()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        # roofline of the dominant (only) kernel on rank 0's launch: algorithmic FLOP / event time
        k0_ms = kernel_ms / max(1, args.steps)
        achieved = (my_flop / (k0_ms * 1e-3)) / 1e12 if k0_ms > 0 else 0.0
        out = {
            "metric": "Mrays/s (primary+secondary) at 4K depth 5",
            "value": rays * args.steps / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.scene}: {len(objs)} objects ({n_sph} spheres, {st['n_emitters']} emissive), "
                                   f"{W}x{H}, MAX_DEPTH {args.depth}, {args.spp} spp, mode {args.mode}, sweep {args.sweep}" + (" (sorted)" if sorted_levels and args.sweep == "auto" else ""),
                       "rays_per_frame": int(rays), "bands": bounds,
                       "rank0_ray_classes": {k: int(st[k]) for k in ("primary_rays", "reflection_rays", "shadow_rays", "hits", "lit",
                                                                         "sphere_tests", "sphere_tests_run", "group_tests_run", "exact_tests_run")},
                       "lanes": n_lanes,
                       "device_allocations_in_timed_region_rank0": int(timed_allocs),
                       "parallelism": (f"8-column strips round-robin x{world}" if strips else f"row-bands x{world}") + ((" + gloo gather (rehearsal)" if rehearsal else " + RCCL gather") if world > 1 else "")},
            "roofline": {"bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_VALU_TFLOPS,
                         "traffic": None,
                         "achieved_kind": "algorithmic (reference-equivalent) FLOP/s: 15 FLOP per executed (ray, sphere) or (ray, bound) test as "
                                          "SURVEY 8d prices the reference's test; the packed filter itself performs about 8 FLOP per (ray, sphere) "
                                          "plus 10 shared by the K rays of a sphere pair, so this is not a VALU-issue fraction -- "
                                          "valu_busy_profiled is",
                         "kernel_ms": k0_ms,
                         "algorithmic_flop_per_launch": int(my_flop),
                         "kernel": ("sorted_level_kernel x levels + far segments + sorted_shade_kernel (sorted sweep: one launch per bounce level; "
                                    "kernel_ms spans them all)" if sorted_levels else
                                    "trace_sweep_kernel (" + ("grouped" if grouped else "flat") + " sweep)"),
                         "brute_force_equivalent_tflops": (brute_flop / (k0_ms * 1e-3)) / 1e12 if k0_ms > 0 else 0.0,
                         # the framebuffer is the only HBM traffic the algorithm asks for: 4 B per pixel per launch
                         "hbm_write_gbps": (band.numel() * 4 / (k0_ms * 1e-3)) / 1e9 if k0_ms > 0 else 0.0,
                         "lanes_note": None if n_lanes == 1 else "kernel_ms is per launch with %d frames in flight: launches overlap on "
                                       "the device, so it overstates the time a launch would take alone" % n_lanes,
                         "note": "15 FLOP x (ray,sphere) and (ray,group) tests EXECUTED + 5 FLOP x rays x planes; for the "
                                 "flat sweep that equals SURVEY 8d's 15 x rays x spheres; brute_force_equivalent = that "
                                 "count over this kernel's time; HBM traffic is the framebuffer only (4 B/px)"},
        }
        # committed rocprofv3 summaries by workload: (file, kernel name in the trace, translation unit, mangled template arguments)
        profiled = {
            ("grid1k", 3840, 2160, 5, 4, "exact", True): ("headline_pmc.csv", "trace_sweep_kernel<false, false, false, 6, false, 1>", "rt_sweep_grouped", "ILb0ELb0ELb0ELi6ELb0ELi1E"),
            ("grid1k", 3840, 2160, 5, 4, "exact", False): ("headline_pmc.csv", "trace_sweep_kernel<false, true, false, 6, false, 0>", "rt_sweep_compact", "ILb0ELb1ELb0ELi6ELb0ELi0E"),
            ("grid16k", 3840, 2160, 5, 1, "exact", True): ("grid16k_pmc.csv", "trace_sweep_kernel<false, false, false, 6, false, 2>", "rt_sweep_grouped3", "ILb0ELb0ELb0ELi6ELb0ELi2E"),
        }.get((args.scene, W, H, args.depth, args.spp, args.mode, bool(grouped)))
        if sorted_levels:
            profiled = None      # (the committed PMC summaries are of the one-launch kernels)
        if world == 1 and profiled:
            # not measurable from inside this process (PMC counters need rocprofv3 around it): taken from the
            # committed rocprofv3 summary of this same command, and only while the kernel is the one profiled
            # (same private-segment size as the build that ships: raytracer_amd/kernel_resources.json)
            prof = os.path.join(ROOT, "profiles", "r3", profiled[0])
            tag = profiled[1]
            traffic, prof_scratch, valu_busy = profiled_traffic(prof, tag)
            built = built_scratch_bytes(profiled[2], profiled[3])
            same = traffic is not None and built is not None and prof_scratch == built
            out["roofline"]["traffic"] = traffic if same else None
            # what the VALU pipes did, as opposed to what the tests were worth: the fraction of cycles a SIMD's VALU was
            # issuing, from the same profile (the algorithmic `frac` falls as the sweep skips more tests; this does not)
            out["roofline"]["valu_busy_profiled"] = valu_busy if same else None
            out["roofline"]["traffic_source"] = (os.path.relpath(prof, ROOT) + ": FETCH_SIZE + WRITE_SIZE per launch (bytes at the L2's memory "
                                                 "side, uncorrected: mostly 4-byte private-memory accesses, a width the guide calls uncalibrated); "
                                                 "profiled kernel scratch %s B, built kernel %s B%s" % (prof_scratch, built, "" if same else
                                                 " -- they differ or one is unknown, so the profiled figure is withheld"))
            # What the VALUs EXECUTED, as opposed to what the tests were worth: SQ_INSTS_VALU_FLOPS_FP32 of the committed
            # instruction-mix pass of this same command (tools/pmc_instmix.sh; the counter counts 1 / 2 / 4 per wave-instruction
            # for add or mul / fma or packed add or mul / packed fma -- calibrated in the same file -- so x 64 lanes = FLOP,
            # idle lanes included), under the same same-build guard as `traffic`.
            mix, mix_scratch = profiled_counters(os.path.join(ROOT, "profiles", "r3", profiled[0].replace("_pmc.csv", "_instmix.csv")), tag)
            if mix.get("SQ_INSTS_VALU_FLOPS_FP32") and built is not None and mix_scratch == built and k0_ms > 0:
                ex_flop = 64.0 * mix["SQ_INSTS_VALU_FLOPS_FP32"]
                out["roofline"]["executed"] = {
                    "flop_per_launch": ex_flop, "tflops": ex_flop / (k0_ms * 1e-3) / 1e12,
                    "frac": ex_flop / (k0_ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS,
                    "valu_instructions_per_launch": mix.get("SQ_INSTS_VALU"),
                    "fp32_instructions_per_launch": sum(mix.get("SQ_INSTS_VALU_" + k, 0.0) for k in ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32")),
                    "source": "profiles/r3/" + profiled[0].replace("_pmc.csv", "_instmix.csv") + ": 64 x SQ_INSTS_VALU_FLOPS_FP32 per launch over this "
                              "run's kernel time; FP32 lane-operations issued whether or not the lane was active"}
            else:
                out["roofline"]["executed"] = None
        if world == 1 and grouped:
            # the same frame through the flat sweep (north_star's every-ray-tests-every-sphere kernel)
            pf = capi.render_params(W, H, args.depth, args.spp, offsets=offsets, mode=mode, sweep=capi.RT_SWEEP_FLAT, **part)
            b = ctx.stats()
            for _ in range(3):
                ctx.render(pf)
            a = ctx.stats()
            f_ms = (a["kernel_ms_total"] - b["kernel_ms_total"]) / 3
            f_ach = (brute_flop / (f_ms * 1e-3)) / 1e12
            out["roofline_flat"] = {"bound": "valu_fp32", "achieved": f_ach, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                                    "frac": f_ach / PEAK_FP32_VALU_TFLOPS, "kernel_ms": f_ms, "mrays_per_s": my_rays / f_ms / 1e3,
                                    "kernel": "trace_sweep_kernel (flat sweep)", "algorithmic_flop_per_launch": int(brute_flop),
                                    "note": "same frame, RT_SWEEP_FLAT: 15 FLOP x rays x spheres + 5 FLOP x rays x planes"}
        out["config"]["frame_crc"] = int(frame.to(torch.int64).sum().item() & 0xFFFFFFFF)
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(capi, W, H, args.depth, args.spp, kind, offsets)
        print(json.dumps(out), flush=True)

    for c, _, _, _ in lanes:
        c.set_stream(0)
        c.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
