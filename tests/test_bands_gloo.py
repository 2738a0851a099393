"""N > 1 path on CPU: world_size 2 and 3 over gloo.

Every rank plans the bands from the same cost map, renders ITS band (here with the CPU oracle
standing in for the device -- tests may do that), and the padded-band gather of
raytracer_amd.multigpu assembles the frame on rank 0, which must equal the single-process frame.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, w, h, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from raytracer_amd import capi, multigpu
    from oracle import oracle_lib

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        objs, cam = capi.make_scene(capi.RT_SCENE_DEFAULT, w, h)
        # cost map = the oracle's per-pixel ray counts at 1/4 resolution (what the GPU pre-pass returns)
        lw, lh = -(-w // 4), -(-h // 4)
        low = oracle_lib.render(objs, multigpu.scaled_camera(cam, lw, lh), capi.render_params(lw, lh, 5, 1),
                                threads=2, want_rgb8=False)[1][..., 3]
        cost = multigpu.row_cost_from_prepass(low, h)
        bounds = multigpu.plan_bands(cost, h, world)
        y0, y1 = bounds[rank], bounds[rank + 1]
        max_rows = max(bounds[i + 1] - bounds[i] for i in range(world))
        p = capi.render_params(w, h, 5, 1, rows=(y0, y1))
        rgb = oracle_lib.render(objs, cam, p, threads=2, want_f32=False)[0]      # full-image indexed
        band = torch.zeros((max_rows, w, 3), dtype=torch.uint8)
        band[: y1 - y0] = torch.from_numpy(rgb[y0:y1])
        frame = multigpu.gather_bands(band, bounds, rank, world)
        if rank == 0:
            full = oracle_lib.render(objs, cam, capi.render_params(w, h, 5, 1), threads=2, want_f32=False)[0]
            q.put(("ok", bool(np.array_equal(frame.numpy(), full)), bounds))
        else:
            assert frame is None
    except Exception as e:  # surface the failure to the parent
        q.put(("err", repr(e), None))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_band_gather_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 96, 72, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    for p in procs:
        if p.is_alive():
            p.terminate()
        assert p.exitcode == 0
    status, equal, bounds = q.get(timeout=5)
    assert status == "ok" and equal
    assert bounds[0] == 0 and bounds[-1] == 72 and len(bounds) == world + 1
    heights = [bounds[i + 1] - bounds[i] for i in range(world)]
    assert min(heights) > 0 and len(set(heights)) > 1      # cost-balanced, hence unequal, bands


def _strip_worker(rank, world, port, w, h, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from raytracer_amd import capi, multigpu
    from oracle import oracle_lib

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        objs, cam = capi.make_scene(capi.RT_SCENE_DEFAULT, w, h)
        # this rank's strips, side by side the way render_params(interleave=(world, rank)) lays them out
        # (the device-side mapping itself is checked on the GPU: test_col_interleave_tiles_the_frame)
        cols = multigpu.interleaved_col_index(w, world, rank)
        assert len(cols) == capi.interleaved_count(w, world, rank)
        full = oracle_lib.render(objs, cam, capi.render_params(w, h, 5, 1), threads=2, want_f32=False)[0]
        part = torch.zeros((h, capi.interleaved_count(w, world, 0), 3), dtype=torch.uint8)
        part[:, : len(cols)] = torch.from_numpy(np.ascontiguousarray(full[:, cols]))
        cache = {}
        frame = multigpu.gather_strips(part, w, rank, world, index_cache=cache)
        recv = {k: v.data_ptr() for k, v in cache.items() if k[0] == "recv"}
        out = torch.zeros((h, w, 3), dtype=torch.uint8) if rank == 0 else None
        again = multigpu.gather_strips(part, w, rank, world, out=out, index_cache=cache)   # cached receive buffer and indices
        if rank == 0:
            # the second call reuses the first call's receive buffer and writes into `out`: nothing new is allocated
            assert len(recv) == 1 and recv == {k: v.data_ptr() for k, v in cache.items() if k[0] == "recv"}
            assert again is out
            q.put(("ok", bool(np.array_equal(frame.numpy(), full) and np.array_equal(again.numpy(), full)), None))
        else:
            assert frame is None and again is None
    except Exception as e:
        q.put(("err", repr(e), None))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,w", [(2, 96), (3, 91), (3, 120)])
def test_strip_gather_gloo(world, w):
    """8-column strips dealt round-robin (rt_render_params.col_interleave): the padded gather and the
    column scatter on rank 0 rebuild the single-process frame, also when the last strip is narrow and
    the shares are unequal (equal shares take the one-copy path, unequal ones the per-rank column scatter)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_strip_worker, args=(r, world, port, w, 40, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    for p in procs:
        if p.is_alive():
            p.terminate()
        assert p.exitcode == 0
    status, equal, _ = q.get(timeout=5)
    assert status == "ok" and equal
