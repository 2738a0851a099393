"""Row-band partitioning of one frame over the ranks of a node (one process per GPU).

The reference's only parallel axis is image rows over std::threads (raytracer_cpp/main.cpp:193-196);
two plans: (a) 8-column strips dealt round-robin to the GPUs (render_params(interleave=...), gather_strips): no cost
estimate needed, every GPU gets the same share of every row, cheap or expensive; (b) rows cut into contiguous bands, one per GPU, balanced by a cost estimate from a
1/16-resolution pre-pass (the F32X4 plane's w channel with RT_OUT_COST = the tests each pixel's rays
took, exact tests weighted: rays near the horizon cost an order of magnitude more than the average).  Every rank runs the
same deterministic pre-pass, so all ranks agree on the bounds without communication.  The only
exchange step is the gather of the finished bands to rank 0 (RCCL when the tensors are on GPUs, gloo
on CPU in the tests).
"""
import ctypes as C

import numpy as np

from . import capi


def scaled_camera(cam, width, height):
    """Same camera pose and lens, different view-plane size (camera.h:24-25,41)."""
    out = capi.Camera()
    C.memmove(C.byref(out), C.byref(cam), C.sizeof(capi.Camera))
    out.image_width = float(width)
    out.image_height = float(height)
    out.aspect_ratio = np.float32(width) / np.float32(height)
    return out


def row_cost_from_prepass(cost_lowres, height):
    """Per-row cost of the full image from the low-resolution ray-count map (rows x cols)."""
    low = np.asarray(cost_lowres, dtype=np.float64).sum(axis=1)
    idx = np.minimum((np.arange(height) * low.shape[0]) // height, low.shape[0] - 1)
    return low[idx].astype(np.float32)


def prepass_row_cost(ctx, objs, cam, width, height, max_depth, factor=4):
    """Renders the scene at 1/factor resolution per axis and returns a per-row cost for `height` rows.

    Leaves the full-size camera uploaded again on return."""
    lw, lh = max(1, -(-width // factor)), max(1, -(-height // factor))
    ctx.set_camera(scaled_camera(cam, lw, lh))       # the camera travels with each launch: nothing is uploaded
    try:
        ctx.render(capi.render_params(lw, lh, max_depth, 1, outputs=capi.RT_OUT_F32X4 | capi.RT_OUT_COST, stats=True))
        f32 = ctx.download(capi.RT_FMT_F32X4, lw, lh)
    finally:
        ctx.set_camera(cam)
    return row_cost_from_prepass(f32[..., 3], height)


def hot_row(row_cost, lead=16):
    """Where to start launching (rt_render_params.first_row): a little above the most expensive row of the
    cost map, so that the frame's slowest waves run under the rest of the frame instead of after it."""
    c = np.asarray(row_cost, dtype=np.float64)
    return int(max(0, int(np.argmax(c)) - lead)) if c.size else 0


def plan_bands(row_cost, height, world):
    """Contiguous bands of near-equal cost: list of world+1 row boundaries (rt_split_bands)."""
    return capi.split_bands(height, world, row_cost)


def interleaved_col_index(width, world, rank):
    """Image columns of rank's strips, in the order its render lays them side by side
    (rt_interleaved_index): strips of capi.RT_STRIP_COLS columns dealt round-robin, strip s to rank s % world."""
    s = capi.RT_STRIP_COLS
    strips = np.arange(rank, -(-width // s), world)
    cols = (strips[:, None] * s + np.arange(s)[None, :]).reshape(-1)
    return cols[cols < width]


def strip_receive_buffer(cache, part, world):
    """Rank 0's receive buffer of gather_strips ([world, rows, max_cols, ...]), made on first use and kept in `cache`:
    call it at set-up so that no frame of a timed loop allocates."""
    import torch
    key = ("recv", world, tuple(part.shape), part.dtype, str(part.device))
    if key not in cache:
        cache[key] = torch.empty((world,) + tuple(part.shape), dtype=part.dtype, device=part.device)
    return cache[key]


def gather_strips(part, width, rank, world, out=None, group=None, index_cache=None):
    """Gathers interleaved renders (render_params(interleave=(world, rank))) to rank 0 and puts every strip
    in its place.  part: torch tensor [rows, max_cols, ...] whose first interleaved_count(width, world, rank)
    columns are this rank's output; all ranks pass the same max_cols = interleaved_count(width, world, 0)
    (a render's own planes are exactly its count wide: when the shares differ, i.e. width is not a multiple of
    8 x world, copy them into the padded tensor first; with equal shares the render can write into it directly).
    Returns the assembled [rows, width, ...] tensor on rank 0 (written into `out` when given), else None.

    index_cache: a dict the caller keeps between frames.  It holds rank 0's receive buffer ([world, rows, max_cols,
    ...], allocated on the first call and reused: with `out` and `index_cache` given, later calls allocate nothing
    on the device) and, for unequal shares, the column indices.  With equal shares (width a multiple of 8 x world)
    the strips go to their places in ONE strided copy: the frame seen as [rows, width / (8 world), world, 8] is
    the receive buffer [world, rows, width / (8 world), 8] with its axes permuted."""
    import torch
    import torch.distributed as dist

    cache = index_cache if index_cache is not None else {}
    s = capi.RT_STRIP_COLS
    stack = None
    if world > 1:
        parts = None
        if rank == 0:
            stack = strip_receive_buffer(cache, part, world)
            parts = list(stack.unbind(0))                 # views of the one buffer: no device allocation
        dist.gather(part, parts, dst=0, group=group)
        if rank != 0:
            return None
    else:
        stack = part.unsqueeze(0)
    return assemble_strips(stack, width, out=out, index_cache=cache)


def assemble_strips(stack, width, out=None, index_cache=None):
    """stack[r] = rank r's strips side by side ([world, rows, max_cols, ...]) -> the frame [rows, width, ...].
    Equal shares (width a multiple of 8 x world): one strided copy, no index tensors, nothing allocated when `out`
    is given.  Otherwise one column scatter per rank (indices cached in `index_cache`)."""
    import torch

    cache = index_cache if index_cache is not None else {}
    s = capi.RT_STRIP_COLS
    world, rows, cols = stack.shape[0], stack.shape[1], stack.shape[2]
    tail = tuple(stack.shape[3:])
    if out is None:
        out = torch.empty((rows, width) + tail, dtype=stack.dtype, device=stack.device)
    if width % (s * world) == 0 and cols * world == width and out.is_contiguous() and stack.is_contiguous():
        t = len(tail)
        src = stack.view((world, rows, cols // s, s) + tail).permute((1, 2, 0, 3) + tuple(range(4, 4 + t)))
        out.view((rows, cols // s, world, s) + tail).copy_(src)
        return out
    for r in range(world):
        key = (width, world, r, str(stack.device))
        if key not in cache:
            cache[key] = torch.from_numpy(interleaved_col_index(width, world, r).astype(np.int64)).to(stack.device)
        idx = cache[key]
        out.index_copy_(1, idx, stack[r][:, : idx.numel()])
    return out


def gather_bands(band, bounds, rank, world, out=None, group=None):
    """Gathers the bands to rank 0, each straight into its rows of the frame.

    band: torch tensor [>= rows, ...] whose first bounds[rank+1]-bounds[rank] rows are this rank's.  Every
    rank sends exactly its rows (RCCL send/recv: each peer has its own xGMI link to rank 0), and rank 0
    receives them in place in `out[y0:y1]` -- no padding, no staging list, no allocation when `out` is given.
    Returns the assembled [height, ...] tensor on rank 0 (`out` when given), else None."""
    import torch
    import torch.distributed as dist

    height = bounds[-1]
    y0, y1 = bounds[rank], bounds[rank + 1]
    if world == 1:
        if out is None:
            return band[: y1 - y0]
        out[: y1 - y0].copy_(band[: y1 - y0])
        return out
    if rank != 0:
        if y1 > y0:
            dist.send(band[: y1 - y0], dst=0, group=group)
        return None
    if out is None:
        out = torch.empty((height,) + tuple(band.shape[1:]), dtype=band.dtype, device=band.device)
    ops = [dist.P2POp(dist.irecv, out[bounds[r]:bounds[r + 1]], r, group) for r in range(1, world) if bounds[r + 1] > bounds[r]]
    out[y0:y1].copy_(band[: y1 - y0])
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out
