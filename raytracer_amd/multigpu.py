"""Row-band partitioning of one frame over the ranks of a node (one process per GPU).

The reference's only parallel axis is image rows over std::threads (raytracer_cpp/main.cpp:193-196);
here rows are cut into contiguous bands, one per GPU, balanced by a cost estimate from a
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
    ctx.upload_scene(objs, scaled_camera(cam, lw, lh))
    ctx.render(capi.render_params(lw, lh, max_depth, 1, outputs=capi.RT_OUT_F32X4 | capi.RT_OUT_COST, stats=True))
    f32 = ctx.download(capi.RT_FMT_F32X4, lw, lh)
    ctx.upload_scene(objs, cam)
    return row_cost_from_prepass(f32[..., 3], height)


def plan_bands(row_cost, height, world):
    """Contiguous bands of near-equal cost: list of world+1 row boundaries (rt_split_bands)."""
    return capi.split_bands(height, world, row_cost)


def gather_bands(band, bounds, rank, world, out=None, group=None):
    """Gathers the bands to rank 0.

    band: torch tensor [max_rows, ...] whose first (bounds[rank+1]-bounds[rank]) rows are this rank's
    rows; all ranks pass the same max_rows (padded gather, since RCCL's gather wants equal counts).
    Returns the assembled [height, ...] tensor on rank 0 (written into `out` when given), else None."""
    import torch
    import torch.distributed as dist

    height = bounds[-1]
    if world == 1:
        rows = bounds[1] - bounds[0]
        if out is None:
            return band[:rows]
        out[:rows].copy_(band[:rows])
        return out
    parts = None
    if rank == 0:
        parts = [torch.empty_like(band) for _ in range(world)]
    dist.gather(band, parts, dst=0, group=group)
    if rank != 0:
        return None
    if out is None:
        out = torch.empty((height,) + tuple(band.shape[1:]), dtype=band.dtype, device=band.device)
    for r in range(world):
        y0, y1 = bounds[r], bounds[r + 1]
        out[y0:y1].copy_(parts[r][: y1 - y0])
    return out
