"""Frame pipeline for camera sweeps (BASELINE config 5).

The reference renders one frame per process run (raytracer_cpp/main.cpp:230-258).  A sweep keeps the
object list on the device, moves only the camera (rt_set_camera = a new `Camera`, camera.h:18-45,
per frame) and overlaps the three stages of consecutive frames:

    render f+1 (trace kernel, render stream)  ||  download f (D2H into pinned memory, copy stream)
                                              ||  encode f-1 (BMP bytes, host thread)

with a ring of framebuffers so that no stage waits for the one behind it.  Frames are independent,
so N ranks take frames f = rank, rank + N, ... (frame-parallel, no communication); inside a frame the
band split of raytracer_amd.multigpu applies unchanged.
"""
import os
import queue
import threading
import time

import numpy as np

from . import capi


def render_sequence(ctx, cameras, width, height, max_depth=5, spp=1, offsets=None, ring=3,
                    on_frame=None, frame_ids=None, bmp_body=False):
    """Renders one frame per camera with the scene already uploaded on `ctx`.

    on_frame(frame_id, data) runs on a host thread while later frames render; `data` is a view into
    pinned memory that is recycled when the callback returns: the rgba8 frame (HxWx4 uint8), or with
    bmp_body=True the BMP pixel array packed on the device (rt_pack_framebuffer, 1-D uint8), i.e.
    the file minus its 54-byte header.  Returns a dict of timings."""
    import torch

    n = len(cameras)
    frame_ids = list(range(n)) if frame_ids is None else list(frame_ids)
    dev = torch.device("cuda", torch.cuda.current_device())
    render_stream, copy_stream = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    dev_bufs = [torch.empty((height, width), dtype=torch.int32, device=dev) for _ in range(ring)]
    body_words = capi.bmp_body_bytes(width, height) // 4
    out_shape = (body_words,) if bmp_body else (height, width)
    packed = [torch.empty(out_shape, dtype=torch.int32, device=dev) for _ in range(ring)] if bmp_body else dev_bufs
    host_bufs = [torch.empty(out_shape, dtype=torch.int32).pin_memory() for _ in range(ring)]
    rendered = [torch.cuda.Event() for _ in range(ring)]
    copied = [torch.cuda.Event() for _ in range(ring)]
    host_free = [threading.Event() for _ in range(ring)]
    for e in host_free:
        e.set()
    params = capi.render_params(width, height, max_depth, spp, offsets=offsets)

    work = queue.Queue()
    errors = []

    def consumer():
        while True:
            item = work.get()
            if item is None:
                return
            slot, fid = item
            try:
                copied[slot].synchronize()
                if on_frame is not None:
                    flat = host_bufs[slot].numpy().view(np.uint8)
                    on_frame(fid, flat.reshape(-1) if bmp_body else flat.reshape(height, width, 4))
            except Exception as exc:  # surfaced to the caller after the loop
                errors.append(exc)
            finally:
                host_free[slot].set()

    worker = threading.Thread(target=consumer, daemon=True)
    worker.start()
    ctx.set_stream(render_stream.cuda_stream)
    stats0 = ctx.stats() if ctx_has_frame(ctx) else None
    t0 = time.perf_counter()
    try:
        for i, cam in enumerate(cameras):
            slot = i % ring
            host_free[slot].wait()                      # the encoder is done with this slot's host buffer
            host_free[slot].clear()
            render_stream.wait_event(copied[slot])      # ... and its device buffer has been downloaded
            ctx.bind_framebuffer(capi.RT_OUT_RGBA8, dev_bufs[slot].data_ptr(), dev_bufs[slot].numel() * 4)
            ctx.set_camera(cam)
            with torch.cuda.stream(render_stream):
                ctx.render(params, wait=False)
                if bmp_body:
                    ctx.pack_framebuffer(capi.RT_FMT_BMP_BODY, packed[slot].data_ptr(), packed[slot].numel() * 4)
                rendered[slot].record(render_stream)
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(rendered[slot])
                host_bufs[slot].copy_(packed[slot], non_blocking=True)
                copied[slot].record(copy_stream)
            work.put((slot, frame_ids[i]))
        work.put(None)
        worker.join()
        torch.cuda.synchronize()
    finally:
        ctx.set_stream(0)
        ctx.bind_framebuffer(capi.RT_OUT_RGBA8, 0, 0)
    elapsed = time.perf_counter() - t0
    if errors:
        raise errors[0]
    st = ctx.stats()
    k0 = stats0["kernel_ms_total"] if stats0 else 0.0
    return {"frames": n, "seconds": elapsed, "fps": n / elapsed if elapsed > 0 else 0.0,
            "kernel_ms_per_frame": (st["kernel_ms_total"] - k0) / max(1, n)}


def ctx_has_frame(ctx):
    try:
        ctx.stats()
        return True
    except capi.RtError:
        return False


def bmp_writer(out_dir, width, height, pattern="frame_%04d.bmp"):
    """on_frame callback for bmp_body=True: header + the device-packed body = the file WriteBitmap
    would have written (writebitmap.h:88-182)."""
    os.makedirs(out_dir, exist_ok=True)
    head = capi.bmp_header(width, height)

    def write(fid, body):
        with open(os.path.join(out_dir, pattern % fid), "wb") as f:
            f.write(head)
            f.write(memoryview(body))

    return write
