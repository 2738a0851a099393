"""ctypes wrapper of oracle/liboracle.so (and oracle/_ref/libref.so when present).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_PATH = os.path.join(_HERE, "liboracle.so")
REF_PATH = os.path.join(_HERE, "_ref", "libref.so")


class OrcStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("primary_rays", "reflection_rays", "shadow_rays_pruned", "shadow_rays_verbatim",
                 "calls", "hits", "lit", "nearest_tests", "shadow_tests", "sphere_tests",
                 "sphere_accepts", "plane_tests", "plane_accepts")] + [("calls_by_depth", C.c_uint64 * 16)]


def load_oracle():
    if not os.path.exists(ORACLE_PATH):
        raise ImportError(f"{ORACLE_PATH} missing: run `make -C oracle`")
    return C.CDLL(ORACLE_PATH)


def load_ref():
    """The reference's own headers compiled in the build container; None when absent."""
    if not os.path.exists(REF_PATH):
        return None
    return C.CDLL(REF_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = load_oracle()
        _lib.orc_render.restype = None
        _lib.orc_lens_ray.restype = None
        _lib.orc_trace.restype = None
        _lib.orc_trace_samples.restype = None
        _lib.orc_trace_samples_v.restype = None
    return _lib


def render(objs, cam, params, verbatim=False, threads=1, want_rgb8=True, want_f32=True, want_stats=False, f32_in=None):
    """orc_render over full-image buffers; returns (rgb8 HxWx3, f32 HxWx4, stats dict|None)."""
    h, w = params.height, params.width
    rgb8 = np.zeros((h, w, 3), dtype=np.uint8) if want_rgb8 else None
    f32 = np.zeros((h, w, 4), dtype=np.float32) if want_f32 else None
    if f32_in is not None:      # progressive accumulation continues from an earlier frame's float plane
        f32 = np.array(f32_in, dtype=np.float32, copy=True).reshape(h, w, 4)
    st = OrcStats() if want_stats else None
    lib().orc_render(objs, C.c_int(len(objs)), C.byref(cam), C.byref(params), C.c_int(int(verbatim)),
                     C.c_int(threads),
                     rgb8.ctypes.data_as(C.c_void_p) if want_rgb8 else None,
                     f32.ctypes.data_as(C.c_void_p) if want_f32 else None,
                     C.byref(st) if want_stats else None)
    sd = None
    if want_stats:
        sd = {n: getattr(st, n) for n, _ in OrcStats._fields_[:-1]}
        sd["calls_by_depth"] = list(st.calls_by_depth)
    return rgb8, f32, sd


def trace_samples(objs, cam, max_depth, sxy, verbatim=False):
    sxy = np.ascontiguousarray(sxy, dtype=np.float32).reshape(-1, 2)
    out = np.zeros((sxy.shape[0], 3), dtype=np.float32)
    lib().orc_trace_samples_v(objs, C.c_int(len(objs)), C.byref(cam), C.c_int(max_depth), C.c_int(int(verbatim)),
                              sxy.ctypes.data_as(C.c_void_p), C.c_int(sxy.shape[0]),
                              out.ctypes.data_as(C.c_void_p))
    return out


def lens_ray(cam, focal_length, lens, sxy):
    """orc_lens_ray: (origin, direction) of the thin-lens ray."""
    o, d = (C.c_float * 3)(), (C.c_float * 3)()
    lib().orc_lens_ray(C.byref(cam), C.c_float(focal_length), C.c_float(lens[0]), C.c_float(lens[1]),
                       C.c_float(sxy[0]), C.c_float(sxy[1]), o, d)
    return list(o), list(d)


def write_bmp(path, rgb8):
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w, _ = rgb8.shape
    return lib().orc_write_bmp(os.fsencode(path), C.c_int(w), C.c_int(h), rgb8.ctypes.data_as(C.c_void_p))


# ---- the reference's own trace loop (oracle/_ref/libref_trace_d<MAX_DEPTH>.so) ---------------------
# Present where oracle/Makefile ran with /root/reference at hand (the build container; the built
# libraries travel to the GPU box with the tree, the reference itself never does).

def ref_trace_path(max_depth):
    return os.path.join(_HERE, "_ref", "libref_trace_d%d.so" % max_depth)


class RefTrace:
    """raytracer_cpp/main.cpp's InitScene / FindNearestObject / TraceRay / DrawScene, compiled from the
    reference's file with MAX_DEPTH = max_depth (oracle/ref_trace_harness.cpp).  One scene at a time per
    library: the reference keeps it in globals."""

    def __init__(self, max_depth):
        path = ref_trace_path(max_depth)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        assert self.lib.ref_trace_max_depth() == max_depth
        self.max_depth = max_depth
        self.lib.ref_trace_samples.restype = None
        self.lib.ref_trace_frame.restype = None
        self.lib.ref_trace_ray.restype = None

    def init_scene(self):
        self.lib.ref_trace_init_scene()

    def set_scene(self, objs, cam, counting=False):
        rc = self.lib.ref_trace_set_scene(objs, C.c_int(len(objs)), C.byref(cam), C.c_int(int(counting)))
        assert rc == 0

    def get_scene(self, object_array_type, camera_type, cap=64):
        objs = (object_array_type * cap)()
        cam = camera_type()
        n = self.lib.ref_trace_get_scene(objs, C.c_int(cap), C.byref(cam))
        return objs, n, cam

    def trace_ray(self, o, d, depth=0):
        F3 = C.c_float * 3
        out = F3()
        self.lib.ref_trace_ray(F3(*o), F3(*d), C.c_int(depth), out)
        return np.array(list(out), dtype=np.float32)

    def find_nearest(self, o, d):
        F3 = C.c_float * 3
        dist = C.c_float(0.0)
        idx = self.lib.ref_trace_find_nearest(F3(*o), F3(*d), C.byref(dist))
        return idx, np.float32(dist.value)

    def trace_samples(self, sxy, threads=8):
        sxy = np.ascontiguousarray(sxy, dtype=np.float32).reshape(-1, 2)
        out = np.zeros((sxy.shape[0], 3), dtype=np.float32)
        self.lib.ref_trace_samples(sxy.ctypes.data_as(C.c_void_p), C.c_int(sxy.shape[0]),
                                   out.ctypes.data_as(C.c_void_p), C.c_int(threads))
        return out

    def frame(self, w, h, spp, offsets, rows=None, threads=8):
        off = np.ascontiguousarray(offsets, dtype=np.float32).reshape(-1)
        assert off.size == 2 * spp
        y0, y1 = rows if rows is not None else (0, h)
        rgb8 = np.zeros((h, w, 3), dtype=np.uint8)
        f32 = np.zeros((h, w, 4), dtype=np.float32)
        self.lib.ref_trace_frame(C.c_int(w), C.c_int(h), C.c_int(spp), off.ctypes.data_as(C.c_void_p), C.c_int(y0), C.c_int(y1),
                                 rgb8.ctypes.data_as(C.c_void_p), f32.ctypes.data_as(C.c_void_p), C.c_int(threads))
        return rgb8, f32

    def draw_scene(self, bmp_path, partitions, antialias):
        w, h = self.lib.ref_trace_image_width(), self.lib.ref_trace_image_height()
        rgb8 = np.zeros((h, w, 3), dtype=np.uint8)
        self.lib.ref_trace_draw_scene(os.fsencode(bmp_path) if bmp_path else None, C.c_int(partitions), C.c_int(antialias),
                                      rgb8.ctypes.data_as(C.c_void_p))
        return rgb8

    def reset_counters(self):
        self.lib.ref_trace_reset_counters()

    def counters(self):
        out = (C.c_ulonglong * 4)()
        self.lib.ref_trace_get_counters(out)
        return {"intersects": out[0], "normals": out[1], "ray_from": out[2], "materials": out[3]}


def load_ref_trace(max_depth):
    """RefTrace for that MAX_DEPTH, or None when the library was not built (no /root/reference)."""
    try:
        return RefTrace(max_depth)
    except FileNotFoundError:
        return None
