"""ctypes binding of the C ABI in include/rt_api.h (raytracer_amd/librt_amd.so).

This is plumbing for the tests and bench.py: every call goes through the same exported symbols a
C/C++ host would use.  There is no fallback: if the shared library is missing the import fails,
and if no HIP device is usable ``Context()`` raises ``RtError``.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "librt_amd.so")  # override: kernel A/B builds

RT_OK = 0
RT_ERR_INVALID_ARG, RT_ERR_NO_DEVICE, RT_ERR_HIP, RT_ERR_BUFFER_TOO_SMALL = -1, -2, -3, -4
RT_ERR_NO_SCENE, RT_ERR_NO_FRAME, RT_ERR_LIMIT = -5, -6, -7
RT_OBJ_SPHERE, RT_OBJ_TILED_PLANE = 0, 1
RT_OUT_RGBA8, RT_OUT_F32X4, RT_OUT_COST = 1, 2, 4
RT_FMT_RGB8, RT_FMT_RGBA8, RT_FMT_F32X4, RT_FMT_BMP_BODY = 0, 1, 2, 3
RT_MODE_EXACT, RT_MODE_FAST = 0, 1
RT_SWEEP_AUTO, RT_SWEEP_FLAT, RT_SWEEP_GROUPED, RT_SWEEP_SORTED = 0, 1, 2, 3
RT_PLAN_AUTO, RT_PLAN_BANDS, RT_PLAN_STRIPS = 0, 1, 2
RT_STRIP_COLS = 8
RT_MAX_DEVICES = 8
RT_SCENE_DEFAULT, RT_SCENE_GRID1K, RT_SCENE_GRID16K = 0, 1, 2


class Material(C.Structure):
    """rt_material == struct Material (raytracer_cpp/sceneobjects.h:3-9)."""
    _fields_ = [("albedo", C.c_float * 3), ("specular", C.c_float * 3),
                ("reflectance", C.c_float), ("emissive", C.c_float * 3)]


class Object(C.Structure):
    """rt_object: POD flattening of Sphere / TiledPlane (sceneobjects.h:36-131)."""
    _fields_ = [("type", C.c_int32), ("a", C.c_float * 3), ("b", C.c_float * 3),
                ("radius", C.c_float), ("mat", Material), ("mat2", Material)]


class Camera(C.Structure):
    """rt_camera == the fields of struct Camera (raytracer_cpp/camera.h:6-16)."""
    _fields_ = [("position", C.c_float * 3), ("view_direction", C.c_float * 3),
                ("right", C.c_float * 3), ("up", C.c_float * 3),
                ("aspect_ratio", C.c_float), ("half_angle", C.c_float),
                ("image_width", C.c_float), ("image_height", C.c_float)]


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("max_depth", C.c_int32),
                ("spp", C.c_int32), ("sample_offsets", C.POINTER(C.c_float)),
                ("row_begin", C.c_int32), ("row_end", C.c_int32), ("outputs", C.c_int32),
                ("mode", C.c_int32), ("collect_stats", C.c_int32), ("sweep", C.c_int32),
                ("lens_offsets", C.POINTER(C.c_float)), ("focal_length", C.c_float),
                ("accumulate", C.c_int32), ("accum_index", C.c_int32),
                ("col_interleave", C.c_int32), ("col_phase", C.c_int32), ("first_row", C.c_int32),
                ("device_plan", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("primary_rays", "reflection_rays", "shadow_rays", "hits", "lit",
                 "sphere_tests", "plane_tests", "n_objects", "n_emitters")] + \
               [("kernel_ms", C.c_double), ("kernel_ms_total", C.c_double), ("kernels_timed", C.c_uint64),
                ("sphere_tests_run", C.c_uint64), ("group_tests_run", C.c_uint64), ("exact_tests_run", C.c_uint64),
                ("n_devices", C.c_uint32), ("device_id", C.c_int32 * RT_MAX_DEVICES), ("device_row_begin", C.c_int32 * RT_MAX_DEVICES),
                ("device_row_end", C.c_int32 * RT_MAX_DEVICES), ("device_kernel_ms", C.c_double * RT_MAX_DEVICES),
                ("device_gather_ms", C.c_double * RT_MAX_DEVICES), ("gather_ms", C.c_double),
                ("replans", C.c_uint32), ("device_plan", C.c_int32), ("sweep", C.c_int32), ("reserved", C.c_int32)]


EXPORTS = [
    "rt_create", "rt_create_multi", "rt_destroy", "rt_plan_groups", "rt_interleaved_count", "rt_interleaved_index", "rt_upload_scene", "rt_set_camera", "rt_render", "rt_download_framebuffer",
    "rt_pack_framebuffer", "rt_framebuffer_device_ptr", "rt_bind_framebuffer", "rt_get_stream", "rt_set_stream", "rt_render_async",
    "rt_sync", "rt_render_noalloc", "rt_get_stats", "rt_last_error", "rt_api_version", "rt_camera_init",
    "rt_make_scene", "rt_split_bands", "rt_write_bmp", "rt_bmp_header",
]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C raytracer_amd/csrc` "
            "(or __graft_entry__.build()). There is no fallback path.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, f32p = C.c_void_p, C.c_int32, C.POINTER(C.c_float)
    lib.rt_create.argtypes = [C.POINTER(vp), C.c_int]
    lib.rt_create_multi.argtypes = [C.POINTER(vp), C.POINTER(C.c_int), C.c_int]
    lib.rt_destroy.argtypes = [vp]
    lib.rt_destroy.restype = None
    lib.rt_upload_scene.argtypes = [vp, C.POINTER(Object), i32, C.POINTER(Camera)]
    lib.rt_set_camera.argtypes = [vp, C.POINTER(Camera)]
    lib.rt_interleaved_count.argtypes = [i32, i32, i32]
    lib.rt_interleaved_count.restype = i32
    lib.rt_interleaved_index.argtypes = [i32, i32, i32]
    lib.rt_interleaved_index.restype = i32
    lib.rt_plan_groups.argtypes = [C.POINTER(Object), i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(C.c_float),
                                   C.POINTER(C.c_float), C.POINTER(i32), C.POINTER(C.c_float)]
    lib.rt_render.argtypes = [vp, C.POINTER(RenderParams)]
    lib.rt_render_async.argtypes = [vp, C.POINTER(RenderParams)]
    lib.rt_sync.argtypes = [vp]
    lib.rt_render_noalloc.argtypes = [vp, C.POINTER(RenderParams), vp, C.c_size_t, vp, C.c_size_t]
    lib.rt_download_framebuffer.argtypes = [vp, vp, C.c_size_t, i32]
    lib.rt_pack_framebuffer.argtypes = [vp, i32, vp, C.c_size_t]
    lib.rt_framebuffer_device_ptr.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.rt_bind_framebuffer.argtypes = [vp, i32, vp, C.c_size_t]
    lib.rt_get_stream.argtypes = [vp, C.POINTER(vp)]
    lib.rt_set_stream.argtypes = [vp, vp]
    lib.rt_get_stats.argtypes = [vp, C.POINTER(Stats)]
    lib.rt_last_error.argtypes = [vp]
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_api_version.argtypes = []
    lib.rt_camera_init.argtypes = [C.POINTER(Camera), f32p, f32p, C.c_float, i32, i32]
    lib.rt_camera_init.restype = None
    lib.rt_make_scene.argtypes = [i32, i32, i32, i32, i32, C.POINTER(Object), i32, C.POINTER(Camera)]
    lib.rt_make_scene.restype = i32
    lib.rt_split_bands.argtypes = [f32p, i32, i32, C.POINTER(i32)]
    lib.rt_write_bmp.argtypes = [C.c_char_p, i32, i32, C.POINTER(C.c_uint8)]
    lib.rt_bmp_header.argtypes = [i32, i32, C.POINTER(C.c_uint8)]
    return lib


lib = _load()


class RtError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"rt_status {code}: {text}")
        self.code = code


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def camera_init(position, direction, fov, width, height):
    """Camera::Camera (camera.h:18-45) through rt_camera_init."""
    cam = Camera()
    lib.rt_camera_init(C.byref(cam), _f3(position), _f3(direction), float(fov), int(width), int(height))
    return cam


def make_scene(kind, width, height, frame=0, n_frames=240):
    """Returns (ctypes array of Object, Camera) from rt_make_scene."""
    cam = Camera()
    n = lib.rt_make_scene(kind, width, height, frame, n_frames, None, 0, None)
    if n < 0:
        raise RtError(n, "rt_make_scene")
    objs = (Object * n)()
    lib.rt_make_scene(kind, width, height, frame, n_frames, objs, n, C.byref(cam))
    return objs, cam


def split_bands(height, n, row_cost=None):
    bounds = (C.c_int32 * (n + 1))()
    cost = None
    if row_cost is not None:
        arr = np.ascontiguousarray(row_cost, dtype=np.float32)
        assert arr.shape == (height,)
        cost = arr.ctypes.data_as(C.POINTER(C.c_float))
    rc = lib.rt_split_bands(cost, height, n, bounds)
    if rc:
        raise RtError(rc, "rt_split_bands")
    return list(bounds)


def write_bmp(path, rgb8):
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w, _ = rgb8.shape
    rc = lib.rt_write_bmp(os.fsencode(path), w, h, rgb8.ctypes.data_as(C.POINTER(C.c_uint8)))
    if rc:
        raise RtError(rc, "rt_write_bmp")


def bmp_header(width, height):
    out = (C.c_uint8 * 54)()
    rc = lib.rt_bmp_header(width, height, out)
    if rc:
        raise RtError(rc, "rt_bmp_header")
    return bytes(out)


def bmp_body_bytes(width, rows):
    return ((width * 3 + 3) & ~3) * rows


def stratified_offsets(spp=16, seed=0x16A11CE):
    """Config 3's sample pattern: sqrt(spp) x sqrt(spp) stratified offsets from the
    scene generators' LCG, one pattern shared by every pixel like the reference's static pattern
    (raytracer_cpp/main.cpp:202)."""
    g = int(round(spp ** 0.5))
    assert g * g == spp
    s, out = seed, []
    for i in range(spp):
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        u = np.float32(s >> 8) * np.float32(2.0 ** -24)
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        v = np.float32(s >> 8) * np.float32(2.0 ** -24)
        out.append(((np.float32(i % g) + u) / np.float32(g), (np.float32(i // g) + v) / np.float32(g)))
    return np.asarray(out, dtype=np.float32)


def render_params(width, height, max_depth=5, spp=1, offsets=None, rows=None, outputs=RT_OUT_RGBA8,
                  mode=RT_MODE_EXACT, stats=False, sweep=0, lens=None, focal_length=0.0, accumulate=None,
                  interleave=None, first_row=0, device_plan=RT_PLAN_AUTO):
    """interleave: (n, phase) = render the 8-column strips s with s % n == phase, side by side (multi-GPU);
    lens: spp x 2 lens-plane offsets (thin-lens depth of field) with focal_length; accumulate: k
    (currentSample) for the progressive running mean, None = overwrite."""
    p = RenderParams()
    p.width, p.height, p.max_depth, p.spp = width, height, max_depth, spp
    keep = None
    if offsets is not None:
        keep = np.ascontiguousarray(offsets, dtype=np.float32).reshape(-1)
        assert keep.size == 2 * spp
        p.sample_offsets = keep.ctypes.data_as(C.POINTER(C.c_float))
    if rows is not None:
        p.row_begin, p.row_end = rows
    p.outputs, p.mode, p.collect_stats, p.sweep = outputs, mode, int(stats), sweep
    keep_lens = None
    if lens is not None:
        keep_lens = np.ascontiguousarray(lens, dtype=np.float32).reshape(-1)
        assert keep_lens.size == 2 * spp
        p.lens_offsets = keep_lens.ctypes.data_as(C.POINTER(C.c_float))
        p.focal_length = float(focal_length)
    if accumulate is not None:
        p.accumulate, p.accum_index = 1, int(accumulate)
    if interleave is not None:
        p.col_interleave, p.col_phase = int(interleave[0]), int(interleave[1])
    p.first_row = int(first_row)
    p.device_plan = int(device_plan)
    p._keep = (keep, keep_lens)  # keep the arrays alive as long as the struct
    return p


class Context:
    """rt_context: one device, one stream."""

    def __init__(self, device=0, devices=None):
        """devices: a list of HIP device ids -> one context over all of them (rt_create_multi)."""
        self._h = C.c_void_p()
        if devices is not None:
            ids = (C.c_int * len(devices))(*devices)
            rc = lib.rt_create_multi(C.byref(self._h), ids, len(devices))
        else:
            rc = lib.rt_create(C.byref(self._h), device)
        if rc:
            raise RtError(rc, lib.rt_last_error(None).decode())

    def _check(self, rc):
        if rc:
            raise RtError(rc, lib.rt_last_error(self._h).decode())

    def close(self):
        if self._h:
            lib.rt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_scene(self, objs, cam):
        self._check(lib.rt_upload_scene(self._h, objs, len(objs), C.byref(cam)))

    def set_camera(self, cam):
        self._check(lib.rt_set_camera(self._h, C.byref(cam)))

    def render(self, params, wait=True):
        fn = lib.rt_render if wait else lib.rt_render_async
        self._check(fn(self._h, C.byref(params)))

    def sync(self):
        self._check(lib.rt_sync(self._h))

    def render_noalloc(self, params, rgba8_ptr, rgba8_bytes, f32x4_ptr=0, f32x4_bytes=0):
        """rt_render_noalloc: enqueue-only render into caller-owned device planes, allocating nothing."""
        self._check(lib.rt_render_noalloc(self._h, C.byref(params), C.c_void_p(rgba8_ptr or None), C.c_size_t(rgba8_bytes),
                                          C.c_void_p(f32x4_ptr or None), C.c_size_t(f32x4_bytes)))

    def stream(self):
        s = C.c_void_p()
        self._check(lib.rt_get_stream(self._h, C.byref(s)))
        return s.value or 0

    def set_stream(self, stream_handle):
        """hipStream_t as an int (e.g. torch.cuda.Stream().cuda_stream); 0/None = the context's own."""
        self._check(lib.rt_set_stream(self._h, C.c_void_p(stream_handle or None)))

    def bind_framebuffer(self, plane, ptr, nbytes):
        self._check(lib.rt_bind_framebuffer(self._h, plane, C.c_void_p(ptr), nbytes))

    def pack_framebuffer(self, fmt, device_ptr, nbytes):
        self._check(lib.rt_pack_framebuffer(self._h, fmt, C.c_void_p(device_ptr), nbytes))

    def device_ptr(self, plane):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(lib.rt_framebuffer_device_ptr(self._h, plane, C.byref(p), C.byref(n)))
        return p.value, n.value

    def download(self, fmt, width, rows):
        if fmt == RT_FMT_F32X4:
            out = np.empty((rows, width, 4), dtype=np.float32)
        elif fmt == RT_FMT_RGBA8:
            out = np.empty((rows, width, 4), dtype=np.uint8)
        elif fmt == RT_FMT_RGB8:
            out = np.empty((rows, width, 3), dtype=np.uint8)
        else:
            stride = width * 3 + (4 - (width * 3) % 4) % 4
            out = np.empty((rows, stride), dtype=np.uint8)
        self._check(lib.rt_download_framebuffer(self._h, out.ctypes.data_as(C.c_void_p), out.nbytes, fmt))
        return out

    def stats(self):
        st = Stats()
        self._check(lib.rt_get_stats(self._h, C.byref(st)))
        return {n: (list(getattr(st, n)) if hasattr(getattr(st, n), "__len__") else getattr(st, n)) for n, _ in Stats._fields_}


# ---- scene construction helpers mirroring the reference's constructors -------------------------

def material(albedo=(0, 0, 0), specular=(0, 0, 0), reflectance=0.0, emissive=(0, 0, 0)):
    """struct Material (sceneobjects.h:3-9)."""
    m = Material()
    m.albedo[:] = [float(v) for v in albedo]
    m.specular[:] = [float(v) for v in specular]
    m.reflectance = float(reflectance)
    m.emissive[:] = [float(v) for v in emissive]
    return m


def sphere(mat, center, radius):
    """Sphere(mat, c, r) (sceneobjects.h:42-47)."""
    o = Object()
    o.type = RT_OBJ_SPHERE
    o.a[:] = [float(v) for v in center]
    o.radius = float(radius)
    o.mat = mat
    return o


def tiled_plane(origin, normal, white=None, black=None):
    """TiledPlane(o, n) (sceneobjects.h:93-104) with its white/black materials (emissive = 0)."""
    o = Object()
    o.type = RT_OBJ_TILED_PLANE
    o.a[:] = [float(v) for v in origin]
    o.b[:] = [float(v) for v in normal]
    o.mat = white if white is not None else material((1, 1, 1), (1, 1, 1), 0.6)
    o.mat2 = black if black is not None else material((0, 0, 0), (0, 0, 0), 0.6)
    return o


def interleaved_count(width, n, phase):
    """Width of the planes a render with interleave=(n, phase) produces (rt_interleaved_count)."""
    return int(lib.rt_interleaved_count(width, n, phase))


def plan_groups(objs, with_metric=False):
    """rt_plan_groups: (group bounds [G,4], super-group bounds [S,4], group index per object [n])."""
    n = len(objs)
    cap = n // 1 + 1
    ng, ns = C.c_int32(0), C.c_int32(0)
    gb = np.zeros((cap, 4), dtype=np.float32)
    sb = np.zeros((cap, 4), dtype=np.float32)
    of = np.zeros(n, dtype=np.int32)
    sigma = (C.c_float * 3)()
    rc = lib.rt_plan_groups(objs, n, cap, C.byref(ng), C.byref(ns), gb.ctypes.data_as(C.POINTER(C.c_float)),
                            sb.ctypes.data_as(C.POINTER(C.c_float)), of.ctypes.data_as(C.POINTER(C.c_int32)), sigma)
    if rc:
        raise RtError(rc, lib.rt_last_error(None).decode())
    if with_metric:
        return gb[:ng.value], sb[:ns.value], of, np.array(list(sigma), dtype=np.float64)
    return gb[:ng.value], sb[:ns.value], of


def load_scene_file(path, width, height):
    """The scene description of raytracer_amd/host/rt_scenefile.hpp: returns (Object array, Camera).
    Numbers go through float32 exactly as the C++ loader's `float` reads do."""
    items, cam = [], None
    for lineno, line in enumerate(open(path), 1):
        words = line.split("#", 1)[0].split()
        if not words:
            continue
        kind, v = words[0], [float(np.float32(w)) for w in words[1:]]
        if kind == "camera" and len(v) == 7:
            cam = camera_init(v[0:3], v[3:6], v[6], width, height)
        elif kind == "sphere" and len(v) in (11, 14):
            items.append(sphere(material(v[4:7], v[7:10], v[10], v[11:14] if len(v) == 14 else (0, 0, 0)), v[0:3], v[3]))
        elif kind == "tiledplane" and len(v) in (6, 20):
            n = np.asarray(v[3:6], dtype=np.float32)
            # glm::normalize: v * (1 / sqrt(dot)), dot = (xx + yy) + zz, all in float
            d = np.float32(np.float32(n[0] * n[0] + n[1] * n[1]) + n[2] * n[2])
            n = n * (np.float32(1.0) / np.sqrt(d))
            white = material(v[6:9], v[9:12], v[12]) if len(v) == 20 else None
            black = material(v[13:16], v[16:19], v[19]) if len(v) == 20 else None
            items.append(tiled_plane(v[0:3], [float(x) for x in n], white, black))
        else:
            raise ValueError(f"{path}:{lineno}: bad statement {line.strip()!r}")
    if cam is None or not items:
        raise ValueError(f"{path}: needs a camera and at least one object")
    return objects(items), cam


def objects(items):
    arr = (Object * len(items))()
    for i, it in enumerate(items):
        arr[i] = it
    return arr
