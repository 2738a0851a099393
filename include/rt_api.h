/*
 * rt_api.h -- C ABI of the MI355X (gfx950) renderer for raytracer_cpp's per-pixel trace loop.
 *
 * The reference has no FFI seam: the path is entered by the direct call
 *     DrawScene(Bitmap*, int partitions, bool antialias)        raytracer_cpp/main.cpp:190 (called :248)
 * which reads the globals sceneObjects / pCamera (main.cpp:18-19) and fills the caller's Bitmap
 * through PutPixel (writebitmap.h:55-71).  This header is the seam a maintainer would bind to
 * replace that one call: upload the scene + camera once, render, download the framebuffer into
 * Bitmap::pData.  Everything is plain C: PODs, pointers and sizes; no C++/torch types.
 *
 * Conventions
 *   - every call returns RT_OK (0) or a negative rt_status; nothing throws or aborts.
 *   - the library copies what it is handed; the caller keeps ownership of its arrays.
 *   - a context is bound to ONE device (rt_create) or shares each frame between several
 *     (rt_create_multi: one process drives them all); the one-process-per-GPU scheme instead gives every
 *     rank its own context and a row band (rt_render_params.row_begin/row_end).
 *   - a context is not re-entrant; distinct contexts may be used from distinct threads.
 *   - there is NO CPU fallback: without a usable HIP device rt_create fails with RT_ERR_NO_DEVICE.
 */
#ifndef RT_API_H
#define RT_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_API_VERSION 3

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID_ARG = -1,
    RT_ERR_NO_DEVICE = -2,     /* no HIP device / runtime unusable: the product has no CPU path  */
    RT_ERR_HIP = -3,           /* a HIP call failed; text in rt_last_error                       */
    RT_ERR_BUFFER_TOO_SMALL = -4,
    RT_ERR_NO_SCENE = -5,      /* rt_render before rt_upload_scene                               */
    RT_ERR_NO_FRAME = -6,      /* rt_download_framebuffer before rt_render                       */
    RT_ERR_LIMIT = -7          /* scene exceeds a documented limit (objects, emitters, depth)    */
} rt_status;

/* Mirrors struct Material, raytracer_cpp/sceneobjects.h:3-9 (40 bytes, same field order). */
typedef struct rt_material {
    float albedo[3];
    float specular[3];
    float reflectance;
    float emissive[3];
} rt_material;

/* Mirrors enum class SceneObjectType, sceneobjects.h:11-15. */
enum { RT_OBJ_SPHERE = 0, RT_OBJ_TILED_PLANE = 1 };

/*
 * POD flattening of one SceneObject (sceneobjects.h:17-131), in LIST ORDER: the order of the
 * array is semantic (nearest-hit ties go to the lower index, main.cpp:71-79; lights are summed in
 * list order, main.cpp:112-184).
 *   sphere      (Sphere,     sceneobjects.h:36-74):   a = center, radius, mat = material
 *   tiled plane (TiledPlane, sceneobjects.h:88-131):  a = origin, b = normal,
 *                                                     mat = whiteMat (even tiles), mat2 = blackMat
 * Fields the reference leaves uninitialised (Material::emissive before main.cpp:45, both
 * TiledPlane materials) must be passed as 0.
 */
typedef struct rt_object {
    int32_t type;
    float a[3];
    float b[3];
    float radius;
    rt_material mat;
    rt_material mat2;
} rt_object;

/* The eight fields of struct Camera, raytracer_cpp/camera.h:6-16, as its constructor
 * (camera.h:18-45) leaves them.  rt_camera_init below is that constructor. */
typedef struct rt_camera {
    float position[3];
    float view_direction[3];
    float right[3];
    float up[3];
    float aspect_ratio;
    float half_angle;
    float image_width;
    float image_height;
} rt_camera;

/* Output planes rt_render fills (bit flags). */
enum {
    RT_OUT_RGBA8 = 1,   /* 4 B/px: r,g,b as PutPixel would store them (main.cpp:213-216), a = 255 */
    RT_OUT_F32X4 = 2,   /* 16 B/px: radiance after the 1/numSamples scale (main.cpp:210); w = rays
                           the pixel cost (primary + reflection + shadow)                       */
    RT_OUT_COST = 4     /* with RT_OUT_F32X4 and collect_stats: w = the work the pixel's rays took
                           instead (tests executed, exact tests weighted 32x, 300 per ray) -- the cost map the
                           row bands are balanced with; the flat sweep of large scenes keeps rays */
};

/* Download formats. */
enum {
    RT_FMT_RGB8 = 0,    /* exactly Bitmap::pData: W*y+x, 3 bytes r,g,b, top-down (writebitmap.h:5-17) */
    RT_FMT_RGBA8 = 1,
    RT_FMT_F32X4 = 2,
    RT_FMT_BMP_BODY = 3 /* BMP pixel array as WriteBitmap emits it after the 54-byte header
                           (writebitmap.h:161-178): bottom-up, b,g,r, rows padded to 4 bytes    */
};

/* Arithmetic contract of the trace kernel. */
enum {
    RT_MODE_EXACT = 0,  /* bit-identical to the reference built with -ffp-contract=off: reference
                           operation order, IEEE div/sqrt, deepest-first shading                 */
    RT_MODE_FAST = 1    /* same hits and shadow decisions, shading by a top-down throughput
                           accumulator (no per-segment records) and a float x^10: radiance within
                           1e-4 relative, 8-bit within +-1 LSB                                    */
};

/* How a ray visits the sphere list.  Every choice gives the same image, bit for bit. */
enum {
    RT_SWEEP_AUTO = 0,   /* flat below 64 spheres, grouped from there on; sorted for scenes of more than
                            4096 spheres when the render has 98 304 paths or more, its buffers fit and
                            nothing rules it out (rt_render_noalloc, RT_OUT_COST)                   */
    RT_SWEEP_FLAT = 1,   /* every ray is tested against every sphere, list streamed through LDS:
                            FindNearestObject's loop (main.cpp:65-82) as it stands                 */
    RT_SWEEP_GROUPED = 2,/* spheres are grouped at upload (8 per group along a Morton curve, 8
                            groups per super-group); a wave skips a group when all its rays pass
                            the group's bounding sphere at a distance.  Scenes of fewer than 64 spheres
                            take the flat sweep regardless                                         */
    RT_SWEEP_SORTED = 3  /* the grouped sweep with TraceRay's bounce levels (main.cpp:103-109) run as
                            separate launches: between two levels the hit points are sorted by where
                            they lie, so that a wave's 64 rays start next to one another whatever the
                            paths did before (large scenes of small reflective spheres, where paths
                            scatter after one bounce).  Needs device memory per path -- 64 B x (max_depth
                            + 1) + 196 B, held by the context -- and falls back to RT_SWEEP_GROUPED above
                            RT_SORTED_MAX_BYTES; not with rt_render_noalloc or RT_OUT_COST              */
};
#define RT_SORTED_MAX_BYTES ((size_t)24 << 30)

typedef struct rt_render_params {
    int32_t width;          /* pixel loop bounds (ImageWidth / ImageHeight, main.cpp:12-13)       */
    int32_t height;
    int32_t max_depth;      /* MAX_DEPTH, main.cpp:16: recursion while depth < max_depth          */
    int32_t spp;            /* numSamples, main.cpp:200: 1..64                                    */
    const float *sample_offsets; /* 2*spp floats (x,y); NULL = the reference pattern
                                    {(.1,.2),(.6,.5),(.8,.7),(.2,.8)} (main.cpp:202), spp 1 or 4   */
    int32_t row_begin;      /* render rows [row_begin,row_end); 0,0 = whole image                 */
    int32_t row_end;
    int32_t outputs;        /* RT_OUT_* flags; 0 = RT_OUT_RGBA8                                   */
    int32_t mode;           /* RT_MODE_*                                                          */
    int32_t collect_stats;  /* non-zero: count rays/tests in-kernel (slower; rt_get_stats)        */
    int32_t sweep;          /* RT_SWEEP_*                                                         */
    /* Options taken from raytracer_win (ray generation and accumulation only; TraceRay is unchanged).
     * All zero = raytracer_cpp's behaviour. */
    const float *lens_offsets; /* thin-lens depth of field, raytracer_win/camera.h:147-173: 2*spp floats,
                                  the lens-plane offset (along right, along up) of each sample -- what the
                                  reference draws with glm::circularRand(0.14f); NULL = pinhole camera  */
    float focal_length;     /* glm::length(focalPoint - position), camera.h:162; used with lens_offsets */
    int32_t accumulate;     /* non-zero: progressive running mean, raytracer_win/main.cpp:292-314:
                               float plane = (plane * k1 + vec4(color, 1)) * (1 / (k1 + 1)), k1 =
                               accum_index; needs RT_OUT_F32X4, whose w then is that mean of 1 instead
                               of the ray count; RGBA8 is the quantised mean (main.cpp:44-72)          */
    int32_t accum_index;    /* currentSample: 0 for the first frame (which ignores the old content)   */
    /* Multi-GPU without a cost estimate: the image is cut into vertical strips of RT_STRIP_COLS columns
     * and strip s belongs to the render with col_phase == s % col_interleave, so that every render
     * gets the same share of every row, cheap or expensive (the reference deals its rows to threads,
     * main.cpp:193-196; here rows near the horizon cost 12x the average, columns do not differ).  The output
     * planes then are rt_interleaved_count(width, n, phase) pixels wide: this render's strips side by
     * side.  Combines with row_begin/row_end.  0 or 1 = off. */
    int32_t col_interleave;
    int32_t col_phase;
    /* Scheduling hint, no effect on the image: the tile rows are launched starting at this image row and
     * wrap around.  A frame's slowest waves (rays grazing the plane at the horizon take 10x the average)
     * finish under the rest of the frame when they are launched first instead of wherever the horizon
     * happens to be; rt_split_bands' cost map tells where that is.  0 = top row first. */
    int32_t first_row;
    /* A context over several devices (rt_create_multi; ignored otherwise): how the frame is shared, RT_PLAN_*. */
    int32_t device_plan;
} rt_render_params;

/* How rt_render shares one frame between the devices of an rt_create_multi context (same image either way). */
enum {
    RT_PLAN_AUTO = 0,   /* strips when every device gets at least one (width >= 8 * n_devices), else bands         */
    RT_PLAN_BANDS = 1,  /* contiguous row bands balanced by a 1/16-resolution cost pre-pass on the first device; the
                           pre-pass runs (and BLOCKS the caller: render + download + prefix sum) whenever the scene,
                           the camera or the frame geometry changed since the last plan -- rt_stats.replans counts */
    RT_PLAN_STRIPS = 2  /* 8-column strips dealt round-robin (col_interleave inside the context): every device gets
                           the same share of every row, no pre-pass, nothing blocks; measured on one GPU: slowest of
                           8 parts 1.05x the mean against 1.12x for the bands                                    */
};

#define RT_STRIP_COLS 8

#define RT_MAX_DEVICES 8

typedef struct rt_stats {
    uint64_t primary_rays;      /* W * rows * spp                                                 */
    uint64_t reflection_rays;   /* recursive TraceRay invocations, main.cpp:108                   */
    uint64_t shadow_rays;       /* one per (hit, emissive object): the pruned count               */
    uint64_t hits;              /* TraceRay calls that found an object                            */
    uint64_t lit;               /* light contributions added, main.cpp:183                        */
    uint64_t sphere_tests;      /* ray-sphere tests the pruned algorithm requires                 */
    uint64_t plane_tests;
    uint64_t n_objects;
    uint64_t n_emitters;
    double kernel_ms;           /* device time of the last rt_render's trace kernel (HIP events)  */
    double kernel_ms_total;     /* cumulative over every render of this context ...               */
    uint64_t kernels_timed;     /* ... and how many launches that covers                          */
    uint64_t sphere_tests_run;  /* (ray, sphere) tests the kernel executed: = sphere_tests (+ odd-tail
                                   padding) for RT_SWEEP_FLAT, fewer for RT_SWEEP_GROUPED            */
    uint64_t group_tests_run;   /* (ray, group) bounding-sphere tests executed (RT_SWEEP_GROUPED)   */
    uint64_t exact_tests_run;   /* (ray, sphere) candidates of the filter that took the reference's
                                   exact test (the accept path of glm::intersectRaySphere and more)  */
    /* A context over several devices (rt_create_multi): the counters above are sums over the devices,
     * kernel_ms is the slowest device's (a frame takes as long as its slowest band); per device: */
    uint32_t n_devices;         /* 0 for a single-device context                                     */
    int32_t device_id[RT_MAX_DEVICES];
    int32_t device_row_begin[RT_MAX_DEVICES];   /* the band device i rendered in the last frame ...  */
    int32_t device_row_end[RT_MAX_DEVICES];
    double device_kernel_ms[RT_MAX_DEVICES];    /* ... the device time of its trace kernel ...        */
    double device_gather_ms[RT_MAX_DEVICES];    /* ... and of copying its band into the frame         */
    double gather_ms;           /* the slowest of those copies                                       */
    uint32_t replans;           /* RT_PLAN_BANDS: cost pre-passes this context has run so far (each one blocks)   */
    int32_t device_plan;        /* RT_PLAN_BANDS or RT_PLAN_STRIPS: what the last frame used                     */
    int32_t sweep;              /* RT_SWEEP_FLAT, _GROUPED or _SORTED: what the last render took (RT_SWEEP_AUTO resolved) */
    int32_t reserved;
} rt_stats;

typedef struct rt_context rt_context;

/* Opens a context on HIP device `device_id`.  Replaces nothing in the reference (it has no
 * device); owns the device copies of the scene, the framebuffer planes and one stream. */
int rt_create(rt_context **out, int device_id);
void rt_destroy(rt_context *ctx);

/* One context over n_devices HIP devices (1..RT_MAX_DEVICES; an id may repeat: its bands then share that
 * device).  Replaces the thread fan-out INSIDE DrawScene (main.cpp:193-227: `partitions` threads fill one
 * shared Bitmap): rt_upload_scene replicates the scene, rt_render shares the frame between the devices
 * (rt_render_params.device_plan: 8-column strips dealt round-robin, or cost-balanced row bands), renders the parts
 * concurrently and assembles the frame on device_ids[0] (peer copies over xGMI), which is where rt_download_framebuffer,
 * rt_pack_framebuffer and rt_framebuffer_device_ptr then find it; rt_get_stats adds per-device kernel and
 * copy times.  Every other call works as on a single-device context, except rt_render_noalloc and
 * col_interleave (RT_ERR_INVALID_ARG).  n_devices == 1 is rt_create. */
int rt_create_multi(rt_context **out, const int *device_ids, int n_devices);

/* Replaces the globals sceneObjects / pCamera (main.cpp:18-19) as filled by InitScene
 * (main.cpp:21-63).  May be called again between renders (camera sweeps). */
int rt_upload_scene(rt_context *ctx, const rt_object *objs, int32_t n_objs, const rt_camera *cam);

/* Replaces only pCamera (main.cpp:19): a camera sweep keeps the uploaded object list and changes
 * the camera between renders without touching the device copy of the scene.  Ordered after the
 * renders already enqueued (the camera travels with each launch). */
int rt_set_camera(rt_context *ctx, const rt_camera *cam);

/* Replaces DrawScene (main.cpp:190-228).  Returns after the frame is complete on the device, so
 * host-side chrono timing around it compares with the reference's timing (main.cpp:246-253). */
int rt_render(rt_context *ctx, const rt_render_params *params);

/* Replaces the PutPixel side effects of DrawScene: copies rows [row_begin,row_end) of the last
 * render into a caller-allocated buffer (dst_bytes >= rows * width * bytes-per-pixel). */
int rt_download_framebuffer(rt_context *ctx, void *dst, size_t dst_bytes, int32_t format);

/* Device-side half of the 24-bit download formats: enqueues the RGBA8 -> RT_FMT_RGB8 /
 * RT_FMT_BMP_BODY packing of the last render's band into caller-owned DEVICE memory (dst_bytes >=
 * the packed size rounded up to 4), so that a pipeline can copy the BMP file body straight out of
 * the GPU (replaces WriteBitmap's pixel loop, writebitmap.h:161-178).  Does not wait. */
int rt_pack_framebuffer(rt_context *ctx, int32_t format, void *device_dst, size_t dst_bytes);

/* Device pointer of an output plane of the last render (band rows only, row_begin first), for
 * device-side assembly (RCCL gather) without a host round trip.  Valid until the next
 * rt_render/rt_destroy on this context. */
int rt_framebuffer_device_ptr(rt_context *ctx, int32_t plane, void **out_ptr, size_t *out_bytes);

/* Lets the caller own the output planes (e.g. a slice of a gather buffer).  NULL = library-owned.
 * `bytes` must cover the band of the next render. */
int rt_bind_framebuffer(rt_context *ctx, int32_t plane, void *device_ptr, size_t bytes);

/* Stream the caller's work should be ordered against (hipStream_t as void*). */
int rt_get_stream(rt_context *ctx, void **out_stream);
/* Makes the context enqueue on the caller's stream (hipStream_t as void*; NULL = back to the
 * context's own stream), so renders order with the caller's other device work (gathers, copies)
 * without host synchronisation.  The caller keeps the stream alive while it is set. */
int rt_set_stream(rt_context *ctx, void *stream);
/* Enqueue-only render: like rt_render but returns without waiting; rt_sync waits.  The trace
 * kernel's HIP events are still recorded and read by rt_get_stats after rt_sync. */
int rt_render_async(rt_context *ctx, const rt_render_params *params);
int rt_sync(rt_context *ctx);

/* NO COUNTERPART IN THE REFERENCE.  For callers that cannot allocate: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  A variant of rt_render_async that allocates
 * nothing -- no hipMalloc/hipFree, no event or buffer creation -- and enqueues the same kernels: the
 * caller passes the output planes as device pointers (16-byte aligned; the plane that params->outputs
 * does not ask for may be NULL), and what rt_render_async keeps in buffers of its own -- the sample-split
 * plane (16 B per sample and pixel, used by the flat sweep of scenes of 48 spheres or more and by spp
 * that are not a power of two) and the ray counters -- lives in a fixed RT_NOALLOC_SCRATCH_BYTES array in
 * device memory that each GPU the library runs on holds until the process ends.  A render whose
 * sample-split plane exceeds it fails with RT_ERR_LIMIT (render fewer rows per call, or use rt_render);
 * renders that need no such plane have no size limit.  Every call on one GPU shares that GPU's array,
 * whichever context it comes from: calls that overlap on one GPU race; calls on one stream do not.
 * Returns without waiting (rt_sync, or the caller's stream); kernel_ms is not measured for these
 * renders.  The scene must have been uploaded (rt_upload_scene allocates; do that outside the loop). */
#define RT_NOALLOC_SCRATCH_BYTES (32u << 20)
#define RT_NOALLOC_STATS_BYTES 256u
int rt_render_noalloc(rt_context *ctx, const rt_render_params *params, void *rgba8_dev, size_t rgba8_bytes,
                      void *f32x4_dev, size_t f32x4_bytes);

int rt_get_stats(rt_context *ctx, rt_stats *out);
const char *rt_last_error(rt_context *ctx);   /* ctx may be NULL: last rt_create failure         */
int rt_api_version(void);

/* ---- host-side helpers (no device needed) ------------------------------------------------ */

/* Columns a render with (col_interleave, col_phase) = (n, phase) produces for an image `width` pixels
 * wide, and (rt_interleaved_index) the image column its output column `k` is.  Host only. */
int32_t rt_interleaved_count(int32_t width, int32_t n, int32_t phase);
int32_t rt_interleaved_index(int32_t k, int32_t n, int32_t phase);

/* The Camera constructor, camera.h:18-45 (double tan, float elsewhere). */
void rt_camera_init(rt_camera *cam, const float position[3], const float dir[3], float fov_degrees,
                    int32_t width, int32_t height);

/* Scene generators shared by the CLI, the tests and the bench.
 *   RT_SCENE_DEFAULT  InitScene, main.cpp:21-63 (6 objects)
 *   RT_SCENE_GRID1K   32x32 spheres + 4 emitters + tiled plane (1029 objects)
 *   RT_SCENE_GRID16K  128x128 reflective spheres + 8 emitters + tiled plane (16393 objects)
 * Returns the object count; writes at most `cap` objects (call with cap = 0 to size).
 * `frame`/`n_frames` select the camera of the GRID16K sweep; ignored elsewhere. */
enum { RT_SCENE_DEFAULT = 0, RT_SCENE_GRID1K = 1, RT_SCENE_GRID16K = 2 };
int32_t rt_make_scene(int32_t kind, int32_t width, int32_t height, int32_t frame, int32_t n_frames,
                      rt_object *objs, int32_t cap, rt_camera *cam);

/* Splits `height` rows into n contiguous bands of near-equal cost given per-row costs
 * (cost == NULL: equal rows).  Writes n+1 boundaries to `bounds`. */
int rt_split_bands(const float *row_cost, int32_t height, int32_t n, int32_t *bounds);

/* BMP file exactly as WriteBitmap produces it (writebitmap.h:88-182) from Bitmap::pData-layout
 * rgb8, with buffered rows instead of one fprintf per byte. */
int rt_write_bmp(const char *path, int32_t width, int32_t height, const uint8_t *rgb8);
/* The 54 header bytes of that file (writebitmap.h:90-155); header + an RT_FMT_BMP_BODY download is
 * the complete file. */
int rt_bmp_header(int32_t width, int32_t height, uint8_t out[54]);

/* Introspection, host only (no device needed): the grouping rt_upload_scene builds for
 * RT_SWEEP_GROUPED.  No counterpart in the reference.  group_bounds / super_bounds receive 4 floats per
 * group / super-group: centre g and radius R of the bound {x : |diag(sigma)(x - g)| <= R}, an ellipsoid
 * whose per-scene shape `sigma` (may be NULL) is 1 along the axis the groups are widest in and 2 or 4 along
 * axes they are thin in -- (1,1,1), plain bounding spheres, for scenes without such an axis;
 * group_of_object receives, per object,
 * the index of its group or -1 for a plane; super-group s holds groups [8s, 8s + 8).  Returns
 * RT_ERR_BUFFER_TOO_SMALL (with the counts filled in) when max_groups is too small. */
int rt_plan_groups(const rt_object *objs, int32_t n, int32_t max_groups, int32_t *n_groups, int32_t *n_supers,
                   float *group_bounds, float *super_bounds, int32_t *group_of_object, float sigma[3]);

#ifdef __cplusplus
}
#endif
#endif /* RT_API_H */
