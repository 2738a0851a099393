// rt_api.hip -- device half of the C ABI (include/rt_api.h): context, scene upload, render,
// framebuffer download.  One context = one HIP device = one stream.  There is no CPU fallback:
// every entry point that needs the device fails with an rt_status when HIP is unusable.
#include "rt_device.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

struct rt_context {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;     // own_stream or the caller's (rt_set_stream)
    std::vector<hipEvent_t> ev_free;  // recycled event pairs (2 entries per launch)
    std::deque<std::pair<hipEvent_t, hipEvent_t>> ev_pending;   // launches not yet read back
    double kernel_ms_last = 0.0, kernel_ms_total = 0.0;
    uint64_t kernels_timed = 0;
    char err[512] = {0};

    // scene
    void *scene_blob = nullptr;       // one allocation holding pairs | geom | sph_id | planes | mats | emit
    size_t scene_blob_bytes = 0;
    RtSceneDev scene = {};            // flat layout (list order)
    RtSceneDev scene_grouped = {};    // grouped layout (RT_SWEEP_GROUPED)
    rt_camera cam = {};
    bool have_scene = false;

    // frame
    void *own_plane[2] = {nullptr, nullptr};    // [0] rgba8, [1] f32x4 (library-owned)
    size_t own_bytes[2] = {0, 0};
    void *bound_plane[2] = {nullptr, nullptr};  // caller-owned (rt_bind_framebuffer)
    size_t bound_bytes[2] = {0, 0};
    void *cur_plane[2] = {nullptr, nullptr};    // planes of the last render
    RtFrameDev frame = {};
    bool have_frame = false;
    int last_outputs = 0;

    hipStream_t side_stream = nullptr;   // RT_SWEEP_SORTED: level 0's far segments run beside the sorted levels
    hipEvent_t ev_level0 = nullptr, ev_side = nullptr;
    void *sorted_blob = nullptr;      // RT_SWEEP_SORTED: records | entry queues | permutation | cell counters | counts (one allocation)
    size_t sorted_bytes = 0;
    float cell_org[3] = {0, 0, 0}, cell_inv[3] = {0, 0, 0};   // the scene's grid of 2^RT_SORTED_CELL_BITS cells (set by rt_upload_scene)
    float far_bound_sorted = INFINITY;   // RtSceneDev::far_bound as RT_SWEEP_SORTED uses it
    int32_t cell_bits[3] = {RT_SORTED_CELL_BITS / 3, RT_SORTED_CELL_BITS / 3, RT_SORTED_CELL_BITS / 3};
    void *sample_buf = nullptr;       // sample split: one float4 per (sample, pixel)
    size_t sample_bytes = 0;
    int sample_split = 1;
    void *pack_buf = nullptr;         // staging for the 24-bit download formats
    size_t pack_bytes = 0;
    RtStatsDev *stats_dev = nullptr;
    unsigned long long *region_dev = nullptr;   // RT_AMD_REGION_DUMP=1 (profiling builds): the last launch's cycles by region
    void *scratch_dev = nullptr;      // this device's copy of `scratch` (rt_render_noalloc)
    const RtStatsDev *stats_src = nullptr;   // where the last render's counters are: stats_dev or scratch_dev
    rt_stats stats = {};
    bool stats_pending = false;
    // ---- a context over several devices (rt_create_multi): this root object holds the assembled frame on
    // parts[0]'s device and one single-device context per entry of device_ids
    std::vector<rt_context *> parts;
    std::vector<int32_t> bounds;      // row bands of the last plan: part i renders [bounds[i], bounds[i+1])
    struct PlanKey {
        int32_t w, h, depth, y0, y1, spp, sweep;
        uint64_t scene_version;
        bool operator==(const PlanKey &o) const     // field by field: the struct has padding bytes
        {
            return w == o.w && h == o.h && depth == o.depth && y0 == o.y0 && y1 == o.y1 && spp == o.spp && sweep == o.sweep &&
                   scene_version == o.scene_version;
        }
    } plan_key = {};
    bool have_plan = false;
    uint32_t replans = 0;             // cost pre-passes run so far (RT_PLAN_BANDS)
    int last_plan = 0;                // RT_PLAN_BANDS / RT_PLAN_STRIPS of the last frame
    std::vector<void *> stage[2];     // RT_PLAN_STRIPS: per part, a copy of its planes on the root's device (parts on other devices)
    std::vector<size_t> stage_bytes[2];
    uint64_t scene_version = 0;       // bumped by rt_upload_scene / rt_set_camera
    std::vector<hipEvent_t> part_done, copy_begin;   // per part: band copied into the frame; its copy began
    hipEvent_t root_ready = nullptr;  // the root stream's work before a render (the caller's, on its stream)
    double gather_ms_last = 0.0;
    bool gather_pending = false;
    int lds_budget = 39 * 1024;       // per workgroup: 4 workgroups (+ their few static bytes) fit a CU's 160 KiB
    bool simple_kernel = false;       // RT_AMD_KERNEL=simple: the straight-transcription kernel (A/B)
};

static char g_create_err[512] = "";

// ---- rt_render_noalloc's scratch ------------------------------------------------------------------
// One array per GPU the library runs on, held until the process ends: [0, 256) the RtStatsDev counters,
// [256, ...) the sample-split plane (one float4 per sample and pixel).  The name `scratch` is a
// host-side stand-in in host code, not a device address: every use goes through the per-device pointer
// that hipGetSymbolAddress returned with that device current (scratch_for_device, called by rt_create).
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];
static_assert(sizeof(RtStatsDev) <= RT_NOALLOC_STATS_BYTES, "the counters must fit the head of the scratch array");
constexpr int kMaxDevices = 64;
static void *g_scratch_ptr[kMaxDevices];      // each GPU holds its own copy of the array at its own address
static std::mutex g_scratch_mutex;

// The calling thread's current device must be `device`.
static hipError_t scratch_for_device(int device, void **out)
{
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *ptr = nullptr;
        hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!ptr) return hipErrorInvalidSymbol;
        g_scratch_ptr[device] = ptr;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

static int fail(rt_context *ctx, int code, const char *fmt, ...)
{
    char *dst = ctx ? ctx->err : g_create_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return code;
}

#define RT_HIP(ctx, call)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(ctx, RT_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                   \
    } while (0)

// The library reads no tuning knobs from the environment.  Three DEBUG switches remain for the test suite and the
// profiling scripts, and only with RT_AMD_DEBUG=1 set as well: RT_AMD_KERNEL=simple (the straight-transcription kernel as
// an independent witness), RT_AMD_FAR_FRACTION (sends all / no lanes through dense_sweep) and RT_AMD_REGION_DUMP
// (profiling builds: wave-cycles by region).
static const char *debug_env(const char *name)
{
    const char *on = getenv("RT_AMD_DEBUG");
    return (on && on[0] == '1') ? getenv(name) : nullptr;
}

static const float kRefPattern[8] = {0.1f, 0.2f, 0.6f, 0.5f, 0.8f, 0.7f, 0.2f, 0.8f};  // main.cpp:202

static bool emissive_nonzero(const rt_material &m)
{
    return !(m.emissive[0] == 0.0f && m.emissive[1] == 0.0f && m.emissive[2] == 0.0f);  // main.cpp:131
}

// Reads back every finished launch's event pair into the running totals.  `block`: wait for all.
static int drain_events(rt_context *ctx, bool block)
{
    while (!ctx->ev_pending.empty()) {
        auto p = ctx->ev_pending.front();
        if (block) RT_HIP(ctx, hipEventSynchronize(p.second));
        else if (hipEventQuery(p.second) != hipSuccess) break;
        float ms = 0.0f;
        RT_HIP(ctx, hipEventElapsedTime(&ms, p.first, p.second));
        ctx->kernel_ms_last = (double)ms;
        ctx->kernel_ms_total += (double)ms;
        ctx->kernels_timed++;
        ctx->ev_free.push_back(p.first);
        ctx->ev_free.push_back(p.second);
        ctx->ev_pending.pop_front();
    }
    return RT_OK;
}

// ---- RGBA8 -> 24-bit packing on the device ---------------------------------------------------
// RT_FMT_RGB8: Bitmap::pData (writebitmap.h:5-17), r,g,b per pixel, rows top-down, no padding.
// RT_FMT_BMP_BODY: what WriteBitmap emits after its header (writebitmap.h:161-178): rows bottom-up,
// b,g,r per pixel, each row padded with zeros to a multiple of 4 bytes.
static size_t packed_bytes(const RtFrameDev &fr, int format)
{
    const size_t w = (size_t)fr.width, rows = (size_t)(fr.row_end - fr.row_begin);
    const size_t stride = format == RT_FMT_BMP_BODY ? ((w * 3 + 3) & ~(size_t)3) : w * 3;
    return stride * rows;
}

// One thread = four consecutive pixels of one output row = three output dwords: 16 B read, 12 B written
// per thread, both contiguous across the wave.  RT_FMT_RGB8 has no row padding, so the whole band is
// one "row" of width*rows pixels; RT_FMT_BMP_BODY rows are flipped, b,g,r and zero-padded to 4 bytes.
__global__ __launch_bounds__(256) void pack_rgb24_kernel(const uint32_t *__restrict__ rgba, uint32_t *__restrict__ out,
                                                         unsigned px_per_row, unsigned n_rows, int bmp_body,
                                                         unsigned groups_per_row, unsigned dw_per_row)
{
    const unsigned j = blockIdx.x * 256u + threadIdx.x;
    const unsigned row = j / groups_per_row, g = j - row * groups_per_row;
    if (row >= n_rows) return;
    const unsigned src_row = bmp_body ? n_rows - 1u - row : row;                       // bottom-up
    const uint32_t *src = rgba + (size_t)src_row * (size_t)px_per_row + 4u * g;
    const unsigned left = px_per_row - 4u * g;                                         // pixels from here to the row's end
    uint32_t c[4];
    if (left >= 4u && ((uintptr_t)src & 15u) == 0) {                                   // one 16-byte load
        const uint4 v = *reinterpret_cast<const uint4 *>(src);
        c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
    } else {
#pragma unroll
        for (unsigned k = 0; k < 4; k++) c[k] = k < left ? src[k] : 0u;
    }
#pragma unroll
    for (unsigned k = 0; k < 4; k++) {
        const uint32_t p = c[k];
        c[k] = bmp_body ? ((p >> 16) & 0xffu) | (p & 0xff00u) | ((p & 0xffu) << 16)    // b,g,r
                        : p & 0xffffffu;                                               // r,g,b
    }
    const uint32_t w[3] = {c[0] | (c[1] << 24), (c[1] >> 8) | (c[2] << 16), (c[2] >> 16) | (c[3] << 8)};
    uint32_t *dst = out + (size_t)row * (size_t)dw_per_row + 3u * g;
#pragma unroll
    for (unsigned k = 0; k < 3; k++)
        if (3u * g + k < dw_per_row) dst[k] = w[k];
}

static hipError_t launch_pack(const RtFrameDev &fr, const uint32_t *rgba, uint32_t *out, int format, hipStream_t stream)
{
    const size_t rows = (size_t)(fr.row_end - fr.row_begin), width = (size_t)fr.width;
    if (rows == 0 || width == 0) return hipSuccess;
    const bool body = format == RT_FMT_BMP_BODY;
    const size_t px_per_row = body ? width : width * rows, n_rows = body ? rows : 1;
    const size_t groups = (px_per_row + 3) / 4, dw_per_row = (px_per_row * 3 + 3) / 4;
    const size_t threads = groups * n_rows;
    if (px_per_row > 0xffffffffull || threads > 0xffffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_rgb24_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, rgba, out,
                       (unsigned)px_per_row, (unsigned)n_rows, body ? 1 : 0, (unsigned)groups, (unsigned)dw_per_row);
    return hipGetLastError();
}

// RT_PLAN_STRIPS, assembly: part `phase` of `n` holds its 8-column strips side by side (w_part columns); column x of it
// is image column ((x / 8) * n + phase) * 8 + x % 8 (rt_render_params.col_interleave).  One thread per element (a
// 4-byte RGBA8 word or a 16-byte float4), rows coalesced in 8-element runs on the writing side.
template <class T>
__global__ __launch_bounds__(256) void scatter_strips_kernel(T *__restrict__ dst, const T *__restrict__ src, unsigned w_img,
                                                             unsigned w_part, unsigned rows, unsigned n, unsigned phase)
{
    const size_t idx = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (idx >= (size_t)w_part * rows) return;
    const unsigned row = (unsigned)(idx / w_part), x = (unsigned)(idx - (size_t)row * w_part);
    const unsigned xi = (((x >> 3) * n + phase) << 3) | (x & 7u);
    dst[(size_t)row * w_img + xi] = src[idx];
}

static hipError_t launch_scatter(void *dst, const void *src, int plane, int w_img, int w_part, int rows, int n, int phase, hipStream_t stream)
{
    const size_t elems = (size_t)w_part * (size_t)rows;
    if (!elems) return hipSuccess;
    const dim3 grid((unsigned)((elems + 255) / 256)), block(256);
    if (plane == 0)
        hipLaunchKernelGGL(scatter_strips_kernel<uint32_t>, grid, block, 0, stream, (uint32_t *)dst, (const uint32_t *)src, (unsigned)w_img,
                           (unsigned)w_part, (unsigned)rows, (unsigned)n, (unsigned)phase);
    else
        hipLaunchKernelGGL(scatter_strips_kernel<float4>, grid, block, 0, stream, (float4 *)dst, (const float4 *)src, (unsigned)w_img,
                           (unsigned)w_part, (unsigned)rows, (unsigned)n, (unsigned)phase);
    return hipGetLastError();
}

static int take_event(rt_context *ctx, hipEvent_t *out)
{
    if (!ctx->ev_free.empty()) {
        *out = ctx->ev_free.back();
        ctx->ev_free.pop_back();
        return RT_OK;
    }
    RT_HIP(ctx, hipEventCreate(out));
    return RT_OK;
}

extern "C" {

int rt_create(rt_context **out, int device_id)
{
    if (!out) return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, RT_ERR_NO_DEVICE, "rt_create: no HIP device (%s); this library has no CPU path",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device_id < 0 || device_id >= n)
        return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create: device %d out of range [0,%d)", device_id, n);
    rt_context *ctx = new rt_context();
    ctx->device = device_id;
    const char *kern = debug_env("RT_AMD_KERNEL");
    ctx->simple_kernel = kern && strcmp(kern, "simple") == 0;
    e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    ctx->stream = ctx->own_stream;
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->stats_dev, sizeof(RtStatsDev));
    if (e == hipSuccess) e = scratch_for_device(device_id, &ctx->scratch_dev);     // device_id is current here
    if (e == hipSuccess && debug_env("RT_AMD_REGION_DUMP")) e = hipMalloc((void **)&ctx->region_dev, RT_N_REGIONS * sizeof(unsigned long long));
    if (e != hipSuccess) {
        fail(nullptr, RT_ERR_HIP, "rt_create: %s", hipGetErrorString(e));
        rt_destroy(ctx);
        return RT_ERR_HIP;
    }
    *out = ctx;
    return RT_OK;
}

int rt_create_multi(rt_context **out, const int *device_ids, int n_devices)
{
    if (!out) return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create_multi: out is NULL");
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > RT_MAX_DEVICES)
        return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create_multi: device_ids NULL or n_devices outside [1,%d]", RT_MAX_DEVICES);
    if (n_devices == 1) return rt_create(out, device_ids[0]);
    rt_context *root = nullptr;
    int rc = rt_create(&root, device_ids[0]);          // owns the assembled frame and the stream callers order against
    if (rc) return rc;
    for (int i = 0; i < n_devices && rc == RT_OK; i++) {
        rt_context *part = nullptr;
        rc = rt_create(&part, device_ids[i]);          // the same device may appear more than once: its bands then share it
        if (rc == RT_OK) root->parts.push_back(part);
    }
    if (rc == RT_OK) {
        // a band's way into the frame: peer access where the devices differ (hipMemcpyPeerAsync falls back to
        // staging through the host when it cannot be had)
        for (int i = 0; i < n_devices; i++) {
            if (device_ids[i] == device_ids[0]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, device_ids[0], device_ids[i]) == hipSuccess && can) {
                (void)hipSetDevice(device_ids[0]);
                hipError_t e = hipDeviceEnablePeerAccess(device_ids[i], 0);
                if (e != hipSuccess) (void)hipGetLastError();       // already enabled
            }
            // and the other way round: the copies are issued on the part's stream and write into the root's memory
            can = 0;
            if (hipDeviceCanAccessPeer(&can, device_ids[i], device_ids[0]) == hipSuccess && can) {
                (void)hipSetDevice(device_ids[i]);
                hipError_t e = hipDeviceEnablePeerAccess(device_ids[0], 0);
                if (e != hipSuccess) (void)hipGetLastError();
            }
        }
        hipError_t e = hipSetDevice(device_ids[0]);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&root->root_ready, hipEventDisableTiming);
        for (int i = 0; i < n_devices && e == hipSuccess; i++) {
            hipEvent_t a = nullptr, b = nullptr;
            e = hipSetDevice(device_ids[i]);
            if (e == hipSuccess) e = hipEventCreate(&a);
            if (e == hipSuccess) e = hipEventCreate(&b);
            root->copy_begin.push_back(a);
            root->part_done.push_back(b);
        }
        if (e != hipSuccess) rc = fail(nullptr, RT_ERR_HIP, "rt_create_multi: %s", hipGetErrorString(e));
    }
    if (rc) {
        rt_destroy(root);
        return rc;
    }
    *out = root;
    return RT_OK;
}

void rt_destroy(rt_context *ctx)
{
    if (!ctx) return;
    for (rt_context *part : ctx->parts) rt_destroy(part);
    for (hipEvent_t e : ctx->part_done) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->copy_begin) if (e) (void)hipEventDestroy(e);
    if (ctx->root_ready) (void)hipEventDestroy(ctx->root_ready);
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (int k = 0; k < 2; k++)
        for (void *p : ctx->stage[k]) if (p) (void)hipFree(p);
    if (ctx->scene_blob) (void)hipFree(ctx->scene_blob);
    for (int i = 0; i < 2; i++)
        if (ctx->own_plane[i]) (void)hipFree(ctx->own_plane[i]);
    if (ctx->stats_dev) (void)hipFree(ctx->stats_dev);
    if (ctx->region_dev) (void)hipFree(ctx->region_dev);
    if (ctx->pack_buf) (void)hipFree(ctx->pack_buf);
    if (ctx->sample_buf) (void)hipFree(ctx->sample_buf);
    if (ctx->sorted_blob) (void)hipFree(ctx->sorted_blob);
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    if (ctx->ev_level0) (void)hipEventDestroy(ctx->ev_level0);
    if (ctx->ev_side) (void)hipEventDestroy(ctx->ev_side);
    for (auto &p : ctx->ev_pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (hipEvent_t e : ctx->ev_free) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char *rt_last_error(rt_context *ctx) { return ctx ? ctx->err : g_create_err; }

// ---- grouped layout (RT_SWEEP_GROUPED) ---------------------------------------------------------
// Spheres are sorted along a Morton curve through their centres and cut greedily into groups of at
// most 2*RT_GROUP_PAIRS; a group is closed early when the next sphere would blow its bounding
// radius past a cap derived from the typical spacing, so a far-away light or a huge sphere ends up
// in a group of its own instead of inflating its neighbours' bound.
struct GroupedScene {
    std::vector<float4> pairs;     // 2 * RT_GROUP_PAIRS * n_groups
    std::vector<int32_t> ids;      // 2 * RT_GROUP_PAIRS * n_groups (list index, -1 = padding)
    std::vector<int32_t> slot_of;  // geom index -> slot
    std::vector<float4> heads;     // 2 * ceil(n_groups / 2)
    std::vector<float4> supers;    // 2 * ceil(n_supers / 2)
    std::vector<float4> supers2;   // 2 * ceil(n_supers2 / 2): bounds of RT_SUPER_GROUPS consecutive super-groups
    int n_groups = 0, n_supers = 0, n_supers2 = 0;
    int n_tail_pairs = 0;                  // records after the groups' that every sweep tests (the emissive spheres, when few)
    float sigma[3] = {1.0f, 1.0f, 1.0f};   // the metric the bounds are measured in: |x|_S = |diag(sigma) x|, min sigma = 1
    float kappa = 1.0f;                    // max sigma
    float box_lo[3] = {-INFINITY, -INFINITY, -INFINITY}, box_hi[3] = {INFINITY, INFINITY, INFINITY};   // around the spheres in groups (not the tail's)
};

static uint32_t spread10(uint32_t v)
{
    v &= 1023u;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// max over the points x of the sphere |x - c| = r of |diag(sig)(x - g)|, v = c - g: how far, in the bounds'
// norm, a member sphere reaches from the bound's centre.  Maximising sum sig_a^2 (v_a + r n_a)^2 over |n| = 1
// gives n_a = sig_a^2 v_a r / (lam - sig_a^2 r^2) with lam > kappa^2 r^2 chosen so that |n| = 1 (bisection; |n|
// decreases in lam); when v has no component along the axes of largest sig and that equation has no root,
// the rest of n goes along such an axis.  Never more than |S v| + kappa r, the bound used as a safety net.
static double reach_S(const double v[3], double r, const double sig[3])
{
    double sv2 = 0, kap = 0;
    for (int a = 0; a < 3; a++) { sv2 += sig[a] * sig[a] * v[a] * v[a]; kap = std::max(kap, sig[a]); }
    const double crude = std::sqrt(sv2) + kap * r;
    if (!(r > 0) || !std::isfinite(crude) || (sig[0] == sig[1] && sig[1] == sig[2])) return crude;
    const double top = kap * kap * r * r;
    auto n_of = [&](double lam, double n[3]) {
        double nn = 0;
        for (int a = 0; a < 3; a++) {
            const double den = lam - sig[a] * sig[a] * r * r;
            n[a] = den > 0 ? sig[a] * sig[a] * v[a] * r / den : 0.0;
            nn += n[a] * n[a];
        }
        return nn;
    };
    double n[3];
    double lo = top, hi = top + 4.0 * (std::sqrt(sv2) * kap * r + top) + 1e-300;    // |n|(hi) < 1
    double best = 0;
    auto value = [&](const double nv[3]) {
        double t = 0;
        for (int a = 0; a < 3; a++) t += sig[a] * sig[a] * (v[a] + r * nv[a]) * (v[a] + r * nv[a]);
        return std::sqrt(t);
    };
    for (int it = 0; it < 200; it++) {
        const double mid = 0.5 * (lo + hi);
        if (n_of(mid, n) > 1.0) lo = mid; else hi = mid;
    }
    // candidates: the root (normalised), and the hard case (components off the top axes at lam = top, the rest
    // along a top axis, either sign)
    double nn = n_of(hi, n);
    if (nn > 0) { const double k = 1.0 / std::sqrt(nn); double m[3] = {n[0] * k, n[1] * k, n[2] * k}; best = std::max(best, value(m)); }
    double h[3] = {0, 0, 0}, hh = 0;
    for (int a = 0; a < 3; a++)
        if (sig[a] < kap) { h[a] = sig[a] * sig[a] * v[a] * r / (top - sig[a] * sig[a] * r * r); hh += h[a] * h[a]; }
    if (hh <= 1.0)
        for (int a = 0; a < 3; a++)
            if (sig[a] == kap)
                for (double sgn : {-1.0, 1.0}) { double m[3] = {h[0], h[1], h[2]}; m[a] = sgn * std::sqrt(1.0 - hh); best = std::max(best, value(m)); }
    return std::min(crude, best * (1.0 + 1e-9));
}

// Emissive spheres are kept apart: every shadow ray is aimed at one of them, so whatever bounds them they are visited by
// every sweep that carries shadow rays -- a light in a group of its own (what the radius cap gives a far-away light)
// costs such a sweep a visit of four records per light, three of them padding.  Up to 8 emissive spheres become the
// TAIL instead: pair records after the last group's, under no bound, that every sweep tests (n_tail_pairs; measured
// on the 1028-sphere scene with its 4 lights: 5.77 -> 5.2 ms per frame).  More than 8 are grouped among themselves,
// 8 to a group in the order of the curve, after all other groups, under bounds like any group.
static void build_groups(const std::vector<float4> &geom, const std::vector<float> &radius,
                         const std::vector<int32_t> &sph_id, const std::vector<char> &emissive, GroupedScene &out)
{
    const size_t n = geom.size();
    const int kMax = 2 * RT_GROUP_PAIRS;
    out = GroupedScene();
    out.slot_of.assign(n, -1);
    if (n == 0) return;

    std::vector<char> finite(n);
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (size_t i = 0; i < n; i++) {
        const float c[3] = {geom[i].x, geom[i].y, geom[i].z};
        finite[i] = std::isfinite(c[0]) && std::isfinite(c[1]) && std::isfinite(c[2]) && std::isfinite(radius[i]);
        if (!finite[i]) continue;
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], (double)c[a]); hi[a] = std::max(hi[a], (double)c[a]); }
    }
    std::vector<std::pair<uint32_t, uint32_t>> order(n);     // (morton, geom index)
    for (size_t i = 0; i < n; i++) {
        uint32_t code = 0xFFFFFFFFu;                          // non-finite spheres go last
        if (finite[i]) {
            const float c[3] = {geom[i].x, geom[i].y, geom[i].z};
            uint32_t q[3];
            for (int a = 0; a < 3; a++) {
                const double ext = hi[a] - lo[a];
                q[a] = ext > 0 ? (uint32_t)std::min(1023.0, (c[a] - lo[a]) / ext * 1024.0) : 0u;
            }
            code = (spread10(q[0]) << 2) | (spread10(q[1]) << 1) | spread10(q[2]);
        }
        order[i] = {code, (uint32_t)i};
    }
    std::stable_sort(order.begin(), order.end());

    // typical spacing along the curve (median distance between neighbours) and typical radius
    std::vector<double> gaps, rads;
    for (size_t k = 0; k < n; k++) {
        const size_t i = order[k].second;
        if (!finite[i]) continue;
        rads.push_back(std::fabs((double)radius[i]));
        if (k + 1 < n && finite[order[k + 1].second]) {
            const size_t j = order[k + 1].second;
            const double dx = geom[i].x - geom[j].x, dy = geom[i].y - geom[j].y, dz = geom[i].z - geom[j].z;
            gaps.push_back(std::sqrt(dx * dx + dy * dy + dz * dz));
        }
    }
    auto median = [](std::vector<double> &v) {
        if (v.empty()) return 0.0;
        std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end());
        return v[v.size() / 2];
    };
    const double cap = 3.5 * median(gaps) + 2.0 * median(rads);

    struct Group { std::vector<uint32_t> members; double c[3]; double R; };
    std::vector<Group> groups;
    auto bound_of = [&](const std::vector<uint32_t> &m, double c[3]) {
        double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
        for (uint32_t i : m) {
            const double cc[3] = {geom[i].x, geom[i].y, geom[i].z}, r = std::fabs((double)radius[i]);
            for (int a = 0; a < 3; a++) { blo[a] = std::min(blo[a], cc[a] - r); bhi[a] = std::max(bhi[a], cc[a] + r); }
        }
        for (int a = 0; a < 3; a++) c[a] = 0.5 * (blo[a] + bhi[a]);
        double R = 0;
        for (uint32_t i : m) {
            const double dx = geom[i].x - c[0], dy = geom[i].y - c[1], dz = geom[i].z - c[2];
            R = std::max(R, std::sqrt(dx * dx + dy * dy + dz * dz) + std::fabs((double)radius[i]));
        }
        return R;
    };
    std::vector<uint32_t> cur;
    auto close = [&]() {
        if (cur.empty()) return;
        Group g;
        g.members = cur;
        g.R = bound_of(cur, g.c);
        groups.push_back(g);
        cur.clear();
    };
    for (size_t k = 0; k < n; k++) {
        const uint32_t i = order[k].second;
        if (!finite[i]) {                   // never hit by anything (dSquared is NaN or r*r is inf): own group
            close();
            cur.push_back(i);
            close();
            continue;
        }
        if (emissive[i]) continue;      // below
        if ((int)cur.size() == kMax) close();
        if (!cur.empty()) {
            std::vector<uint32_t> trial = cur;
            trial.push_back(i);
            double c[3];
            const double R = bound_of(trial, c);
            if (R > cap + std::fabs((double)radius[i])) close();
        }
        cur.push_back(i);
    }
    close();
    std::vector<uint32_t> tail;                                  // at most kMax emissive spheres: no group, no bound
    {
        std::vector<uint32_t> em;
        for (size_t k = 0; k < n; k++)
            if (finite[order[k].second] && emissive[order[k].second]) em.push_back(order[k].second);
        if (em.size() <= (size_t)kMax) {
            tail = em;
        } else {
            for (uint32_t i : em) {                              // along the same curve, 8 to a group
                if ((int)cur.size() == kMax) close();
                cur.push_back(i);
            }
            close();
        }
    }

    // The metric of the bounds.  Spheres standing on a floor make groups that are wide and flat; a round
    // bound of such a group is mostly air above it, and every ray that rises through that air visits the
    // group for nothing.  So bounds are ELLIPSOIDS with one shape for the whole scene: spheres in the norm
    // |x|_S = |diag(sigma) x|, sigma_a >= 1 stretching the axes along which the groups are thin (sigma_a ~ 1 /
    // median extent of a group along a, the widest axis at 1, capped at 4; powers of two so that the scaling is
    // exact).  Scenes whose groups are round get sigma = 1: plain bounding spheres.
    {
        std::vector<double> exts[3];
        for (const Group &g : groups) {
            if (!std::isfinite(g.R)) continue;
            double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
            for (uint32_t i : g.members) {
                const double cc[3] = {geom[i].x, geom[i].y, geom[i].z}, r = std::fabs((double)radius[i]);
                for (int a = 0; a < 3; a++) { blo[a] = std::min(blo[a], cc[a] - r); bhi[a] = std::max(bhi[a], cc[a] + r); }
            }
            for (int a = 0; a < 3; a++) exts[a].push_back(bhi[a] - blo[a]);
        }
        const size_t cnt = exts[0].size();
        const double ext[3] = {median(exts[0]), median(exts[1]), median(exts[2])};      // the typical group, not the outliers
        const double widest = std::max(ext[0], std::max(ext[1], ext[2]));
        for (int a = 0; a < 3; a++) {
            double sg = (cnt && ext[a] > 0 && std::isfinite(widest)) ? widest / ext[a] : 1.0;
            sg = sg >= 3.0 ? 4.0 : sg >= 1.6 ? 2.0 : 1.0;
            out.sigma[a] = (float)sg;
            out.kappa = std::max(out.kappa, (float)sg);
        }
    }
    const double sig[3] = {out.sigma[0], out.sigma[1], out.sigma[2]};
    // centre = middle of the members' box; radius = the farthest any member sphere reaches from it in the norm
    auto bound_S = [&](const std::vector<uint32_t> &m, size_t begin, size_t end, const std::vector<Group> *of_groups, double c[3]) {
        double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
        auto each = [&](auto &&fn) {
            if (of_groups) { for (size_t g = begin; g < end; g++) for (uint32_t i : (*of_groups)[g].members) fn(i); }
            else { for (uint32_t i : m) fn(i); }
        };
        bool ok = true;
        each([&](uint32_t i) {
            const double cc[3] = {geom[i].x, geom[i].y, geom[i].z}, r = std::fabs((double)radius[i]);
            ok = ok && finite[i];
            for (int a = 0; a < 3; a++) { blo[a] = std::min(blo[a], cc[a] - r); bhi[a] = std::max(bhi[a], cc[a] + r); }
        });
        if (!ok) { c[0] = c[1] = c[2] = 0.0; return (double)INFINITY; }
        for (int a = 0; a < 3; a++) c[a] = 0.5 * (blo[a] + bhi[a]);
        double R = 0;
        each([&](uint32_t i) {
            const double v[3] = {geom[i].x - c[0], geom[i].y - c[1], geom[i].z - c[2]};
            R = std::max(R, reach_S(v, std::fabs((double)radius[i]), sig));
        });
        return R;
    };
    for (Group &g : groups) g.R = bound_S(g.members, 0, 0, nullptr, g.c);

    out.n_groups = (int)groups.size();
    const float kNoHit = 3.402823466e+38f;
    out.n_tail_pairs = (int)((tail.size() + 1) / 2);
    const size_t n_records = (size_t)RT_GROUP_PAIRS * groups.size() + (size_t)out.n_tail_pairs;
    out.pairs.assign(2 * n_records, make_float4(0.0f, 0.0f, kNoHit, kNoHit));
    out.ids.assign(2 * n_records, -1);
    for (size_t p = 0; p < n_records; p++) out.pairs[2 * p] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (size_t g = 0; g <= groups.size(); g++) {       // (the last round: the tail, laid out like one more group)
        if (g == groups.size() && tail.empty()) break;
        std::vector<uint32_t> &m = g < groups.size() ? groups[g].members : tail;
        std::sort(m.begin(), m.end());      // list order inside a group: most ties resolve by slot order already
        for (size_t j = 0; j < m.size(); j++) {
            const size_t slot = g * kMax + j, pr = slot >> 1;
            const float4 s = geom[m[j]];
            float4 &A = out.pairs[2 * pr], &B = out.pairs[2 * pr + 1];
            if (slot & 1) { A.y = s.x; A.w = s.y; B.y = s.z; B.w = -s.w; }
            else          { A.x = s.x; A.z = s.y; B.x = s.z; B.z = -s.w; }
            out.ids[slot] = sph_id[m[j]];
            out.slot_of[m[j]] = (int32_t)slot;
        }
    }
    // bounding spheres, two per record; the centre is rounded to float, so the radius is widened by
    // that displacement and then rounded up
    auto pack_heads = [&](const std::vector<std::array<double, 4>> &b, std::vector<float4> &dst) {
        dst.assign(2 * ((b.size() + 1) / 2), make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        for (size_t g = 0; g < b.size(); g++) {
            float c[3];
            double d2 = 0;
            for (int a = 0; a < 3; a++) {
                c[a] = (float)b[g][a];
                d2 += sig[a] * sig[a] * ((double)c[a] - b[g][a]) * ((double)c[a] - b[g][a]);
            }
            const double Rd = (b[g][3] + std::sqrt(d2)) * (1.0 + 0x1p-20);
            // a NaN or overflowing bound: always visit
            const float R = (Rd == Rd) && Rd < 1e37 ? std::nextafterf((float)Rd, INFINITY) : INFINITY;
            float4 &A = dst[2 * (g >> 1)], &B = dst[2 * (g >> 1) + 1];
            if (g & 1) { A.y = c[0]; A.w = c[1]; B.y = c[2]; B.w = R; }
            else       { A.x = c[0]; A.z = c[1]; B.x = c[2]; B.z = R; }
        }
    };
    std::vector<std::array<double, 4>> gb(groups.size());
    for (size_t g = 0; g < groups.size(); g++) gb[g] = {groups[g].c[0], groups[g].c[1], groups[g].c[2], groups[g].R};
    pack_heads(gb, out.heads);

    // the box around the grouped spheres, rounded outwards; any non-finite member: no box
    {
        double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
        bool finite = !groups.empty();
        for (const Group &g : groups)
            for (uint32_t i : g.members) {
                const double cc[3] = {geom[i].x, geom[i].y, geom[i].z}, r = std::fabs((double)radius[i]);
                for (int a = 0; a < 3; a++) {
                    finite = finite && std::isfinite(cc[a]) && std::isfinite(r);
                    blo[a] = std::min(blo[a], cc[a] - r); bhi[a] = std::max(bhi[a], cc[a] + r);
                }
            }
        for (int a = 0; a < 3 && finite; a++) {
            out.box_lo[a] = std::nextafterf((float)blo[a], -INFINITY);
            out.box_hi[a] = std::nextafterf((float)bhi[a], INFINITY);
        }
    }

    // super-groups: RT_SUPER_GROUPS consecutive groups (Morton neighbours), bounded the same way
    out.n_supers = (int)((groups.size() + RT_SUPER_GROUPS - 1) / RT_SUPER_GROUPS);
    std::vector<std::array<double, 4>> sb(out.n_supers);
    for (int sgi = 0; sgi < out.n_supers; sgi++) {
        const size_t g0 = (size_t)sgi * RT_SUPER_GROUPS, g1 = std::min(groups.size(), g0 + RT_SUPER_GROUPS);
        double c[3];
        const double R = bound_S(groups[g0].members, g0, g1, &groups, c);     // over the member spheres of all its groups
        sb[sgi] = {c[0], c[1], c[2], R};
    }
    pack_heads(sb, out.supers);
    // and one level up: RT_SUPER_GROUPS consecutive super-groups (the per-ray tests of large scenes start here)
    out.n_supers2 = (out.n_supers + RT_SUPER_GROUPS - 1) / RT_SUPER_GROUPS;
    std::vector<std::array<double, 4>> s2(out.n_supers2);
    for (int i = 0; i < out.n_supers2; i++) {
        const size_t g0 = (size_t)i * RT_SUPER_GROUPS * RT_SUPER_GROUPS, g1 = std::min(groups.size(), g0 + (size_t)RT_SUPER_GROUPS * RT_SUPER_GROUPS);
        double c[3];
        const double R = bound_S(groups[g0].members, g0, g1, &groups, c);
        s2[i] = {c[0], c[1], c[2], R};
    }
    pack_heads(s2, out.supers2);
}

int rt_upload_scene(rt_context *ctx, const rt_object *objs, int32_t n, const rt_camera *cam)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (!objs || n <= 0 || !cam) return fail(ctx, RT_ERR_INVALID_ARG, "rt_upload_scene: objs/cam NULL or n <= 0");
    if (n > (1 << 22)) return fail(ctx, RT_ERR_LIMIT, "rt_upload_scene: %d objects > limit %d", n, 1 << 22);
    ctx->scene_version++;
    for (rt_context *part : ctx->parts) {              // a context over several devices: the scene is replicated
        int rc = rt_upload_scene(part, objs, n, cam);
        if (rc) return fail(ctx, rc, "device %d: %s", part->device, part->err);
    }
    if (!ctx->parts.empty()) {
        ctx->cam = *cam;
        ctx->have_scene = true;
        return RT_OK;
    }

    std::vector<float4> geom;
    std::vector<float> radius;
    std::vector<int32_t> sph_id;
    std::vector<RtPlaneDev> planes;
    std::vector<rt_material> mats((size_t)2 * n);
    std::vector<RtEmitterDev> emit;
    for (int32_t k = 0; k < n; k++) {
        const rt_object &o = objs[k];
        mats[2 * (size_t)k] = o.mat;
        mats[2 * (size_t)k + 1] = o.mat2;
        bool emits;
        int32_t slot = -1;
        if (o.type == RT_OBJ_SPHERE) {
            slot = (int32_t)geom.size();
            geom.push_back(make_float4(o.a[0], o.a[1], o.a[2], o.radius * o.radius));  // sceneobjects.h:71
            radius.push_back(o.radius);
            sph_id.push_back(k);
            mats[2 * (size_t)k + 1] = o.mat;
            emits = emissive_nonzero(o.mat);
        } else if (o.type == RT_OBJ_TILED_PLANE) {
            RtPlaneDev p = {o.a[0], o.a[1], o.a[2], o.b[0], o.b[1], o.b[2], k, 0};
            planes.push_back(p);
            emits = emissive_nonzero(o.mat) || emissive_nonzero(o.mat2);
        } else {
            return fail(ctx, RT_ERR_INVALID_ARG, "rt_upload_scene: object %d has unknown type %d", k, o.type);
        }
        if (emits) {
            RtEmitterDev e = {o.a[0], o.a[1], o.a[2], k, o.type == RT_OBJ_TILED_PLANE ? 1 : 0, slot, 0,
                              o.type == RT_OBJ_TILED_PLANE ? INFINITY : fabsf(o.radius)};   // order: below
            emit.push_back(e);
        }
    }
    // Sweep order of the shadow rays: at most 5 rays share a sweep, and rays towards emitters that lie
    // in similar directions touch the same groups of spheres, so emitters are ordered along the axis
    // their positions vary most on (then the next).  Which emitters share a sweep changes no result.
    {
        const size_t ne = emit.size();
        double mean[3] = {0, 0, 0}, var[3] = {0, 0, 0};
        for (const RtEmitterDev &e : emit) { mean[0] += e.cx; mean[1] += e.cy; mean[2] += e.cz; }
        for (int a = 0; a < 3; a++) mean[a] /= ne ? (double)ne : 1.0;
        for (const RtEmitterDev &e : emit) {
            const double c[3] = {e.cx, e.cy, e.cz};
            for (int a = 0; a < 3; a++) var[a] += (c[a] - mean[a]) * (c[a] - mean[a]);
        }
        int ax[3] = {0, 1, 2};
        std::sort(ax, ax + 3, [&](int a, int b) { return var[a] > var[b]; });
        std::vector<int32_t> ord(ne);
        for (size_t i = 0; i < ne; i++) ord[i] = (int32_t)i;
        auto coord = [&](int32_t i, int a) { return a == 0 ? emit[i].cx : a == 1 ? emit[i].cy : emit[i].cz; };
        const double split = mean[ax[0]];
        std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) {
            const bool sa = coord(a, ax[0]) < split, sb = coord(b, ax[0]) < split;     // two halves along the main axis ...
            if (sa != sb) return sa;
            return coord(a, ax[1]) < coord(b, ax[1]);                                  // ... each ordered along the second
        });
        for (size_t i = 0; i < ne; i++) emit[i].order = ord[i];
    }
    std::vector<RtEmitterDev> emit_sweep(emit.size());
    for (size_t i = 0; i < emit.size(); i++) {
        emit_sweep[i] = emit[(size_t)emit[i].order];
        emit_sweep[i].order = emit[i].order;
    }
    if (planes.size() > RT_MAX_PLANES)
        return fail(ctx, RT_ERR_LIMIT, "rt_upload_scene: %zu planes > limit %d", planes.size(), RT_MAX_PLANES);
    if (emit.size() > RT_MAX_EMITTERS)
        return fail(ctx, RT_ERR_LIMIT, "rt_upload_scene: %zu emissive objects > limit %d", emit.size(), RT_MAX_EMITTERS);

    std::vector<float4> pairs;
    for (size_t i = 0; i < geom.size(); i += 2) {
        const float4 g0 = geom[i];
        // odd tail: a sphere at the origin with -r*r = +FLT_MAX, which no ray can hit
        const float4 g1 = i + 1 < geom.size() ? geom[i + 1] : make_float4(0.0f, 0.0f, 0.0f, -3.402823466e+38f);
        pairs.push_back(make_float4(g0.x, g1.x, g0.y, g1.y));
        pairs.push_back(make_float4(g0.z, g1.z, -g0.w, -g1.w));
    }

    GroupedScene grp;
    std::vector<char> emissive(geom.size(), 0);
    for (const RtEmitterDev &e : emit)
        if (!e.is_plane) emissive[(size_t)e.sphere_slot] = 1;
    build_groups(geom, radius, sph_id, emissive, grp);

    auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off_pairs = 0;
    size_t off_geom = align(off_pairs + pairs.size() * sizeof(float4));
    size_t off_id = align(off_geom + geom.size() * sizeof(float4));
    size_t off_planes = align(off_id + sph_id.size() * sizeof(int32_t));
    size_t off_mats = align(off_planes + planes.size() * sizeof(RtPlaneDev));
    size_t off_emit = align(off_mats + mats.size() * sizeof(rt_material));
    size_t off_emit_sweep = align(off_emit + emit.size() * sizeof(RtEmitterDev));
    size_t off_gpairs = align(off_emit_sweep + emit_sweep.size() * sizeof(RtEmitterDev));
    size_t off_gids = align(off_gpairs + grp.pairs.size() * sizeof(float4));
    size_t off_heads = align(off_gids + grp.ids.size() * sizeof(int32_t));
    size_t off_supers = align(off_heads + grp.heads.size() * sizeof(float4));
    // the same bounds one per float4 {g, R} (beam test: one bound per lane)
    std::vector<float4> heads_flat((size_t)grp.n_groups), supers_flat((size_t)grp.n_supers);
    auto unpair = [](const std::vector<float4> &rec, std::vector<float4> &dst) {
        for (size_t g = 0; g < dst.size(); g++) {
            const float4 A = rec[2 * (g >> 1)], B = rec[2 * (g >> 1) + 1];
            dst[g] = (g & 1) ? make_float4(A.y, A.w, B.y, B.w) : make_float4(A.x, A.z, B.x, B.z);
        }
    };
    unpair(grp.heads, heads_flat);
    unpair(grp.supers, supers_flat);
    size_t off_supers2 = align(off_supers + grp.supers.size() * sizeof(float4));
    size_t off_heads_flat = align(off_supers2 + grp.supers2.size() * sizeof(float4));
    size_t off_supers_flat = align(off_heads_flat + heads_flat.size() * sizeof(float4));
    size_t total = align(off_supers_flat + supers_flat.size() * sizeof(float4)) + 256;

    std::vector<uint8_t> blob(total, 0);
    if (!pairs.empty()) memcpy(blob.data() + off_pairs, pairs.data(), pairs.size() * sizeof(float4));
    if (!geom.empty()) memcpy(blob.data() + off_geom, geom.data(), geom.size() * sizeof(float4));
    if (!sph_id.empty()) memcpy(blob.data() + off_id, sph_id.data(), sph_id.size() * sizeof(int32_t));
    if (!planes.empty()) memcpy(blob.data() + off_planes, planes.data(), planes.size() * sizeof(RtPlaneDev));
    memcpy(blob.data() + off_mats, mats.data(), mats.size() * sizeof(rt_material));
    if (!emit.empty()) memcpy(blob.data() + off_emit, emit.data(), emit.size() * sizeof(RtEmitterDev));
    if (!emit_sweep.empty()) memcpy(blob.data() + off_emit_sweep, emit_sweep.data(), emit_sweep.size() * sizeof(RtEmitterDev));
    if (!grp.pairs.empty()) memcpy(blob.data() + off_gpairs, grp.pairs.data(), grp.pairs.size() * sizeof(float4));
    if (!grp.ids.empty()) memcpy(blob.data() + off_gids, grp.ids.data(), grp.ids.size() * sizeof(int32_t));
    if (!grp.heads.empty()) memcpy(blob.data() + off_heads, grp.heads.data(), grp.heads.size() * sizeof(float4));
    if (!grp.supers.empty()) memcpy(blob.data() + off_supers, grp.supers.data(), grp.supers.size() * sizeof(float4));
    if (!grp.supers2.empty()) memcpy(blob.data() + off_supers2, grp.supers2.data(), grp.supers2.size() * sizeof(float4));
    if (!heads_flat.empty()) memcpy(blob.data() + off_heads_flat, heads_flat.data(), heads_flat.size() * sizeof(float4));
    if (!supers_flat.empty()) memcpy(blob.data() + off_supers_flat, supers_flat.data(), supers_flat.size() * sizeof(float4));

    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));     // a render may still read the old scene
    if (total > ctx->scene_blob_bytes) {
        if (ctx->scene_blob) RT_HIP(ctx, hipFree(ctx->scene_blob));
        ctx->scene_blob = nullptr;
        ctx->scene_blob_bytes = 0;
        RT_HIP(ctx, hipMalloc(&ctx->scene_blob, total));
        ctx->scene_blob_bytes = total;
    }
    RT_HIP(ctx, hipMemcpyAsync(ctx->scene_blob, blob.data(), total, hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));     // blob is a stack-lifetime host buffer

    uint8_t *base = (uint8_t *)ctx->scene_blob;
    ctx->scene.pairs = (const float4 *)(base + off_pairs);
    ctx->scene.geom = (const float4 *)(base + off_geom);
    ctx->scene.sph_id = (const int32_t *)(base + off_id);
    ctx->scene.planes = (const RtPlaneDev *)(base + off_planes);
    ctx->scene.mats = (const rt_material *)(base + off_mats);
    ctx->scene.emit = (const RtEmitterDev *)(base + off_emit);
    ctx->scene.emit_sweep = (const RtEmitterDev *)(base + off_emit_sweep);
    ctx->scene.n_spheres = (int32_t)geom.size();
    ctx->scene.n_planes = (int32_t)planes.size();
    ctx->scene.n_objects = n;
    ctx->scene.n_emit = (int32_t)emit.size();
    ctx->scene.n_pairs = (int32_t)(pairs.size() / 2);
    ctx->scene.heads = ctx->scene.supers = nullptr;
    ctx->scene.n_groups = ctx->scene.n_supers = ctx->scene.n_tail_pairs = 0;
    ctx->scene.metric[0] = ctx->scene.metric[1] = ctx->scene.metric[2] = ctx->scene.slack_scale = 1.0f;
    ctx->scene.sigma[0] = ctx->scene.sigma[1] = ctx->scene.sigma[2] = 1.0f;
    ctx->scene.heads_flat = ctx->scene.supers_flat = ctx->scene.supers2 = nullptr;
    ctx->scene.n_supers2 = 0;
    ctx->scene.beam_max_rho = ctx->scene.beam_max_chord2 = -1.0f;
    ctx->scene.far_bound = INFINITY;
    for (int a = 0; a < 3; a++) { ctx->scene.box_lo[a] = -INFINITY; ctx->scene.box_hi[a] = INFINITY; }
    ctx->scene_grouped = ctx->scene;
    ctx->scene_grouped.pairs = (const float4 *)(base + off_gpairs);
    ctx->scene_grouped.sph_id = (const int32_t *)(base + off_gids);
    ctx->scene_grouped.heads = (const float4 *)(base + off_heads);
    ctx->scene_grouped.n_pairs = (int32_t)(grp.pairs.size() / 2);
    ctx->scene_grouped.supers = (const float4 *)(base + off_supers);
    ctx->scene_grouped.n_groups = grp.n_groups;
    ctx->scene_grouped.n_tail_pairs = grp.n_tail_pairs;
    ctx->scene_grouped.n_supers = grp.n_supers;
    ctx->scene_grouped.supers2 = (const float4 *)(base + off_supers2);
    ctx->scene_grouped.n_supers2 = grp.n_supers2;
    ctx->scene_grouped.heads_flat = (const float4 *)(base + off_heads_flat);
    ctx->scene_grouped.supers_flat = (const float4 *)(base + off_supers_flat);
    ctx->scene_grouped.beam_max_rho = 8.0f;       // (measured: grid-1024 flat from 0.5 to 32; the 16k scene 36.8 ms at 0.5, 36.1 at 2, 33.3 at 8, 32.5 at 32)
    ctx->scene_grouped.beam_max_chord2 = 0.0625f;
    {
        // far lanes (rt_sweep_kernel.inc, dense_sweep): the filter keeps a sphere whose centre lies within about
        // sqrt(r^2 + 1.25 * 2^-17 |pos|^2) of a ray's line; once that reaches half (large scenes: a quarter) of the scene's radius the
        // lane's rays are tested against everything instead.  (RT_AMD_FAR_FRACTION: A/B; <= 0 turns it off.)
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (size_t i = 0; i < geom.size(); i++) {
            const double c[3] = {geom[i].x, geom[i].y, geom[i].z}, rr = std::fabs((double)radius[i]);
            for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], c[a] - rr); hi[a] = std::max(hi[a], c[a] + rr); }
        }
        double diag2 = 0.0;
        for (int a = 0; a < 3; a++) diag2 += (hi[a] - lo[a]) * (hi[a] - lo[a]);
        // RT_SWEEP_SORTED: 64 cells per axis over the box around the spheres (hit points outside it -- planes -- clamp to its faces)
        double cell_scale = 1.0;
        if (const char *v = debug_env("RT_AMD_CELL_SCALE")) cell_scale = atof(v);
        {
            double ext[3];
            bool ok = !geom.empty();
            for (int a = 0; a < 3; a++) { ext[a] = (hi[a] - lo[a]) * cell_scale; ok = ok && ext[a] == ext[a] && ext[a] < 1e30; if (!(ext[a] > 1e-3)) ext[a] = 1e-3; }
            // the bits of a cell's number between the axes: the next bit goes to the axis whose cells are the longest
            int bits[3] = {0, 0, 0};
            for (int k = 0; k < RT_SORTED_CELL_BITS; k++) {
                int best = 0;
                for (int a = 1; a < 3; a++) if (ext[a] / (double)(1 << bits[a]) > ext[best] / (double)(1 << bits[best])) best = a;
                bits[best]++;
            }
            if (const char *v = debug_env("RT_AMD_CELL_BITS")) sscanf(v, "%d,%d,%d", &bits[0], &bits[1], &bits[2]);
            if (bits[0] < 0 || bits[1] < 0 || bits[2] < 0 || bits[0] + bits[1] + bits[2] != RT_SORTED_CELL_BITS) { bits[0] = bits[1] = bits[2] = (RT_SORTED_CELL_BITS) / 3; }
            for (int a = 0; a < 3; a++) {
                const double mid = 0.5 * (hi[a] + lo[a]);
                ctx->cell_bits[a] = bits[a];
                ctx->cell_org[a] = ok ? (float)(mid - 0.5 * ext[a]) : 0.0f;
                ctx->cell_inv[a] = ok ? (float)((double)(1 << bits[a]) / ext[a]) : 0.0f;
            }
        }
        // (measured, headline / 16k scene at 4K: 0.25 6.00 / 24.0 ms, 0.5 5.88 / 24.9, 1 6.14 / 25.0, off 7.05 / 25.5; the 16k scene at
        // 1080p: 0.25 7.7 ms, 0.5 13.2 -- a wave that keeps a half-far lane walks most of the scene with all its lanes, which
        // takes the longer the larger the scene is, and at lower resolutions that one wave is the frame)
        double fraction = geom.size() <= 2048 ? 0.5 : 0.25;
        if (const char *v = debug_env("RT_AMD_FAR_FRACTION")) fraction = atof(v);
        const double reach = fraction * 0.5 * std::sqrt(diag2);        // of the radius of the box around the spheres
        const double fb = reach * reach / 5.0;                          // allowance^2 = 5 * (2^-19 |pos|^2) = 5 * filter bound
        ctx->scene_grouped.far_bound = (fraction > 0.0 && fb == fb && fb < 1e30 && !geom.empty()) ? (float)fb : INFINITY;
        // RT_SWEEP_SORTED gives every far segment a wave of its own, so a lane on the way to far no longer holds up 63 others
        // and the threshold can sit lower (measured on the 16k scene, 4K / 1080p: half the fraction 17.4 / 6.9 ms, the same
        // fraction 17.8 / 8.6, a quarter 21.2 / 7.9)
        ctx->far_bound_sorted = ctx->scene_grouped.far_bound * 0.25f;
    }
    for (int a = 0; a < 3; a++) { ctx->scene_grouped.box_lo[a] = grp.box_lo[a]; ctx->scene_grouped.box_hi[a] = grp.box_hi[a]; }
    for (int a = 0; a < 3; a++) ctx->scene_grouped.sigma[a] = grp.sigma[a];
    for (int a = 0; a < 3; a++) ctx->scene_grouped.metric[a] = grp.sigma[a] * grp.sigma[a];
    ctx->scene_grouped.slack_scale = grp.kappa * grp.kappa;
    ctx->cam = *cam;
    ctx->have_scene = true;
    return RT_OK;
}

int rt_plan_groups(const rt_object *objs, int32_t n, int32_t max_groups, int32_t *n_groups, int32_t *n_supers,
                   float *group_bounds, float *super_bounds, int32_t *group_of_object, float *sigma)
{
    if (!objs || n <= 0 || !n_groups || !n_supers) return fail(nullptr, RT_ERR_INVALID_ARG, "rt_plan_groups: NULL argument or n <= 0");
    std::vector<float4> geom;
    std::vector<float> radius;
    std::vector<int32_t> sph_id;
    std::vector<char> emissive;
    for (int32_t k = 0; k < n; k++) {
        if (group_of_object) group_of_object[k] = -1;
        if (objs[k].type != RT_OBJ_SPHERE) continue;
        geom.push_back(make_float4(objs[k].a[0], objs[k].a[1], objs[k].a[2], objs[k].radius * objs[k].radius));
        radius.push_back(objs[k].radius);
        sph_id.push_back(k);
        emissive.push_back(emissive_nonzero(objs[k].mat) ? 1 : 0);
    }
    GroupedScene grp;
    build_groups(geom, radius, sph_id, emissive, grp);
    *n_groups = grp.n_groups;
    *n_supers = grp.n_supers;
    for (int a = 0; sigma && a < 3; a++) sigma[a] = grp.sigma[a];
    if (grp.n_groups > max_groups)
        return fail(nullptr, RT_ERR_BUFFER_TOO_SMALL, "rt_plan_groups: %d groups, room for %d", grp.n_groups, max_groups);
    auto unpack = [](const std::vector<float4> &rec, int count, float *dst) {
        for (int g = 0; dst && g < count; g++) {
            const float4 A = rec[2 * (g >> 1)], B = rec[2 * (g >> 1) + 1];
            dst[4 * g + 0] = (g & 1) ? A.y : A.x;
            dst[4 * g + 1] = (g & 1) ? A.w : A.z;
            dst[4 * g + 2] = (g & 1) ? B.y : B.x;
            dst[4 * g + 3] = (g & 1) ? B.w : B.z;
        }
    };
    unpack(grp.heads, grp.n_groups, group_bounds);
    unpack(grp.supers, grp.n_supers, super_bounds);
    for (size_t i = 0; group_of_object && i < sph_id.size(); i++) {
        const int g = grp.slot_of[i] / (2 * RT_GROUP_PAIRS);
        group_of_object[sph_id[i]] = g < grp.n_groups ? g : -1;        // (the tail -- emissive spheres under no bound -- has no group)
    }
    return RT_OK;
}

int rt_set_camera(rt_context *ctx, const rt_camera *cam)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (!cam) return fail(ctx, RT_ERR_INVALID_ARG, "rt_set_camera: cam NULL");
    if (!ctx->have_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_set_camera: no scene uploaded");
    ctx->cam = *cam;          // passed by value with every launch: nothing on the device changes
    ctx->scene_version++;
    for (rt_context *part : ctx->parts) part->cam = *cam;
    return RT_OK;
}

static int ensure_plane(rt_context *ctx, int plane, size_t bytes)
{
    if (ctx->bound_plane[plane]) {
        if (ctx->bound_bytes[plane] < bytes)
            return fail(ctx, RT_ERR_BUFFER_TOO_SMALL, "bound framebuffer plane %d holds %zu bytes, render needs %zu",
                        plane, ctx->bound_bytes[plane], bytes);
        ctx->cur_plane[plane] = ctx->bound_plane[plane];
        return RT_OK;
    }
    if (ctx->own_bytes[plane] < bytes) {
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->own_plane[plane]) RT_HIP(ctx, hipFree(ctx->own_plane[plane]));
        ctx->own_plane[plane] = nullptr;
        ctx->own_bytes[plane] = 0;
        RT_HIP(ctx, hipMalloc(&ctx->own_plane[plane], bytes));
        ctx->own_bytes[plane] = bytes;
    }
    ctx->cur_plane[plane] = ctx->own_plane[plane];
    return RT_OK;
}

// RT_SWEEP_SORTED's buffers for a frame of `paths` paths and `levels` bounce levels (rt_sweep_sorted.hip): what they take ...
static size_t sorted_bytes_needed(size_t paths, int levels)
{
    const size_t cells = rt_sorted_cells();
    return paths * 64 * (size_t)levels + 4 * paths * 48 + paths * 4 + cells * 4 * (size_t)levels + 256;
}
// ... and the buffers themselves, grown when a render needs more (kept between renders)
static int ensure_sorted(rt_context *ctx, size_t paths, int levels, RtSortedDev *out)
{
    const size_t need = sorted_bytes_needed(paths, levels);
    if (!ctx->side_stream) {
        int lo = 0, hi = 0;
        RT_HIP(ctx, hipDeviceGetStreamPriorityRange(&lo, &hi));     // (lo: the numerically greatest = least urgent)
        // (the least urgent priority there is: it is filler work.  Measured: no different from the default priority.)
        RT_HIP(ctx, hipStreamCreateWithPriority(&ctx->side_stream, hipStreamNonBlocking, lo));
    }
    if (!ctx->ev_level0) RT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_level0, hipEventDisableTiming));
    if (!ctx->ev_side) RT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_side, hipEventDisableTiming));
    if (ctx->sorted_bytes < need) {
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->sorted_blob) RT_HIP(ctx, hipFree(ctx->sorted_blob));
        ctx->sorted_blob = nullptr;
        ctx->sorted_bytes = 0;
        RT_HIP(ctx, hipMalloc(&ctx->sorted_blob, need));
        ctx->sorted_bytes = need;
    }
    unsigned char *b = (unsigned char *)ctx->sorted_blob;
    const size_t cells = rt_sorted_cells();
    out->rec = (float4 *)b; b += paths * 64 * (size_t)levels;
    out->queue[0] = (float4 *)b; b += paths * 48;
    out->queue[1] = (float4 *)b; b += paths * 48;
    out->farq = (float4 *)b; b += paths * 48;
    out->farq0 = (float4 *)b; b += paths * 48;
    out->perm = (unsigned *)b; b += paths * 4;
    out->hist = (unsigned *)b; b += cells * 4 * (size_t)levels;
    out->count = (unsigned *)b;
    out->count_far = out->count + (RT_MAX_LEVELS + 1);
    out->count_kept = out->count + 2 * (RT_MAX_LEVELS + 1);
    out->cap = (unsigned)paths;
    // (measured on the 16k-sphere scene, levels sorted by themselves up to 1 / 2 / 3 / 5 of MAX_DEPTH 5: 4K 20.7 / 18.5 / 17.6 / 17.4 ms,
    // 1080p 6.55 / 6.26 / 6.25 / 6.41 ms: deep levels of a small frame hold too few segments to be worth a launch of their own)
    out->final_level = paths >= ((size_t)1 << 22) ? RT_MAX_LEVELS : 3;
    if (const char *v = debug_env("RT_AMD_SORTED_FINAL")) out->final_level = atoi(v);
    out->far0_waves = 0;
    if (const char *v = debug_env("RT_AMD_SORTED_FAR0_WAVES")) out->far0_waves = atoi(v);
    for (int a = 0; a < 3; a++) { out->cell_org[a] = ctx->cell_org[a]; out->cell_inv[a] = ctx->cell_inv[a]; out->cell_bits[a] = ctx->cell_bits[a]; }
    return RT_OK;
}

// Caller-owned output planes of rt_render_noalloc; NULL = the allocating path of rt_render_async.
struct NoAllocPlanes {
    void *plane[2];       // [0] rgba8, [1] f32x4
    size_t bytes[2];
};

// ---- one frame over several devices (rt_create_multi) --------------------------------------------------
// The reference parallelises INSIDE DrawScene: its rows are dealt to `partitions` threads that fill one shared
// Bitmap (main.cpp:193-227).  Here the rows [y0, y1) are cut into one contiguous band per device, balanced by a
// cost map from a 1/16-resolution pre-pass on the first device (per pixel: the tests its rays executed, exact
// tests weighted 32x, 300 per ray -- RT_OUT_COST); every device renders its band on its own stream into planes
// of its own, and the bands are copied into the frame on the first device (hipMemcpyPeerAsync over xGMI; a plain
// device copy where a band's device is the frame's).  The root context's stream waits for every copy, so the
// download / pack / caller's work that follows on it sees the whole frame.
static int plan_bands_multi(rt_context *ctx, const rt_render_params *p, int y0, int y1)
{
    const int n = (int)ctx->parts.size();
    rt_context *first = ctx->parts[0];
    const int lw = (p->width + 3) / 4, lh = (p->height + 3) / 4;
    rt_render_params q = {};
    q.width = lw;
    q.height = lh;
    q.max_depth = p->max_depth;
    q.spp = 1;
    q.outputs = RT_OUT_F32X4 | RT_OUT_COST;
    q.collect_stats = 1;
    q.mode = p->mode;
    q.sweep = p->sweep;
    const rt_camera full = first->cam;
    rt_camera low = full;                                // same pose and lens, smaller view plane (camera.h:24-25,41)
    low.image_width = (float)lw;
    low.image_height = (float)lh;
    low.aspect_ratio = (float)lw / (float)lh;
    first->cam = low;
    void *keep[2] = {first->bound_plane[0], first->bound_plane[1]};
    first->bound_plane[0] = first->bound_plane[1] = nullptr;
    (void)drain_events(first, true);                     // the pre-pass is not one of device 0's frames: its kernel time
    const double keep_total = first->kernel_ms_total, keep_last = first->kernel_ms_last;     // stays out of the totals
    const uint64_t keep_timed = first->kernels_timed;
    int rc = rt_render(first, &q);
    std::vector<float> plane((size_t)lw * (size_t)lh * 4);
    if (rc == RT_OK) rc = rt_download_framebuffer(first, plane.data(), plane.size() * sizeof(float), RT_FMT_F32X4);
    (void)drain_events(first, true);
    first->kernel_ms_total = keep_total;
    first->kernel_ms_last = keep_last;
    first->kernels_timed = keep_timed;
    ctx->replans++;
    first->cam = full;
    first->bound_plane[0] = keep[0];
    first->bound_plane[1] = keep[1];
    if (rc) return fail(ctx, rc, "pre-pass on device %d: %s", first->device, first->err);
    std::vector<double> low_cost((size_t)lh, 0.0);
    for (int r = 0; r < lh; r++)
        for (int c = 0; c < lw; c++) low_cost[(size_t)r] += (double)plane[((size_t)r * (size_t)lw + (size_t)c) * 4 + 3];
    std::vector<float> cost((size_t)(y1 - y0));
    for (int y = y0; y < y1; y++) {
        const long long r = std::min<long long>(((long long)y * lh) / p->height, lh - 1);
        const double c = low_cost[(size_t)r];
        cost[(size_t)(y - y0)] = (c >= 0.0 && c < 3.0e38) ? (float)c : 0.0f;
    }
    std::vector<int32_t> b((size_t)n + 1);
    rc = rt_split_bands(cost.data(), y1 - y0, n, b.data());
    if (rc) return fail(ctx, rc, "rt_split_bands failed on the pre-pass cost map");
    ctx->bounds.resize((size_t)n + 1);
    for (int i = 0; i <= n; i++) ctx->bounds[(size_t)i] = y0 + b[(size_t)i];
    return RT_OK;
}

static int render_impl(rt_context *ctx, const rt_render_params *p, const struct NoAllocPlanes *na);

// Queues the frame's parts on their devices; a failure after the first part was queued is handled by the caller
// (render_multi waits for what was launched).
static int render_multi_parts(rt_context *ctx, const rt_render_params *p, int y0, int y1, bool strips)
{
    const int n = (int)ctx->parts.size();
    RT_HIP(ctx, hipEventRecord(ctx->root_ready, ctx->stream));
    for (int i = 0; i < n; i++) {
        rt_context *part = ctx->parts[(size_t)i];
        const int b0 = strips ? y0 : ctx->bounds[(size_t)i], b1 = strips ? y1 : ctx->bounds[(size_t)i + 1];
        const int w_part = strips ? rt_interleaved_count(p->width, n, i) : p->width;
        RT_HIP(ctx, hipSetDevice(part->device));
        RT_HIP(ctx, hipStreamWaitEvent(part->stream, ctx->root_ready, 0));
        RT_HIP(ctx, hipEventRecord(ctx->copy_begin[(size_t)i], part->stream));     // (an empty part: begin == done)
        if (b1 > b0 && w_part > 0) {
            rt_render_params q = *p;
            q.row_begin = b0;
            q.row_end = b1;
            q.first_row = (p->first_row > b0 && p->first_row < b1) ? p->first_row : 0;
            q.col_interleave = strips ? n : 0;
            q.col_phase = strips ? i : 0;
            int rc = render_impl(part, &q, nullptr);
            if (rc) return fail(ctx, rc, "device %d, rows [%d,%d): %s", part->device, b0, b1, part->err);
            RT_HIP(ctx, hipEventRecord(ctx->copy_begin[(size_t)i], part->stream));
            for (int k = 0; k < 2; k++) {
                if (!ctx->cur_plane[k]) continue;
                const size_t bpp = k ? 16 : 4, bytes = (size_t)w_part * (size_t)(b1 - b0) * bpp;
                if (!strips) {
                    char *dst = (char *)ctx->cur_plane[k] + (size_t)p->width * (size_t)(b0 - y0) * bpp;
                    if (part->device == ctx->device)
                        RT_HIP(ctx, hipMemcpyAsync(dst, part->cur_plane[k], bytes, hipMemcpyDeviceToDevice, part->stream));
                    else
                        RT_HIP(ctx, hipMemcpyPeerAsync(dst, ctx->device, part->cur_plane[k], part->device, bytes, part->stream));
                } else if (part->device == ctx->device) {
                    // same device: the strips go straight to their columns (parts write disjoint columns)
                    RT_HIP(ctx, launch_scatter(ctx->cur_plane[k], part->cur_plane[k], k, p->width, w_part, b1 - b0, n, i, part->stream));
                } else {
                    // another device: one contiguous peer copy into the root device's staging copy of the part; the root's
                    // stream puts the strips in place once the copy is done (below)
                    RT_HIP(ctx, hipMemcpyPeerAsync(ctx->stage[k][(size_t)i], ctx->device, part->cur_plane[k], part->device, bytes, part->stream));
                }
            }
        }
        RT_HIP(ctx, hipEventRecord(ctx->part_done[(size_t)i], part->stream));
    }
    RT_HIP(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < n; i++) {
        RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->part_done[(size_t)i], 0));
        rt_context *part = ctx->parts[(size_t)i];
        if (!strips || part->device == ctx->device) continue;
        const int w_part = rt_interleaved_count(p->width, n, i);
        for (int k = 0; k < 2; k++)
            if (ctx->cur_plane[k] && w_part > 0 && y1 > y0)
                RT_HIP(ctx, launch_scatter(ctx->cur_plane[k], ctx->stage[k][(size_t)i], k, p->width, w_part, y1 - y0, n, i, ctx->stream));
    }
    return RT_OK;
}

static int render_multi(rt_context *ctx, const rt_render_params *p)
{
    ctx->have_frame = false;          // until this render is queued: a failure below leaves no frame, not the previous one's description
    if (!p) return fail(ctx, RT_ERR_INVALID_ARG, "rt_render: params NULL");
    if (!ctx->have_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_render: no scene uploaded");
    if (p->width <= 0 || p->height <= 0 || p->width > 65536 || p->height > 65536)
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_render: bad image size %dx%d", p->width, p->height);
    if (p->col_interleave > 1)
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_render: a context over several devices shares the frame itself (device_plan); col_interleave is for one context per device");
    if (p->device_plan != RT_PLAN_AUTO && p->device_plan != RT_PLAN_BANDS && p->device_plan != RT_PLAN_STRIPS)
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_render: unknown device_plan %d", p->device_plan);
    int y0 = p->row_begin, y1 = p->row_end;
    if (y0 == 0 && y1 == 0) y1 = p->height;
    if (y0 < 0 || y1 > p->height || y0 > y1)
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_render: bad row range [%d,%d) for height %d", y0, y1, p->height);
    const int n = (int)ctx->parts.size();
    int outputs = p->outputs ? p->outputs : RT_OUT_RGBA8;

    // 1. the plan.  Strips need none (and nothing blocks).  Bands: kept while scene, camera and frame geometry stay,
    //    and for the whole of a progressive accumulation, whose running mean lives in each device's part (a running
    //    mean also keeps the KIND of plan its first frame had)
    const bool continuing = p->accumulate && p->accum_index > 0 && ctx->last_plan != 0;
    bool strips = p->device_plan == RT_PLAN_STRIPS || (p->device_plan == RT_PLAN_AUTO && p->width >= RT_STRIP_COLS * n);
    if (continuing) strips = ctx->last_plan == RT_PLAN_STRIPS;
    // (fewer strips than devices: the devices past the last strip render nothing)
    if (!strips) {
        const rt_context::PlanKey key = {p->width, p->height, p->max_depth, y0, y1, p->spp, p->sweep, ctx->scene_version};
        const bool same_bands = continuing && ctx->have_plan && ctx->plan_key.w == key.w && ctx->plan_key.h == key.h &&
                                ctx->plan_key.y0 == key.y0 && ctx->plan_key.y1 == key.y1;
        if (!same_bands && !(ctx->have_plan && key == ctx->plan_key)) {
            ctx->have_plan = false;
            int rc = plan_bands_multi(ctx, p, y0, y1);
            if (rc) return rc;
            ctx->plan_key = key;
            ctx->have_plan = true;
        }
    }

    // 2. the frame on the first device (and, for strips rendered on other devices, a staging copy of each part)
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const size_t px = (size_t)p->width * (size_t)(y1 - y0);
    ctx->cur_plane[0] = ctx->cur_plane[1] = nullptr;
    for (int k = 0; k < 2; k++) {
        if (!(outputs & (k ? RT_OUT_F32X4 : RT_OUT_RGBA8))) continue;
        const size_t need = px * (k ? 16 : 4);
        int rc = ensure_plane(ctx, k, need ? need : 16);
        if (rc) return rc;
        if (!strips) continue;
        ctx->stage[k].resize((size_t)n, nullptr);
        ctx->stage_bytes[k].resize((size_t)n, 0);
        for (int i = 0; i < n; i++) {
            if (ctx->parts[(size_t)i]->device == ctx->device) continue;
            const size_t bytes = (size_t)rt_interleaved_count(p->width, n, i) * (size_t)(y1 - y0) * (k ? 16 : 4);
            if (ctx->stage_bytes[k][(size_t)i] >= bytes) continue;
            RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->stage[k][(size_t)i]) RT_HIP(ctx, hipFree(ctx->stage[k][(size_t)i]));
            ctx->stage[k][(size_t)i] = nullptr;
            ctx->stage_bytes[k][(size_t)i] = 0;
            RT_HIP(ctx, hipMalloc(&ctx->stage[k][(size_t)i], bytes));
            ctx->stage_bytes[k][(size_t)i] = bytes;
        }
    }

    // 3. the parts: each device renders on its own stream, after whatever is already on the root's
    if (strips) {
        ctx->bounds.assign((size_t)n + 1, y1);          // (reported per device by rt_get_stats: every device renders all rows)
        ctx->bounds[0] = y0;
    }
    {
        int rc = render_multi_parts(ctx, p, y0, y1, strips);
        if (rc) {
            // nothing is left running behind a failed render: wait for the parts that were queued
            for (rt_context *part : ctx->parts)
                if (hipSetDevice(part->device) == hipSuccess) (void)hipStreamSynchronize(part->stream);
            (void)hipSetDevice(ctx->device);
            return rc;
        }
    }

    RtFrameDev fr = {};
    fr.rgba8 = (uint32_t *)ctx->cur_plane[0];
    fr.f32x4 = (float4 *)ctx->cur_plane[1];
    fr.width = p->width;
    fr.height = p->height;
    fr.row_begin = y0;
    fr.row_end = y1;
    fr.spp = p->spp;
    ctx->frame = fr;
    ctx->have_frame = true;
    ctx->last_plan = strips ? RT_PLAN_STRIPS : RT_PLAN_BANDS;
    ctx->last_outputs = outputs & (RT_OUT_RGBA8 | RT_OUT_F32X4);
    ctx->stats_pending = p->collect_stats != 0;
    ctx->gather_pending = true;
    return RT_OK;
}

static int render_impl(rt_context *ctx, const rt_render_params *p, const NoAllocPlanes *na)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (!ctx->parts.empty()) {
        if (na) return fail(ctx, RT_ERR_INVALID_ARG, "rt_render_noalloc: not on a context over several devices (its frame assembly allocates)");
        return render_multi(ctx, p);
    }
    if (!p) return fail(ctx, RT_ERR_INVALID_ARG, "rt_render: params NULL");
    if (!ctx->have_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_render: no scene uploaded");
    if (p->width <= 0 || p->height <= 0 || p->width > 65536 || p->height > 65536)
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_render: bad image size %dx%d", p->width, p->height);
    if (p->max_depth < 0 || p->max_depth > RT_MAX_LEVELS - 1)
        return fail(ctx, RT_ERR_LIMIT, "rt_render: max_depth %d outside [0,%d]", p->max_depth, RT_MAX_LEVELS - 1);
    if (p->spp < 1 || p->spp > RT_MAX_SPP)
        return fail(ctx, RT_ERR_LIMIT, "rt_render: spp %d outside [1,%d]", p->spp, RT_MAX_SPP);
    if (!p->sample_offsets && p->spp != 1 && p->spp != 4)
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_render: spp %d needs sample_offsets (the reference pattern covers 1 and 4)", p->spp);
    int y0 = p->row_begin, y1 = p->row_end;
    if (y0 == 0 && y1 == 0) y1 = p->height;
    if (y0 < 0 || y1 > p->height || y0 > y1)
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_render: bad row range [%d,%d) for height %d", y0, y1, p->height);
    if (p->first_row < 0 || p->first_row >= p->height)
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_render: first_row %d outside the image", p->first_row);
    const bool strips = p->col_interleave > 1;
    if (p->col_interleave < 0 || (strips && (p->col_phase < 0 || p->col_phase >= p->col_interleave)))
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_render: bad col_interleave/col_phase %d/%d", p->col_interleave, p->col_phase);
    if (strips && ctx->simple_kernel)
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_render: the transcription kernel (RT_AMD_KERNEL=simple) has no col_interleave");
    const int out_width = strips ? rt_interleaved_count(p->width, p->col_interleave, p->col_phase) : p->width;
    if (p->mode != RT_MODE_EXACT && p->mode != RT_MODE_FAST)
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_render: unknown mode %d", p->mode);
    if (p->sweep != RT_SWEEP_AUTO && p->sweep != RT_SWEEP_FLAT && p->sweep != RT_SWEEP_GROUPED && p->sweep != RT_SWEEP_SORTED)
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_render: unknown sweep %d", p->sweep);
    int outputs = p->outputs ? p->outputs : RT_OUT_RGBA8;
    if (outputs & ~(RT_OUT_RGBA8 | RT_OUT_F32X4 | RT_OUT_COST))
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_render: unknown output flags 0x%x", outputs);
    if ((outputs & RT_OUT_COST) && (!(outputs & RT_OUT_F32X4) || !p->collect_stats || p->accumulate))
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_render: RT_OUT_COST needs RT_OUT_F32X4 and collect_stats, and excludes accumulate");
    if (p->accumulate && !(outputs & RT_OUT_F32X4))
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_render: accumulate needs the RT_OUT_F32X4 plane (it holds the running mean)");
    if (p->accumulate && p->accum_index < 0)
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_render: accum_index %d is negative", p->accum_index);
    if (ctx->simple_kernel && (p->lens_offsets || p->accumulate))
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_render: the transcription kernel (RT_AMD_KERNEL=simple) has no lens/accumulate options");

    RT_HIP(ctx, hipSetDevice(ctx->device));
    const size_t px = (size_t)out_width * (size_t)(y1 - y0);
    ctx->cur_plane[0] = ctx->cur_plane[1] = nullptr;
    ctx->have_frame = false;          // until this render is queued: a failure below leaves no frame, not the previous one's description
    for (int i = 0; i < 2; i++) {
        if (!(outputs & (i ? RT_OUT_F32X4 : RT_OUT_RGBA8))) continue;
        const size_t need = px * (i ? 16 : 4);
        if (na) {
            if (!na->plane[i] || ((uintptr_t)na->plane[i] & 15))
                return fail(ctx, RT_ERR_INVALID_ARG, "rt_render_noalloc: output plane %d is NULL or not 16-byte aligned", i);
            if (na->bytes[i] < need)
                return fail(ctx, RT_ERR_BUFFER_TOO_SMALL, "rt_render_noalloc: output plane %d holds %zu bytes, render needs %zu", i, na->bytes[i], need);
            ctx->cur_plane[i] = na->plane[i];
        } else {
            int rc = ensure_plane(ctx, i, need ? need : (i ? 16 : 4));
            if (rc) return rc;
        }
    }

    RtFrameDev fr = {};
    fr.rgba8 = (uint32_t *)ctx->cur_plane[0];
    fr.f32x4 = (float4 *)ctx->cur_plane[1];
    fr.width = out_width;                                    // of the output planes
    fr.height = p->height;
    fr.row_begin = y0;
    fr.row_end = y1;
    fr.strip_stride = strips ? p->col_interleave : 0;
    fr.strip_phase = strips ? p->col_phase : 0;
    fr.max_depth = p->max_depth;
    fr.spp = p->spp;
    fr.inv_spp = 1.0f / (float)p->spp;                       // main.cpp:210
    fr.cost_w = (outputs & RT_OUT_COST) != 0;
    fr.lens_on = p->lens_offsets != nullptr;
    fr.focal_length = p->focal_length;
    fr.accumulate = p->accumulate != 0;
    fr.k1 = (float)p->accum_index;                           // raytracer_win/main.cpp:292-293
    fr.k2 = 1.f / (fr.k1 + 1.f);
    bool grouped = false;
    // (scenes of a few dozen spheres gain nothing from grouping: they always take the flat sweep)
    if (p->sweep != RT_SWEEP_FLAT && !ctx->simple_kernel && ctx->scene.n_spheres >= 64) {
        grouped = true;
        fr.chunk = 0;         // nothing is staged in LDS: the grouped arrays are read through the scalar cache
    }
    if (grouped) {
    } else if (ctx->simple_kernel) {
        int cap = ctx->lds_budget / (int)sizeof(float4);
        fr.chunk = ctx->scene.n_spheres < cap ? ctx->scene.n_spheres : cap;
    } else {
        int cap = ctx->lds_budget / (4 * (int)sizeof(float4));    // geometry pair + camera-relative pair
        int n_pairs = (ctx->scene.n_spheres + 1) / 2;
        fr.chunk = n_pairs < cap ? n_pairs : cap;
    }

    // RT_SWEEP_SORTED: the bounce levels as separate launches (rt_sweep_sorted.hip); what it cannot do goes the one-launch way
    bool sorted = false;
    RtSortedDev sorted_dev = {};
    if (grouped && p->sweep == RT_SWEEP_SORTED) {
        if (na) return fail(ctx, RT_ERR_INVALID_ARG, "rt_render_noalloc: RT_SWEEP_SORTED keeps per-path buffers that the context allocates");
        if (outputs & RT_OUT_COST) return fail(ctx, RT_ERR_INVALID_ARG, "rt_render: RT_OUT_COST needs a one-launch sweep (RT_SWEEP_GROUPED)");
    }
    {
        // RT_SWEEP_AUTO takes it where it was measured to win: scenes large enough for the three-level bounds (paths scatter
        // over thousands of spheres after one bounce) at a quarter of a million paths or more (16k-sphere scene: 4K frame 20.1 ->
        // 17.4 ms, an eighth of it in strips 8.3 -> 4.6 ms, 1080p 6.4 -> 6.3 ms; the 1028-sphere scene loses: 4.7 -> 8.2 ms)
        const size_t paths = px * (size_t)p->spp;
        const bool fits = paths > 0 && paths < ((size_t)1 << 31) && sorted_bytes_needed(paths, p->max_depth + 1) <= RT_SORTED_MAX_BYTES;
        const bool wanted = p->sweep == RT_SWEEP_SORTED ||
                            (p->sweep == RT_SWEEP_AUTO && !na && !(outputs & RT_OUT_COST) && ctx->scene_grouped.n_supers > 64 && paths >= ((size_t)3 << 15));      // (measured on the 16k scene: 57 600 paths 3.25 vs 2.14 ms one-launch, 129 600 paths 2.67 vs 5.27)
        sorted = grouped && wanted && fits;
    }
    // One sample per lane where the kernel allows it (see the kernel's comment) ...
    fr.tile_w = fr.tile_h = 8;
    const RtSceneDev &use_scene = grouped ? ctx->scene_grouped : ctx->scene;
    if (ctx->sample_split && !ctx->simple_kernel && rt_sweep_can_put_samples_in_lanes(use_scene, fr, grouped)) {
        static const int kTile[7][2] = {{8, 8}, {8, 4}, {4, 4}, {4, 2}, {2, 2}, {2, 1}, {1, 1}};   // spp 1, 2, 4 ... 64
        int lg = 0;
        while ((1 << lg) < p->spp) lg++;
        fr.lane_samples = p->spp;
        fr.tile_w = kTile[lg][0];
        fr.tile_h = kTile[lg][1];
    }
    if (p->first_row > y0 && p->first_row < y1) fr.first_tile_row = (p->first_row - y0) / fr.tile_h;   // launch order only
    // ... else the sample split: spp waves per tile, their sums resolved in order afterwards.  Worth its buffer
    // traffic (32 B per sample and pixel) on scenes whose sweeps are long (rt_sweep_can_split_samples);
    // capped at 1 GiB.
    const size_t split_bytes = px * (size_t)p->spp * sizeof(float4);
    const bool want_split = ctx->sample_split && !fr.lane_samples && !ctx->simple_kernel && px > 0 && !sorted &&
                            rt_sweep_can_split_samples(use_scene, fr, grouped);
    if (sorted) {
        int rc = ensure_sorted(ctx, px * (size_t)p->spp, p->max_depth + 1, &sorted_dev);
        if (rc && p->sweep == RT_SWEEP_AUTO && !ctx->sorted_blob) {
            // no room for the buffers (a GPU shared with others): AUTO was a preference, the one-launch sweep needs none
            (void)hipGetLastError();
            sorted = false;
        } else if (rc) {
            return rc;
        }
    }
    if (na) {
        // the sample-split plane lives in this device's scratch array behind the counters; a render whose
        // plane does not fit is rejected (the allocating path would grow its buffer instead)
        const size_t room = (size_t)RT_NOALLOC_SCRATCH_BYTES - RT_NOALLOC_STATS_BYTES;
        if (want_split) {
            if (split_bytes > room)
                return fail(ctx, RT_ERR_LIMIT, "rt_render_noalloc: %zu pixels x %d samples need %zu bytes of scratch, the fixed array holds %zu "
                            "(render fewer rows per call, or use rt_render)", px, p->spp, split_bytes, room);
            fr.sample_buf = (float4 *)((unsigned char *)ctx->scratch_dev + RT_NOALLOC_STATS_BYTES);
        }
    } else if (want_split && split_bytes <= ((size_t)1 << 30)) {
        if (ctx->sample_bytes < split_bytes) {
            RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->sample_buf) RT_HIP(ctx, hipFree(ctx->sample_buf));
            ctx->sample_buf = nullptr;
            ctx->sample_bytes = 0;
            RT_HIP(ctx, hipMalloc(&ctx->sample_buf, split_bytes));
            ctx->sample_bytes = split_bytes;
        }
        fr.sample_buf = (float4 *)ctx->sample_buf;
    }

    RtSamples smp;
    memset(&smp, 0, sizeof smp);
    memcpy(smp.xy, p->sample_offsets ? p->sample_offsets : kRefPattern, sizeof(float) * 2 * (size_t)p->spp);
    if (p->lens_offsets) memcpy(smp.lens, p->lens_offsets, sizeof(float) * 2 * (size_t)p->spp);

    if (ctx->region_dev && !na) {
        fr.region_cycles = ctx->region_dev;
        RT_HIP(ctx, hipMemsetAsync(ctx->region_dev, 0, RT_N_REGIONS * sizeof(unsigned long long), ctx->stream));
    }
    RtStatsDev *sdev = nullptr;
    if (p->collect_stats) {
        // (the scratch array keeps the last call's counters: they start from zero every time)
        sdev = na ? (RtStatsDev *)ctx->scratch_dev : ctx->stats_dev;
        RT_HIP(ctx, hipMemsetAsync(sdev, 0, sizeof(RtStatsDev), ctx->stream));
    }
    if (na) {
        // no events here: creating one allocates, and nothing may; kernel_ms keeps its last value
        if (px > 0) {
            RT_HIP(ctx, rt_launch_trace_sweep(grouped ? ctx->scene_grouped : ctx->scene, ctx->cam, fr, smp, sdev,
                                              p->mode == RT_MODE_FAST, grouped, ctx->stream));
            if (fr.sample_buf) RT_HIP(ctx, rt_launch_resolve_samples(fr, ctx->stream));
        }
    } else {
    if (ctx->ev_pending.size() >= 256) {          // bound the pool: fold the oldest launches in
        int rc = drain_events(ctx, false);
        if (rc) return rc;
        if (ctx->ev_pending.size() >= 256) {
            RT_HIP(ctx, hipEventSynchronize(ctx->ev_pending.front().second));
            if ((rc = drain_events(ctx, false))) return rc;
        }
    }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // the pair goes back to the pool on every early return below (a failed launch must not leak events)
    struct EventPair {
        rt_context *c; hipEvent_t *a, *b; bool keep = false;
        ~EventPair() { if (!keep) { if (*a) c->ev_free.push_back(*a); if (*b) c->ev_free.push_back(*b); } }
    } pair_guard{ctx, &ev0, &ev1};
    { int rc = take_event(ctx, &ev0); if (rc) return rc; rc = take_event(ctx, &ev1); if (rc) return rc; }
    RT_HIP(ctx, hipEventRecord(ev0, ctx->stream));
    if (px > 0) {
        // RT_MODE_FAST: same sweeps, shading by throughput accumulation (no per-segment records).
        // The straight-transcription kernel has only the exact formulation.
        if (ctx->simple_kernel) RT_HIP(ctx, rt_launch_trace_exact(ctx->scene, ctx->cam, fr, smp, sdev, ctx->stream));
        else if (sorted) {
            RtSceneDev sc = ctx->scene_grouped;
            sc.far_bound = ctx->far_bound_sorted;
            RT_HIP(ctx, rt_launch_trace_sorted(sc, ctx->cam, fr, smp, sdev, p->mode == RT_MODE_FAST, sorted_dev, ctx->stream, ctx->side_stream,
                                               ctx->ev_level0, ctx->ev_side));
        }
        else RT_HIP(ctx, rt_launch_trace_sweep(grouped ? ctx->scene_grouped : ctx->scene, ctx->cam, fr, smp, sdev,
                                               p->mode == RT_MODE_FAST, grouped, ctx->stream));
        if (fr.sample_buf) RT_HIP(ctx, rt_launch_resolve_samples(fr, ctx->stream));
    }
    RT_HIP(ctx, hipEventRecord(ev1, ctx->stream));
    ctx->ev_pending.emplace_back(ev0, ev1);
    pair_guard.keep = true;
    }

    ctx->stats_src = sdev;
    ctx->frame = fr;
    ctx->have_frame = true;
    ctx->last_outputs = outputs & (RT_OUT_RGBA8 | RT_OUT_F32X4);
    ctx->stats_pending = p->collect_stats != 0;
    memset(&ctx->stats, 0, sizeof ctx->stats);
    ctx->stats.primary_rays = (uint64_t)px * (uint64_t)p->spp;
    ctx->stats.n_objects = (uint64_t)ctx->scene.n_objects;
    ctx->stats.n_emitters = (uint64_t)ctx->scene.n_emit;
    ctx->stats.sweep = sorted ? RT_SWEEP_SORTED : grouped ? RT_SWEEP_GROUPED : RT_SWEEP_FLAT;
    return RT_OK;
}

int rt_render_async(rt_context *ctx, const rt_render_params *p) { return render_impl(ctx, p, nullptr); }

int rt_render_noalloc(rt_context *ctx, const rt_render_params *p, void *rgba8_dev, size_t rgba8_bytes, void *f32x4_dev,
                      size_t f32x4_bytes)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (ctx->simple_kernel) return fail(ctx, RT_ERR_INVALID_ARG, "rt_render_noalloc: not with the transcription kernel (RT_AMD_KERNEL=simple)");
    const NoAllocPlanes na = {{rgba8_dev, f32x4_dev}, {rgba8_bytes, f32x4_bytes}};
    return render_impl(ctx, p, &na);
}

int rt_sync(rt_context *ctx)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    for (rt_context *part : ctx->parts) {
        int rc = rt_sync(part);
        if (rc) return fail(ctx, rc, "device %d: %s", part->device, part->err);
    }
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_render(rt_context *ctx, const rt_render_params *p)
{
    int rc = rt_render_async(ctx, p);
    if (rc) return rc;
    return rt_sync(ctx);
}

int rt_get_stream(rt_context *ctx, void **out_stream)
{
    if (!ctx || !out_stream) return RT_ERR_INVALID_ARG;
    *out_stream = (void *)ctx->stream;
    return RT_OK;
}

int rt_set_stream(rt_context *ctx, void *stream)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));      // nothing of ours is left on the old stream
    { int rc = drain_events(ctx, true); if (rc) return rc; }
    ctx->stream = stream ? (hipStream_t)stream : ctx->own_stream;
    return RT_OK;
}

int rt_bind_framebuffer(rt_context *ctx, int32_t plane, void *device_ptr, size_t bytes)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    int idx = plane == RT_OUT_RGBA8 ? 0 : plane == RT_OUT_F32X4 ? 1 : -1;
    if (idx < 0) return fail(ctx, RT_ERR_INVALID_ARG, "rt_bind_framebuffer: plane must be RT_OUT_RGBA8 or RT_OUT_F32X4");
    if (device_ptr && ((uintptr_t)device_ptr & 15))
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_bind_framebuffer: pointer must be 16-byte aligned");
    ctx->bound_plane[idx] = device_ptr;
    ctx->bound_bytes[idx] = device_ptr ? bytes : 0;
    return RT_OK;
}

int rt_framebuffer_device_ptr(rt_context *ctx, int32_t plane, void **out_ptr, size_t *out_bytes)
{
    if (!ctx || !out_ptr) return RT_ERR_INVALID_ARG;
    if (!ctx->have_frame) return fail(ctx, RT_ERR_NO_FRAME, "rt_framebuffer_device_ptr: nothing rendered yet");
    int idx = plane == RT_OUT_RGBA8 ? 0 : plane == RT_OUT_F32X4 ? 1 : -1;
    if (idx < 0 || !ctx->cur_plane[idx])
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_framebuffer_device_ptr: plane %d was not rendered", plane);
    *out_ptr = ctx->cur_plane[idx];
    if (out_bytes)
        *out_bytes = (size_t)ctx->frame.width * (size_t)(ctx->frame.row_end - ctx->frame.row_begin) * (idx ? 16 : 4);
    return RT_OK;
}

int rt_download_framebuffer(rt_context *ctx, void *dst, size_t dst_bytes, int32_t format)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (!dst) return fail(ctx, RT_ERR_INVALID_ARG, "rt_download_framebuffer: dst NULL");
    if (!ctx->have_frame) return fail(ctx, RT_ERR_NO_FRAME, "rt_download_framebuffer: nothing rendered yet");
    const RtFrameDev &fr = ctx->frame;
    const size_t w = (size_t)fr.width, rows = (size_t)(fr.row_end - fr.row_begin), px = w * rows;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (format == RT_FMT_F32X4) {
        if (!ctx->cur_plane[1]) return fail(ctx, RT_ERR_INVALID_ARG, "rt_download_framebuffer: F32X4 plane was not rendered");
        if (dst_bytes < px * 16) return fail(ctx, RT_ERR_BUFFER_TOO_SMALL, "rt_download_framebuffer: need %zu bytes, got %zu", px * 16, dst_bytes);
        RT_HIP(ctx, hipMemcpyAsync(dst, ctx->cur_plane[1], px * 16, hipMemcpyDeviceToHost, ctx->stream));
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return RT_OK;
    }
    if (!ctx->cur_plane[0]) return fail(ctx, RT_ERR_INVALID_ARG, "rt_download_framebuffer: RGBA8 plane was not rendered");
    if (format == RT_FMT_RGBA8) {
        if (dst_bytes < px * 4) return fail(ctx, RT_ERR_BUFFER_TOO_SMALL, "rt_download_framebuffer: need %zu bytes, got %zu", px * 4, dst_bytes);
        RT_HIP(ctx, hipMemcpyAsync(dst, ctx->cur_plane[0], px * 4, hipMemcpyDeviceToHost, ctx->stream));
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return RT_OK;
    }
    if (format != RT_FMT_RGB8 && format != RT_FMT_BMP_BODY)
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_download_framebuffer: unknown format %d", format);
    const size_t need = packed_bytes(fr, format);
    if (dst_bytes < need) return fail(ctx, RT_ERR_BUFFER_TOO_SMALL, "rt_download_framebuffer: need %zu bytes, got %zu", need, dst_bytes);
    if (need == 0) return RT_OK;
    // pack on the device (3 bytes per pixel leave the GPU instead of 4, and no host loop), then one copy
    const size_t staged = (need + 3) & ~(size_t)3;
    if (ctx->pack_bytes < staged) {
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->pack_buf) RT_HIP(ctx, hipFree(ctx->pack_buf));
        ctx->pack_buf = nullptr;
        ctx->pack_bytes = 0;
        RT_HIP(ctx, hipMalloc(&ctx->pack_buf, staged));
        ctx->pack_bytes = staged;
    }
    RT_HIP(ctx, launch_pack(fr, (const uint32_t *)ctx->cur_plane[0], (uint32_t *)ctx->pack_buf, format, ctx->stream));
    RT_HIP(ctx, hipMemcpyAsync(dst, ctx->pack_buf, need, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_pack_framebuffer(rt_context *ctx, int32_t format, void *device_dst, size_t dst_bytes)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (!device_dst || ((uintptr_t)device_dst & 3)) return fail(ctx, RT_ERR_INVALID_ARG, "rt_pack_framebuffer: dst NULL or not 4-byte aligned");
    if (!ctx->have_frame || !ctx->cur_plane[0]) return fail(ctx, RT_ERR_NO_FRAME, "rt_pack_framebuffer: no RGBA8 frame rendered");
    if (format != RT_FMT_RGB8 && format != RT_FMT_BMP_BODY)
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_pack_framebuffer: format must be RT_FMT_RGB8 or RT_FMT_BMP_BODY");
    const size_t need = (packed_bytes(ctx->frame, format) + 3) & ~(size_t)3;
    if (dst_bytes < need) return fail(ctx, RT_ERR_BUFFER_TOO_SMALL, "rt_pack_framebuffer: need %zu bytes, got %zu", need, dst_bytes);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, launch_pack(ctx->frame, (const uint32_t *)ctx->cur_plane[0], (uint32_t *)device_dst, format, ctx->stream));
    return RT_OK;
}

int rt_get_stats(rt_context *ctx, rt_stats *out)
{
    if (!ctx || !out) return RT_ERR_INVALID_ARG;
    if (!ctx->have_frame) return fail(ctx, RT_ERR_NO_FRAME, "rt_get_stats: nothing rendered yet");
    if (!ctx->parts.empty()) {
        // over several devices: counters add up, times are per device; the frame took as long as its slowest band
        int rc = rt_sync(ctx);
        if (rc) return rc;
        rt_stats total = {};
        total.n_devices = (uint32_t)ctx->parts.size();
        for (size_t i = 0; i < ctx->parts.size(); i++) {
            rt_context *part = ctx->parts[i];
            total.device_id[i] = part->device;
            const bool strips = ctx->last_plan == RT_PLAN_STRIPS;
            const int rb = strips ? ctx->bounds[0] : ctx->bounds[i], re = strips ? ctx->bounds.back() : ctx->bounds[i + 1];
            total.device_row_begin[i] = rb;
            total.device_row_end[i] = re;
            if (re == rb || (strips && rt_interleaved_count(ctx->frame.width, (int)ctx->parts.size(), (int)i) <= 0)) continue;   // an empty part rendered nothing
            rt_stats st = {};
            if ((rc = rt_get_stats(part, &st))) return fail(ctx, rc, "device %d: %s", part->device, part->err);
            total.primary_rays += st.primary_rays;
            total.reflection_rays += st.reflection_rays;
            total.shadow_rays += st.shadow_rays;
            total.hits += st.hits;
            total.lit += st.lit;
            total.sphere_tests += st.sphere_tests;
            total.plane_tests += st.plane_tests;
            total.sphere_tests_run += st.sphere_tests_run;
            total.group_tests_run += st.group_tests_run;
            total.exact_tests_run += st.exact_tests_run;
            total.n_objects = st.n_objects;
            total.n_emitters = st.n_emitters;
            total.sweep = st.sweep;
            total.device_kernel_ms[i] = st.kernel_ms;
            total.kernel_ms = std::max(total.kernel_ms, st.kernel_ms);
            total.kernel_ms_total = std::max(total.kernel_ms_total, st.kernel_ms_total);
            total.kernels_timed = std::max(total.kernels_timed, st.kernels_timed);
            if (ctx->gather_pending) {
                float ms = 0.0f;
                RT_HIP(ctx, hipSetDevice(part->device));
                RT_HIP(ctx, hipEventElapsedTime(&ms, ctx->copy_begin[i], ctx->part_done[i]));
                total.device_gather_ms[i] = (double)ms;
                total.gather_ms = std::max(total.gather_ms, (double)ms);
            }
        }
        if (!ctx->gather_pending) total.gather_ms = ctx->gather_ms_last;
        ctx->gather_ms_last = total.gather_ms;
        total.replans = ctx->replans;
        total.device_plan = ctx->last_plan;
        *out = total;
        return RT_OK;
    }
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    { int rc = drain_events(ctx, true); if (rc) return rc; }
    if (ctx->region_dev) {                            // profiling builds (-DRT_REGION_TIMERS): wave-cycles by kernel region
        unsigned long long h[RT_N_REGIONS];
        RT_HIP(ctx, hipMemcpy(h, ctx->region_dev, sizeof h, hipMemcpyDeviceToHost));
        fprintf(stderr, "rt region cycles:");
        for (int i = 0; i < RT_N_REGIONS; i++) fprintf(stderr, " %llu", h[i]);
        fprintf(stderr, "\n");
    }
    ctx->stats.kernel_ms = ctx->kernel_ms_last;
    ctx->stats.kernel_ms_total = ctx->kernel_ms_total;
    ctx->stats.kernels_timed = ctx->kernels_timed;
    if (ctx->stats_pending) {
        RtStatsDev h;
        RT_HIP(ctx, hipMemcpy(&h, ctx->stats_src ? ctx->stats_src : ctx->stats_dev, sizeof h, hipMemcpyDeviceToHost));
        ctx->stats.reflection_rays = h.reflection;
        ctx->stats.shadow_rays = h.shadow;
        ctx->stats.hits = h.hits;
        ctx->stats.lit = h.lit;
        uint64_t rays = ctx->stats.primary_rays + h.reflection + h.shadow;
        ctx->stats.sphere_tests = rays * (uint64_t)ctx->scene.n_spheres;
        ctx->stats.plane_tests = rays * (uint64_t)ctx->scene.n_planes;
        ctx->stats.sphere_tests_run = h.sphere_tests_run;
        ctx->stats.group_tests_run = h.group_tests_run;
        ctx->stats.exact_tests_run = h.exact_tests_run;
        ctx->stats_pending = false;

    }
    *out = ctx->stats;
    return RT_OK;
}

} // extern "C"
