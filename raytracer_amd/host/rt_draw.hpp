// rt_draw.hpp -- DrawScene with the reference's signature (raytracer_cpp/main.cpp:190), served by the
// MI355X library: the scene list and camera (the reference's globals, main.cpp:18-19) are flattened,
// uploaded, traced on the device and the frame lands in Bitmap::pData.  There is no CPU path: a
// failure of any rt_* call is reported through DrawSceneError and nothing is drawn.
#pragma once
#include <cmath>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "rt_api.h"
#include "rt_bitmap.hpp"
#include "rt_camera.hpp"
#include "rt_scene.hpp"

extern std::vector<std::shared_ptr<SceneObject>> sceneObjects;
extern std::shared_ptr<Camera> pCamera;

struct DrawSceneError : std::runtime_error {
    int status;
    DrawSceneError(int s, const std::string &what) : std::runtime_error(what), status(s) {}
};

struct DrawSceneOptions {
    int maxDepth = 5;           // MAX_DEPTH, main.cpp:16
    int device = 0;
    std::vector<int> devices;   // more than one entry: the frame is shared between these HIP devices (rt_create_multi),
                                // the counterpart of the reference's thread fan-out inside DrawScene (main.cpp:193-227)
    int mode = RT_MODE_EXACT;   // RT_MODE_FAST: radiance to rounding instead of bit-identical
    int sweep = RT_SWEEP_AUTO;  // RT_SWEEP_FLAT / RT_SWEEP_GROUPED / RT_SWEEP_SORTED: same image every way
    double *kernelMs = nullptr; // optional: device time of the trace kernel (summed over frames)
    // raytracer_win's progressive mode (raytracer_win/main.cpp:280-327, camera.h:147-173): `frames`
    // single-sample frames folded into a running mean, each with its own sub-pixel jitter and, when
    // lensRadius > 0, its own point on the rim of the lens (what glm::circularRand(radius) draws).
    // The reference seeds rand() from the clock; here the sequence is the scene generators' LCG so a
    // run can be reproduced.  frames = 1, lensRadius = 0 is raytracer_cpp's single frame.
    int frames = 1;
    float lensRadius = 0.0f;
    float focalLength = 0.0f;   // glm::length(focalPoint - position)
    uint32_t seed = 0x5EED1E25u;
};

namespace rt_detail {
inline float lcg_unit(uint32_t &s)
{
    s = s * 1664525u + 1013904223u;
    return (float)(s >> 8) * (1.0f / 16777216.0f);
}
} // namespace rt_detail

// `partitions` is the reference's CPU thread count; a device schedules its own waves, so the
// value only has to be positive.  The parallelism the reference has inside this call -- rows dealt to threads
// -- is DrawSceneOptions::devices here: rows dealt to GPUs.  `antialias` selects the reference's 4 fixed sub-pixel samples.
inline void DrawScene(Bitmap *pBitmap, int partitions, bool antialias, const DrawSceneOptions &opt = DrawSceneOptions())
{
    if (!pBitmap || partitions < 1 || !pCamera || sceneObjects.empty())
        throw DrawSceneError(RT_ERR_INVALID_ARG, "DrawScene: bitmap, camera and a non-empty scene are required");
    std::vector<rt_object> flat(sceneObjects.size());
    for (size_t i = 0; i < sceneObjects.size(); i++) sceneObjects[i]->Flatten(flat[i]);   // list order is semantic
    const rt_camera cam = pCamera->Flatten();

    rt_context *ctx = nullptr;
    int rc = opt.devices.size() > 1 ? rt_create_multi(&ctx, opt.devices.data(), (int)opt.devices.size())
                                    : rt_create(&ctx, opt.devices.size() == 1 ? opt.devices[0] : opt.device);
    if (rc != RT_OK) throw DrawSceneError(rc, rt_last_error(nullptr));
    auto fail = [&](int code) {
        const std::string msg = rt_last_error(ctx);
        rt_destroy(ctx);
        throw DrawSceneError(code, msg);
    };
    if ((rc = rt_upload_scene(ctx, flat.data(), (int32_t)flat.size(), &cam)) != RT_OK) fail(rc);
    rt_render_params p{};
    p.width = pBitmap->width;
    p.height = pBitmap->height;
    p.max_depth = opt.maxDepth;
    p.mode = opt.mode;
    p.sweep = opt.sweep;
    const bool progressive = opt.frames > 1 || opt.lensRadius > 0.0f;
    double kernelMs = 0.0;
    uint32_t s = opt.seed;
    for (int k = 0; k < (opt.frames > 0 ? opt.frames : 1); k++) {
        float jitter[2], lens[2];
        if (progressive) {
            p.spp = 1;
            jitter[0] = rt_detail::lcg_unit(s);
            jitter[1] = rt_detail::lcg_unit(s);
            p.sample_offsets = jitter;
            p.outputs = RT_OUT_RGBA8 | RT_OUT_F32X4;
            p.accumulate = 1;
            p.accum_index = k;
            if (opt.lensRadius > 0.0f) {
                const double a = 6.283185307179586 * (double)rt_detail::lcg_unit(s);
                lens[0] = (float)std::cos(a) * opt.lensRadius;
                lens[1] = (float)std::sin(a) * opt.lensRadius;
                p.lens_offsets = lens;
                p.focal_length = opt.focalLength;
            }
        } else {
            p.spp = antialias ? 4 : 1;             // main.cpp:200
            p.outputs = RT_OUT_RGBA8;
        }
        if ((rc = rt_render(ctx, &p)) != RT_OK) fail(rc);
        if (opt.kernelMs) {
            rt_stats st{};
            if ((rc = rt_get_stats(ctx, &st)) != RT_OK) fail(rc);
            kernelMs += st.kernel_ms;
        }
    }
    if (opt.kernelMs) *opt.kernelMs = kernelMs;
    const size_t bytes = (size_t)pBitmap->width * (size_t)pBitmap->height * sizeof(Color);
    if ((rc = rt_download_framebuffer(ctx, pBitmap->pData, bytes, RT_FMT_RGB8)) != RT_OK) fail(rc);
    rt_destroy(ctx);
}
