// ref_trace_harness.cpp -- TEST INFRASTRUCTURE.  extern "C" driver over the REFERENCE'S OWN trace loop.
//
// This file is never compiled alone.  oracle/Makefile streams raytracer_cpp/main.cpp, as it lies under
// /root/reference, through a line filter into the compiler's stdin and appends this file to the same
// translation unit, so everything below sees the reference's globals and functions verbatim:
//     sceneObjects, pCamera (main.cpp:18-19), InitScene (:21-63), FindNearestObject (:65-82),
//     TraceRay (:84-188), DrawScene (:190-228), ImageWidth/ImageHeight (:12-13).
// The filter drops exactly three things and the Makefile checks that it found each exactly once:
//     the `#include "cmdparser\cmdparser.hpp"` line (:10; un-vendored submodule, only main() uses it),
//     the `#define MAX_DEPTH 5` line (:16; replaced by -DMAX_DEPTH=<n>, one library per depth),
//     main() itself (:230-258).
// Nothing of the reference is written to disk or committed: the filtered text exists only in the pipe,
// and only the built oracle/_ref/libref_trace_d<n>.so stays (git-ignored).
//
// The scene is filled through the reference's own constructors (Sphere, TiledPlane, Camera); data the
// constructors do not take (TiledPlane's two materials, a Camera given by its eight fields) is stored
// into the objects' public members afterwards.  Counting subclasses override nothing but forward to
// the reference's virtuals, so the counters see exactly the calls TraceRay makes.
#include <atomic>
#include <cstdio>

#include "rt_api.h"

namespace refh {

static vec3 V(const float *p) { return vec3(p[0], p[1], p[2]); }
static void S(float *o, const vec3 &v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

static Material to_material(const rt_material &m)
{
    Material r;
    r.albedo = V(m.albedo);
    r.specular = V(m.specular);
    r.reflectance = m.reflectance;
    r.emissive = V(m.emissive);
    return r;
}

static void from_material(rt_material &d, const Material &m, bool emissive_defined)
{
    S(d.albedo, m.albedo);
    S(d.specular, m.specular);
    d.reflectance = m.reflectance;
    if (emissive_defined) S(d.emissive, m.emissive);
    else d.emissive[0] = d.emissive[1] = d.emissive[2] = 0.0f;
}

// calls the trace loop makes into the scene objects (relaxed atomics: DrawScene runs threads)
static std::atomic<unsigned long long> n_intersects{0}, n_normals{0}, n_ray_from{0}, n_materials{0};

struct CountingSphere : Sphere {
    CountingSphere(const Material &m, const vec3 &c, float r) : Sphere(m, c, r) {}
    const Material &GetMaterial(const vec3 &pos) const override { n_materials.fetch_add(1, std::memory_order_relaxed); return Sphere::GetMaterial(pos); }
    vec3 GetSurfaceNormal(const vec3 &pos) const override { n_normals.fetch_add(1, std::memory_order_relaxed); return Sphere::GetSurfaceNormal(pos); }
    vec3 GetRayFrom(const vec3 &from) const override { n_ray_from.fetch_add(1, std::memory_order_relaxed); return Sphere::GetRayFrom(from); }
    bool Intersects(const vec3 &o, const vec3 &d, float &dist) const override { n_intersects.fetch_add(1, std::memory_order_relaxed); return Sphere::Intersects(o, d, dist); }
};

struct CountingPlane : TiledPlane {
    CountingPlane(const vec3 &o, const vec3 &n) : TiledPlane(o, n) {}
    const Material &GetMaterial(const vec3 &pos) const override { n_materials.fetch_add(1, std::memory_order_relaxed); return TiledPlane::GetMaterial(pos); }
    vec3 GetSurfaceNormal(const vec3 &pos) const override { n_normals.fetch_add(1, std::memory_order_relaxed); return TiledPlane::GetSurfaceNormal(pos); }
    vec3 GetRayFrom(const vec3 &from) const override { n_ray_from.fetch_add(1, std::memory_order_relaxed); return TiledPlane::GetRayFrom(from); }
    bool Intersects(const vec3 &o, const vec3 &d, float &dist) const override { n_intersects.fetch_add(1, std::memory_order_relaxed); return TiledPlane::Intersects(o, d, dist); }
};

} // namespace refh

extern "C" {

int ref_trace_max_depth(void) { return MAX_DEPTH; }
int ref_trace_image_width(void) { return ImageWidth; }
int ref_trace_image_height(void) { return ImageHeight; }

// InitScene (main.cpp:21-63) verbatim, into empty globals.
void ref_trace_init_scene(void)
{
    sceneObjects.clear();
    pCamera.reset();
    InitScene();
}

// Fills sceneObjects / pCamera from the C ABI's PODs, list order kept.  counting != 0 uses the counting
// subclasses.
int ref_trace_set_scene(const rt_object *objs, int n, const rt_camera *cam, int counting)
{
    using namespace refh;
    sceneObjects.clear();
    for (int k = 0; k < n; k++) {
        const rt_object &o = objs[k];
        if (o.type == RT_OBJ_SPHERE) {
            if (counting) sceneObjects.push_back(std::make_shared<CountingSphere>(to_material(o.mat), V(o.a), o.radius));
            else sceneObjects.push_back(std::make_shared<Sphere>(to_material(o.mat), V(o.a), o.radius));
        } else if (o.type == RT_OBJ_TILED_PLANE) {
            std::shared_ptr<TiledPlane> p;
            if (counting) p = std::make_shared<CountingPlane>(V(o.a), V(o.b));
            else p = std::make_shared<TiledPlane>(V(o.a), V(o.b));
            p->whiteMat = to_material(o.mat);      // the constructor's own values for the default plane; here as data
            p->blackMat = to_material(o.mat2);
            sceneObjects.push_back(p);
        } else {
            return -1;
        }
    }
    // any constructor arguments do: all eight fields are overwritten with the caller's (which the tests
    // pin separately against the constructor, tests/golden/camera.json)
    pCamera = std::make_shared<Camera>(vec3(0.0f), vec3(0.0f, 0.0f, -1.0f), 60.0f, 4, 4u);
    pCamera->position = V(cam->position);
    pCamera->viewDirection = V(cam->view_direction);
    pCamera->right = V(cam->right);
    pCamera->up = V(cam->up);
    pCamera->aspectRatio = cam->aspect_ratio;
    pCamera->halfAngle = cam->half_angle;
    pCamera->imageWidth = cam->image_width;
    pCamera->imageHeight = cam->image_height;
    return 0;
}

// Dumps the current globals (after ref_trace_init_scene: InitScene's data) as C-ABI PODs.  The emissive
// member that InitScene leaves unassigned for its first two spheres (main.cpp:29-39) and that TiledPlane
// never assigns (sceneobjects.h:93-104) is reported as 0, the value include/rt_api.h defines for it.
int ref_trace_get_scene(rt_object *out, int cap, rt_camera *cam)
{
    using namespace refh;
    const int n = (int)sceneObjects.size();
    for (int k = 0; k < n && k < cap; k++) {
        rt_object &d = out[k];
        memset(&d, 0, sizeof d);
        SceneObject *o = sceneObjects[k].get();
        if (o->GetSceneObjectType() == SceneObjectType::Sphere) {
            Sphere *s = static_cast<Sphere *>(o);
            d.type = RT_OBJ_SPHERE;
            S(d.a, s->center);
            d.radius = s->radius;
            from_material(d.mat, s->material, k >= 2);     // first assignment of emissive: main.cpp:45
        } else {
            TiledPlane *p = static_cast<TiledPlane *>(o);
            d.type = RT_OBJ_TILED_PLANE;
            S(d.a, p->origin);
            S(d.b, p->normal);
            from_material(d.mat, p->whiteMat, false);
            from_material(d.mat2, p->blackMat, false);
        }
    }
    if (cam && pCamera) {
        S(cam->position, pCamera->position);
        S(cam->view_direction, pCamera->viewDirection);
        S(cam->right, pCamera->right);
        S(cam->up, pCamera->up);
        cam->aspect_ratio = pCamera->aspectRatio;
        cam->half_angle = pCamera->halfAngle;
        cam->image_width = pCamera->imageWidth;
        cam->image_height = pCamera->imageHeight;
    }
    return n;
}

// TraceRay (main.cpp:84-188) verbatim.
void ref_trace_ray(const float o[3], const float d[3], int depth, float out[3])
{
    refh::S(out, TraceRay(refh::V(o), refh::V(d), depth));
}

// FindNearestObject (main.cpp:65-82) verbatim; returns the list index of the object or -1.
int ref_trace_find_nearest(const float o[3], const float d[3], float *dist)
{
    SceneObject *obj = FindNearestObject(refh::V(o), refh::V(d), *dist);
    if (!obj) return -1;
    for (size_t k = 0; k < sceneObjects.size(); k++)
        if (sceneObjects[k].get() == obj) return (int)k;
    return -2;
}

// The two statements of DrawScene's sample loop (main.cpp:207-208) for n image samples, on `threads`
// threads: ray = pCamera->GetWorldRay(sample); TraceRay(pCamera->position, ray, 0).
void ref_trace_samples(const float *sxy, int n, float *rgb_out, int threads)
{
    if (threads < 1) threads = 1;
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
        pool.emplace_back([=]() {
            for (int i = t; i < n; i += threads) {
                vec2 sample(sxy[2 * i], sxy[2 * i + 1]);
                auto ray = pCamera->GetWorldRay(sample);
                refh::S(rgb_out + 3 * i, TraceRay(pCamera->position, ray, 0));
            }
        });
    for (auto &t : pool) t.join();
}

// DrawScene's pixel loop (main.cpp:196-217) at the camera's own image size and with a caller-given
// sample pattern, built from the reference's GetWorldRay, TraceRay, glm::clamp and its float -> uint8_t
// conversion, statement for statement; used for sizes other than ImageWidth x ImageHeight, and checked
// against the verbatim DrawScene at that size by tests/gen_golden.py before anything is recorded from it.
// f32x4: the colour after `color *= (1.0f / numSamples)`, w = 0.  Either output may be NULL.
void ref_trace_frame(int width, int height, int spp, const float *offsets, int row_begin, int row_end, uint8_t *rgb8,
                     float *f32x4, int threads)
{
    if (threads < 1) threads = 1;
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
        pool.emplace_back([=]() {
            for (int y = row_begin + t; y < row_end; y += threads) {
                for (int x = 0; x < width; x++) {
                    vec3 color{0.0f, 0.0f, 0.0f};
                    for (auto i = 0; i < spp; i++) {
                        vec2 sample(float(x) + offsets[2 * i], float(y) + offsets[2 * i + 1]);
                        auto ray = pCamera->GetWorldRay(sample);
                        color += TraceRay(pCamera->position, ray, 0);
                    }
                    color *= (1.0f / spp);
                    const size_t idx = (size_t)y * (size_t)width + (size_t)x;
                    if (f32x4) { f32x4[4 * idx] = color.x; f32x4[4 * idx + 1] = color.y; f32x4[4 * idx + 2] = color.z; f32x4[4 * idx + 3] = 0.0f; }
                    color = color * 255.0f;
                    color = clamp(color, vec3(0.0f, 0.0f, 0.0f), vec3(255.0f, 255.0f, 255.0f));
                    if (rgb8) { rgb8[3 * idx] = uint8_t(color.x); rgb8[3 * idx + 1] = uint8_t(color.y); rgb8[3 * idx + 2] = uint8_t(color.z); }
                }
            }
        });
    for (auto &t : pool) t.join();
}

// What main() does around DrawScene (main.cpp:240-255) minus the option parsing and the clock:
// CreateBitmap(ImageWidth, ImageHeight), ClearBitmap(127), DrawScene verbatim, WriteBitmap, and a copy
// of Bitmap::pData for the caller.  Uses whatever scene the globals hold.
int ref_trace_draw_scene(const char *bmp_path, int partitions, int antialias, uint8_t *rgb8_out)
{
    Bitmap *pBitmap = CreateBitmap(ImageWidth, ImageHeight);
    Color col{127, 127, 127};
    ClearBitmap(pBitmap, col);
    DrawScene(pBitmap, partitions, antialias == 1 ? true : false);
    if (bmp_path) WriteBitmap(pBitmap, const_cast<char *>(bmp_path));
    if (rgb8_out) memcpy(rgb8_out, pBitmap->pData, (size_t)ImageWidth * (size_t)ImageHeight * 3);
    DestroyBitmap(pBitmap);
    return 0;
}

void ref_trace_reset_counters(void)
{
    refh::n_intersects = 0; refh::n_normals = 0; refh::n_ray_from = 0; refh::n_materials = 0;
}

// out[0] Intersects calls, [1] GetSurfaceNormal calls (= hits), [2] GetRayFrom calls (= shadow rays as
// written: hits x objects), [3] GetMaterial calls
void ref_trace_get_counters(unsigned long long out[4])
{
    out[0] = refh::n_intersects; out[1] = refh::n_normals; out[2] = refh::n_ray_from; out[3] = refh::n_materials;
}

} // extern "C"
