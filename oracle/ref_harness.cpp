// ref_harness.cpp -- TEST INFRASTRUCTURE.  Thin extern "C" driver over the REFERENCE'S OWN headers,
// compiled where they lie under /root/reference (never copied): raytracer_cpp/common.h,
// sceneobjects.h, camera.h, writebitmap.h and the vendored GLM 0.9.9.0.  Output: oracle/_ref/libref.so
// (git-ignored).  It exposes the reference's primitives so that tests/gen_golden.py can record
// known-answer fixtures under tests/golden/ and so that oracle/rt_oracle.c can be validated.
//
// The trace loop of raytracer_cpp/main.cpp (InitScene, FindNearestObject, TraceRay, DrawScene) is
// compiled by its own recipe: ref_trace_harness.cpp -> oracle/_ref/libref_trace_d<MAX_DEPTH>.so.
#include <cassert>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "common.h"
#include "writebitmap.h"
#include "sceneobjects.h"
#include "camera.h"

#include "../include/rt_api.h"

static vec3 V(const float *p) { return vec3(p[0], p[1], p[2]); }
static void S(float *o, const vec3 &v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

extern "C" {

int ref_sphere_intersects(const float c[3], float r, const float o[3], const float d[3], float *dist)
{
    Material m{};
    Sphere s(m, V(c), r);
    const SceneObject &obj = s;
    return obj.Intersects(V(o), V(d), *dist) ? 1 : 0;
}

int ref_plane_intersects(const float po[3], const float pn[3], const float o[3], const float d[3], float *dist)
{
    TiledPlane p(V(po), V(pn));
    const SceneObject &obj = p;
    return obj.Intersects(V(o), V(d), *dist) ? 1 : 0;
}

int ref_checker_white(const float pos[3])
{
    TiledPlane p(vec3(0.0f), vec3(0.0f, 1.0f, 0.0f));
    return &p.GetMaterial(V(pos)) == &p.whiteMat ? 1 : 0;
}

void ref_plane_materials(rt_material *white, rt_material *black)
{
    TiledPlane p(vec3(0.0f), vec3(0.0f, 1.0f, 0.0f));
    const Material *src[2] = {&p.whiteMat, &p.blackMat};
    rt_material *dst[2] = {white, black};
    for (int i = 0; i < 2; i++) {
        S(dst[i]->albedo, src[i]->albedo);
        S(dst[i]->specular, src[i]->specular);
        dst[i]->reflectance = src[i]->reflectance;
        dst[i]->emissive[0] = dst[i]->emissive[1] = dst[i]->emissive[2] = 0.0f; // never set by the reference
    }
}

void ref_sphere_normal(const float c[3], float r, const float pos[3], float out[3])
{
    Material m{};
    Sphere s(m, V(c), r);
    S(out, s.GetSurfaceNormal(V(pos)));
}

void ref_sphere_ray_from(const float c[3], float r, const float from[3], float out[3])
{
    Material m{};
    Sphere s(m, V(c), r);
    S(out, s.GetRayFrom(V(from)));
}

void ref_reflect_dir(const float i[3], const float n[3], float out[3])
{
    S(out, glm::normalize(glm::reflect(V(i), V(n))));   // the expression at main.cpp:100
}

void ref_camera_init(rt_camera *cam, const float pos[3], const float dir[3], float fov, int w, int h)
{
    Camera c(V(pos), V(dir), fov, w, (unsigned)h);
    S(cam->position, c.position);
    S(cam->view_direction, c.viewDirection);
    S(cam->right, c.right);
    S(cam->up, c.up);
    cam->aspect_ratio = c.aspectRatio;
    cam->half_angle = c.halfAngle;
    cam->image_width = c.imageWidth;
    cam->image_height = c.imageHeight;
}

void ref_world_ray(const float pos[3], const float dir[3], float fov, int w, int h, float sx, float sy, float out[3])
{
    Camera c(V(pos), V(dir), fov, w, (unsigned)h);
    S(out, c.GetWorldRay(vec2(sx, sy)));
}

int ref_write_bitmap(const char *path, int w, int h, const uint8_t *rgb8)
{
    Bitmap *b = CreateBitmap(w, h);
    memcpy(b->pData, rgb8, (size_t)w * (size_t)h * 3);
    WriteBitmap(b, const_cast<char *>(path));
    DestroyBitmap(b);
    return 0;
}

void ref_sizes(int out[4])
{
    out[0] = (int)sizeof(Material);
    out[1] = (int)sizeof(Sphere);
    out[2] = (int)sizeof(TiledPlane);
    out[3] = (int)sizeof(Camera);
}

} // extern "C"
