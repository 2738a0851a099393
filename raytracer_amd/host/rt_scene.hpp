// rt_scene.hpp -- host-side scene API with the reference's names and meaning
// (raytracer_cpp/sceneobjects.h:3-131): Material, SceneObjectType, SceneObject and its five
// virtuals, Sphere, Plane, TiledPlane.  A host program builds its scene with these exactly as it
// does against the reference; DrawScene (rt_draw.hpp) flattens the list for the device.  The
// virtuals evaluate on the CPU for host-side queries only -- rendering never goes through them.
#pragma once
#include <cfloat>
#include <climits>
#include <cmath>

#include "rt_api.h"
#include "rt_vec.hpp"

struct Material {
    vec3 albedo;
    vec3 specular;
    float reflectance = 0.0f;
    vec3 emissive;          // the reference leaves this uninitialised until first assigned; here it is 0
};

enum class SceneObjectType { Sphere, Plane };

struct SceneObject {
    virtual ~SceneObject() = default;
    virtual const Material &GetMaterial(const vec3 &pos) const = 0;
    virtual SceneObjectType GetSceneObjectType() const = 0;
    virtual vec3 GetSurfaceNormal(const vec3 &pos) const = 0;
    virtual vec3 GetRayFrom(const vec3 &from) const = 0;
    virtual bool Intersects(const vec3 &rayOrigin, const vec3 &rayDir, float &distance) const = 0;
    // device flattening (no counterpart in the reference)
    virtual void Flatten(rt_object &out) const = 0;
};

namespace rt_detail {
inline void store(float *dst, const vec3 &v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }
inline rt_material flatten(const Material &m)
{
    rt_material r;
    store(r.albedo, m.albedo);
    store(r.specular, m.specular);
    r.reflectance = m.reflectance;
    store(r.emissive, m.emissive);
    return r;
}
} // namespace rt_detail

struct Sphere : SceneObject {
    vec3 center;
    float radius;
    Material material;

    Sphere(const Material &mat, const vec3 &c, const float r) : center(c), radius(r), material(mat) {}

    const Material &GetMaterial(const vec3 &) const override { return material; }
    SceneObjectType GetSceneObjectType() const override { return SceneObjectType::Sphere; }
    vec3 GetSurfaceNormal(const vec3 &pos) const override { return normalize(pos - center); }
    vec3 GetRayFrom(const vec3 &from) const override { return normalize(center - from); }

    // glm::intersectRaySphere on normalize(rayDir) with radius^2 (gtx/intersect.inl:105-123)
    bool Intersects(const vec3 &rayOrigin, const vec3 &rayDir, float &distance) const override
    {
        const vec3 dn = normalize(rayDir);
        const vec3 diff = center - rayOrigin;
        const float t0 = dot(diff, dn);
        const float dsq = dot(diff, diff) - t0 * t0;
        const float r2 = radius * radius;
        if (dsq > r2) return false;
        const float t1 = std::sqrt(r2 - dsq);
        distance = t0 > t1 + FLT_EPSILON ? t0 - t1 : t0 + t1;
        return distance > FLT_EPSILON;
    }

    void Flatten(rt_object &out) const override
    {
        out = rt_object{};
        out.type = RT_OBJ_SPHERE;
        rt_detail::store(out.a, center);
        out.radius = radius;
        out.mat = out.mat2 = rt_detail::flatten(material);
    }
};

struct Plane : SceneObject {
    vec3 normal;
    vec3 origin;
    SceneObjectType GetSceneObjectType() const override { return SceneObjectType::Plane; }
};

struct TiledPlane : Plane {
    Material blackMat;
    Material whiteMat;

    TiledPlane(const vec3 &o, const vec3 &n)
    {
        normal = n;
        origin = o;
        blackMat.reflectance = whiteMat.reflectance = 0.6f;
        whiteMat.specular = whiteMat.albedo = vec3(1.0f);
    }

    // checkerboard: int(floor(x) + floor(z)) & 1 in double, with x86's out-of-range conversion
    const Material &GetMaterial(const vec3 &pos) const override
    {
        const double s = std::floor((double)pos.x) + std::floor((double)pos.z);
        const int cell = (s > -2147483649.0 && s < 2147483648.0) ? (int)s : INT_MIN;
        return (cell & 1) == 0 ? whiteMat : blackMat;
    }
    vec3 GetSurfaceNormal(const vec3 &) const override { return normal; }
    vec3 GetRayFrom(const vec3 &from) const override { return normalize(origin - from); }

    // glm::intersectRayPlane (gtx/intersect.inl:7-24): one-sided, no t > 0 test
    bool Intersects(const vec3 &rayOrigin, const vec3 &rayDir, float &distance) const override
    {
        const float d = dot(rayDir, normal);
        if (d < FLT_EPSILON) {
            distance = dot(origin - rayOrigin, normal) / d;
            return true;
        }
        return false;
    }

    void Flatten(rt_object &out) const override
    {
        out = rt_object{};
        out.type = RT_OBJ_TILED_PLANE;
        rt_detail::store(out.a, origin);
        rt_detail::store(out.b, normal);
        out.mat = rt_detail::flatten(whiteMat);
        out.mat2 = rt_detail::flatten(blackMat);
    }
};
