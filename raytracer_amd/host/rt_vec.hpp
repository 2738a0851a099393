// rt_vec.hpp -- the few vector operations the host API needs, with GLM 0.9.9.0's operation order
// (raytracer_cpp/glm/glm/detail/func_geometric.inl:54-116) so that host-side results (camera basis,
// SceneObject queries) carry the same bits as the reference's.  Build with -ffp-contract=off.
#pragma once
#include <cmath>

struct vec2 {
    float x = 0.0f, y = 0.0f;
    vec2() = default;
    vec2(float x_, float y_) : x(x_), y(y_) {}
};

struct vec3 {
    float x = 0.0f, y = 0.0f, z = 0.0f;
    vec3() = default;
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    vec3 &operator+=(const vec3 &o) { x += o.x; y += o.y; z += o.z; return *this; }
    vec3 &operator-=(const vec3 &o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
    vec3 &operator*=(float s) { x *= s; y *= s; z *= s; return *this; }
};

inline vec3 operator+(vec3 a, const vec3 &b) { return a += b; }
inline vec3 operator-(vec3 a, const vec3 &b) { return a -= b; }
inline vec3 operator*(const vec3 &a, const vec3 &b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator*(vec3 a, float s) { return a *= s; }
inline bool operator==(const vec3 &a, const vec3 &b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

inline float dot(const vec3 &a, const vec3 &b)
{
    const vec3 t = a * b;
    return t.x + t.y + t.z;
}
inline vec3 cross(const vec3 &a, const vec3 &b)
{
    return vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
inline vec3 normalize(const vec3 &v) { return v * (1.0f / std::sqrt(dot(v, v))); }
inline vec3 reflect(const vec3 &i, const vec3 &n) { return i - (n * dot(n, i)) * 2.0f; }
