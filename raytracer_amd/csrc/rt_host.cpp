// rt_host.cpp -- host-only half of the C ABI (include/rt_api.h): the Camera constructor, the
// scene generators, band splitting and the BMP writer.  No HIP here; usable without a device.
// Built with -ffp-contract=off: the camera basis and generated scenes feed bit-exact parity tests.
#include "rt_api.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

struct F3 { float x, y, z; };

inline F3 f3(const float *p) { return {p[0], p[1], p[2]}; }
inline void put(float *o, F3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }
// GLM 0.9.9.0 semantics (func_geometric.inl:54-96): dot = (xx + yy) + zz, cross as written there,
// normalize = v * (1 / sqrt(dot(v, v))).
inline float dot3(F3 a, F3 b) { float x = a.x * b.x, y = a.y * b.y, z = a.z * b.z; return x + y + z; }
inline F3 cross3(F3 a, F3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline F3 unit3(F3 v) { float s = 1.0f / sqrtf(dot3(v, v)); return {v.x * s, v.y * s, v.z * s}; }

// PRNG of SURVEY.md section 8d: s = s*1664525 + 1013904223 (u32); u = (s >> 8) * 2^-24
struct Lcg {
    uint32_t s;
    float next() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f); }
};

rt_material material(float ar, float ag, float ab, float sr, float sg, float sb, float refl,
                     float er, float eg, float eb)
{
    rt_material m;
    m.albedo[0] = ar; m.albedo[1] = ag; m.albedo[2] = ab;
    m.specular[0] = sr; m.specular[1] = sg; m.specular[2] = sb;
    m.reflectance = refl;
    m.emissive[0] = er; m.emissive[1] = eg; m.emissive[2] = eb;
    return m;
}

rt_object sphere(const rt_material &m, float cx, float cy, float cz, float r)
{
    rt_object o;
    memset(&o, 0, sizeof o);
    o.type = RT_OBJ_SPHERE;
    o.a[0] = cx; o.a[1] = cy; o.a[2] = cz;
    o.radius = r;
    o.mat = m;
    return o;
}

// TiledPlane(origin, normal), sceneobjects.h:93-104: white = albedo 1, specular 1, refl .6;
// black = albedo 0, specular 0, refl .6; emissive (never set by the reference) defined as 0.
rt_object tiled_plane(float ox, float oy, float oz, float nx, float ny, float nz)
{
    rt_object o;
    memset(&o, 0, sizeof o);
    o.type = RT_OBJ_TILED_PLANE;
    o.a[0] = ox; o.a[1] = oy; o.a[2] = oz;
    o.b[0] = nx; o.b[1] = ny; o.b[2] = nz;
    o.mat = material(1, 1, 1, 1, 1, 1, 0.6f, 0, 0, 0);
    o.mat2 = material(0, 0, 0, 0, 0, 0, 0.6f, 0, 0, 0);
    return o;
}

// InitScene, main.cpp:21-63 (object data only; the camera is set by the caller below).
void default_scene(std::vector<rt_object> &v)
{
    v.push_back(sphere(material(.7f, .1f, .1f, .9f, .1f, .1f, 0.5f, 0, 0, 0), 0.0f, 2.0f, 0.f, 2.0f));
    v.push_back(sphere(material(0.7f, 0.0f, 0.7f, 0.9f, 0.9f, 0.8f, 0.5f, 0, 0, 0), -2.5f, 1.0f, 2.f, 1.0f));
    v.push_back(sphere(material(0.0f, 0.3f, 1.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0, 0, 0), -0.0f, 0.5f, 3.f, 0.5f));
    v.push_back(sphere(material(1.0f, 1.0f, 1.0f, 0, 0, 0, .0f, 1.0f, 1.0f, 0.0f), 2.8f, 0.8f, 2.0f, 0.8f));
    v.push_back(sphere(material(0.0f, 0.8f, 0.0f, 0, 0, 0, 0.0f, 1.0f, 1.0f, 1.0f), -10.8f, 8.4f, 10.0f, 0.4f));
    F3 n = unit3({0.0f, 1.0f, 0.0f});                                   // main.cpp:62
    v.push_back(tiled_plane(0.0f, 0.0f, 0.0f, n.x, n.y, n.z));
}

// "grid" scenes of SURVEY.md section 8d: side x side jittered spheres on the plane, a few
// emissive spheres above, the tiled plane last.
void grid_scene(std::vector<rt_object> &v, int side, uint32_t seed, bool all_reflective, int n_emit)
{
    Lcg g{seed};
    const float half = 0.5f * (float)(side - 1);
    for (int j = 0; j < side; j++) {
        for (int i = 0; i < side; i++) {
            int k = j * side + i;
            float r = .25f + .35f * g.next();
            float cx = ((float)i - half) * 1.5f + (g.next() - .5f) * .4f;
            float cz = 4.0f - 1.5f * (float)j + (g.next() - .5f) * .4f;
            float a0 = g.next(), a1 = g.next(), a2 = g.next();
            float s0 = g.next() * .5f, s1 = g.next() * .5f, s2 = g.next() * .5f;
            float refl;
            if (all_reflective) refl = .3f + .5f * g.next();
            else refl = (k % 4 == 0) ? .5f : 0.0f;
            v.push_back(sphere(material(a0, a1, a2, s0, s1, s2, refl, 0, 0, 0), cx, r, cz, r));
        }
    }
    // emitters: pairs at x = +-(extent + 6), spread in z over the grid's depth
    const float extent = half * 1.5f;
    const float zc = 4.0f - 1.5f * half;
    const int pairs = n_emit / 2;
    for (int p = 0; p < pairs; p++) {
        float t = pairs > 1 ? (float)p / (float)(pairs - 1) : 0.5f;
        float z = zc + (t - 0.5f) * 2.0f * extent;
        float y = 14.0f + 0.25f * extent;
        float er = side > 64 ? 3.0f : 1.0f;
        rt_material m = material(1, 1, 1, 0, 0, 0, 0.0f, 1.0f, 1.0f, 1.0f);
        v.push_back(sphere(m, -(extent + 6.0f), y, z, er));
        v.push_back(sphere(m, (extent + 6.0f), y, z, er));
    }
    v.push_back(tiled_plane(0, 0, 0, 0, 1, 0));
}

} // namespace

extern "C" {

int rt_api_version(void) { return RT_API_VERSION; }

int32_t rt_interleaved_count(int32_t width, int32_t n, int32_t phase)
{
    if (width <= 0 || n < 1 || phase < 0 || phase >= n) return 0;
    const int32_t strips = (width + RT_STRIP_COLS - 1) / RT_STRIP_COLS;
    int32_t cols = 0;
    for (int32_t s = phase; s < strips; s += n) cols += std::min<int32_t>(RT_STRIP_COLS, width - s * RT_STRIP_COLS);
    return cols;
}

int32_t rt_interleaved_index(int32_t k, int32_t n, int32_t phase)
{
    return ((k / RT_STRIP_COLS) * n + phase) * RT_STRIP_COLS + k % RT_STRIP_COLS;
}

// Camera::Camera, camera.h:18-45
void rt_camera_init(rt_camera *cam, const float position[3], const float dir[3], float fov_degrees,
                    int32_t width, int32_t height)
{
    F3 view = f3(dir);
    F3 right = cross3(view, {0.0f, 1.0f, 0.0f});
    F3 up = cross3(right, view);
    put(cam->position, f3(position));
    put(cam->up, unit3(up));
    put(cam->right, unit3(right));
    put(cam->view_direction, unit3(view));
    cam->image_width = (float)width;
    cam->image_height = (float)height;
    cam->aspect_ratio = (float)width / (float)height;
    // glm::radians(float) multiplies in float; the halving and tan() are double (camera.h:44)
    float rad = fov_degrees * 0.01745329251994329576923690768489f;
    cam->half_angle = (float)tan((double)rad / 2.0);
}

int32_t rt_make_scene(int32_t kind, int32_t width, int32_t height, int32_t frame, int32_t n_frames,
                      rt_object *objs, int32_t cap, rt_camera *cam)
{
    std::vector<rt_object> v;
    float pos[3], dir[3];
    switch (kind) {
    case RT_SCENE_DEFAULT:
        default_scene(v);
        pos[0] = 0.0f; pos[1] = 6.0f; pos[2] = 8.0f;                   // main.cpp:23-26
        dir[0] = 0.0f; dir[1] = -.8f; dir[2] = -1.0f;
        break;
    case RT_SCENE_GRID1K:
        grid_scene(v, 32, 0x5EED0400u, false, 4);
        pos[0] = 0.0f; pos[1] = 10.0f; pos[2] = 12.0f;
        dir[0] = 0.0f; dir[1] = -.45f; dir[2] = -1.0f;
        break;
    case RT_SCENE_GRID16K: {
        grid_scene(v, 128, 0x5EED4000u, true, 8);
        // camera sweep: circle of radius 120, height 40, about the grid centre, looking at it
        const double zc = 4.0 - 1.5 * 63.5;
        int nf = n_frames > 0 ? n_frames : 240;
        double th = 2.0 * 3.14159265358979323846 * (double)frame / (double)nf;
        pos[0] = (float)(120.0 * cos(th)); pos[1] = 40.0f; pos[2] = (float)(zc + 120.0 * sin(th));
        dir[0] = 0.0f - pos[0]; dir[1] = 0.0f - pos[1]; dir[2] = (float)zc - pos[2];
        break;
    }
    default:
        return RT_ERR_INVALID_ARG;
    }
    if (cam) rt_camera_init(cam, pos, dir, 60.0f, width, height);
    int32_t n = (int32_t)v.size();
    if (objs) {
        int32_t m = n < cap ? n : cap;
        if (m > 0) memcpy(objs, v.data(), (size_t)m * sizeof(rt_object));
    }
    return n;
}

int rt_split_bands(const float *row_cost, int32_t height, int32_t n, int32_t *bounds)
{
    if (height < 0 || n < 1 || !bounds) return RT_ERR_INVALID_ARG;
    double total = 0.0;
    for (int32_t y = 0; y < height; y++) {
        double c = row_cost ? (double)row_cost[y] : 1.0;
        if (!(c >= 0.0)) return RT_ERR_INVALID_ARG;
        total += c;
    }
    bounds[0] = 0;
    double acc = 0.0;
    int32_t y = 0;
    for (int32_t b = 1; b < n; b++) {
        double target = total * (double)b / (double)n;
        // advance while adding the next row keeps us at or below the target (closest boundary)
        while (y < height) {
            double c = row_cost ? (double)row_cost[y] : 1.0;
            if (acc + 0.5 * c > target) break;
            acc += c;
            y++;
        }
        // keep every remaining band non-empty when there are enough rows
        int32_t min_y = b < height ? b : height;
        int32_t max_y = height - (n - b) > min_y ? height - (n - b) : min_y;
        if (y < min_y) { while (y < min_y) { acc += row_cost ? (double)row_cost[y] : 1.0; y++; } }
        if (y > max_y) { while (y > max_y) { y--; acc -= row_cost ? (double)row_cost[y] : 1.0; } }
        bounds[b] = y;
    }
    bounds[n] = height;
    return RT_OK;
}

// The 54 header bytes WriteBitmap emits (writebitmap.h:90-155) for a width x height image.
int rt_bmp_header(int32_t width, int32_t height, uint8_t out[54])
{
    if (!out || width <= 0 || height <= 0) return RT_ERR_INVALID_ARG;
    const size_t stride = ((size_t)width * 3 + 3) & ~(size_t)3;
    const uint32_t body = (uint32_t)(stride * (size_t)height);
    const uint32_t words[13] = {body + 54u, 0u, 54u, 40u, (uint32_t)width, (uint32_t)height,
                                1u | (24u << 16), 0u, body, 0u, 0u, 0u, 0u};
    out[0] = 'B'; out[1] = 'M';
    for (int i = 0; i < 13; i++)
        for (int b = 0; b < 4; b++) out[2 + 4 * i + b] = (uint8_t)(words[i] >> (8 * b));
    return RT_OK;
}

// WriteBitmap, writebitmap.h:88-182: identical bytes, one fwrite per row instead of one fprintf
// per byte (which costs ~100 M calls at 8K).
int rt_write_bmp(const char *path, int32_t width, int32_t height, const uint8_t *rgb8)
{
    if (!path || !rgb8 || width <= 0 || height <= 0) return RT_ERR_INVALID_ARG;
    int pad = (4 - (width * 3) % 4) % 4;
    size_t stride = (size_t)width * 3 + (size_t)pad;
    uint8_t head[54];
    rt_bmp_header(width, height, head);
    FILE *f = fopen(path, "wb");
    if (!f) return RT_ERR_INVALID_ARG;
    bool ok = fwrite(head, 1, sizeof head, f) == sizeof head;
    std::vector<uint8_t> line(stride, 0);
    for (int32_t y = height - 1; ok && y >= 0; y--) {
        const uint8_t *src = rgb8 + (size_t)y * (size_t)width * 3;
        for (int32_t x = 0; x < width; x++) {
            line[3 * (size_t)x + 0] = src[3 * (size_t)x + 2];
            line[3 * (size_t)x + 1] = src[3 * (size_t)x + 1];
            line[3 * (size_t)x + 2] = src[3 * (size_t)x + 0];
        }
        ok = fwrite(line.data(), 1, stride, f) == stride;
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? RT_OK : RT_ERR_INVALID_ARG;
}

} // extern "C"
