// host_san_driver.cpp -- the host-only entry points of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only: no
// HIP call is made).  rt_plan_groups (the Morton sort, the greedy grouping and the bisection behind RT_SWEEP_GROUPED: the host code
// that runs on every rt_upload_scene), rt_make_scene, rt_split_bands, rt_camera_init, rt_bmp_header / rt_write_bmp,
// rt_interleaved_*, on the built-in scenes and on random ones of the kinds tests/fuzz_sweeps.py renders (slabs, blobs, shells, lines,
// coincident spheres, huge and tiny radii, NaN / inf entries, shuffled order).  Built and run by tests/test_host_sanitizers.py.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "rt_api.h"

static uint32_t g_state = 0x5EED1234u;
static float rnd() { g_state = g_state * 1664525u + 1013904223u; return (float)(g_state >> 8) * (1.0f / 16777216.0f); }

static int check_groups(const std::vector<rt_object> &objs, const char *what)
{
    const int n = (int)objs.size();
    int n_spheres = 0;
    for (const rt_object &o : objs) n_spheres += o.type == RT_OBJ_SPHERE;
    int32_t ng = 0, ns = 0;
    const int cap = n + 8;
    std::vector<float> gb(4 * (size_t)cap), sb(4 * (size_t)cap);
    std::vector<int32_t> of(n, -2);
    float sigma[3] = {0, 0, 0};
    const int rc = rt_plan_groups(objs.data(), n, cap, &ng, &ns, gb.data(), sb.data(), of.data(), sigma);
    if (rc != RT_OK) { fprintf(stderr, "%s: rt_plan_groups failed: %d\n", what, rc); return 1; }
    if (ng < 0 || ns < 0 || ng > cap || ns > cap) { fprintf(stderr, "%s: %d groups, %d super-groups of %d objects\n", what, ng, ns, n); return 1; }
    std::vector<int> members(ng > 0 ? ng : 1, 0);
    for (int k = 0; k < n; k++) {
        if (objs[k].type != RT_OBJ_SPHERE) { if (of[k] != -1) { fprintf(stderr, "%s: plane %d in group %d\n", what, k, of[k]); return 1; } continue; }
        if (of[k] < -1 || of[k] >= ng) { fprintf(stderr, "%s: sphere %d in group %d of %d\n", what, k, of[k], ng); return 1; }
        if (of[k] >= 0) members[of[k]]++;
    }
    for (int g = 0; g < ng; g++)
        if (members[g] > 8) { fprintf(stderr, "%s: group %d holds %d spheres\n", what, g, members[g]); return 1; }
    // too small a buffer is reported, not overrun
    if (ng > 1) {
        std::vector<float> small(4), small2(4 * (size_t)cap);
        int32_t ng2 = 0, ns2 = 0;
        const int rc2 = rt_plan_groups(objs.data(), n, 1, &ng2, &ns2, small.data(), small2.data(), of.data(), sigma);
        if (rc2 != RT_ERR_BUFFER_TOO_SMALL || ng2 != ng) { fprintf(stderr, "%s: max_groups 1 gave %d (%d groups)\n", what, rc2, ng2); return 1; }
    }
    (void)n_spheres;
    return 0;
}

static rt_object sphere(float x, float y, float z, float r, bool emissive)
{
    rt_object o;
    memset(&o, 0, sizeof o);
    o.type = RT_OBJ_SPHERE;
    o.a[0] = x; o.a[1] = y; o.a[2] = z;
    o.radius = r;
    o.mat.albedo[0] = rnd(); o.mat.reflectance = rnd() < 0.3f ? 0.5f : 0.0f;
    if (emissive) o.mat.emissive[0] = o.mat.emissive[1] = o.mat.emissive[2] = 1.0f;
    return o;
}

static std::vector<rt_object> random_scene(int kind, int n)
{
    std::vector<rt_object> v;
    const float inf = std::numeric_limits<float>::infinity(), nan = std::numeric_limits<float>::quiet_NaN();
    for (int i = 0; i < n; i++) {
        float x = 0, y = 0, z = 0, r = 0.05f + 2.0f * rnd() * rnd();
        switch (kind % 6) {
        case 0: x = 200 * rnd() - 100; y = r; z = 200 * rnd() - 100; break;                       // a slab
        case 1: x = 3 * (rnd() - .5f); y = 3 * (rnd() - .5f); z = 3 * (rnd() - .5f); break;       // a blob
        case 2: { const float t = 6.2831853f * rnd(), p = 3.14159f * rnd(); x = 50 * sinf(p) * cosf(t); y = 50 * cosf(p); z = 50 * sinf(p) * sinf(t); break; }
        case 3: x = 1000 * rnd(); y = 0; z = 0; r = 0.01f; break;                                  // a line
        case 4: x = 1; y = 2; z = 3; r = 1; break;                                                 // coincident
        default: x = 1e6f * (rnd() - .5f); y = 1e-3f * rnd(); z = 1e4f * rnd(); r = rnd() < 0.1f ? 3600.0f : 1e-4f; break;
        }
        v.push_back(sphere(x, y, z, r, rnd() < 0.02f));
    }
    if (kind >= 6 && n > 4) {            // entries nothing can hit in the reference either
        v[1].a[0] = nan; v[2].radius = inf; v[3].a[2] = -inf; v[4].radius = nan;
    }
    if (kind & 1) {
        rt_object p;
        memset(&p, 0, sizeof p);
        p.type = RT_OBJ_TILED_PLANE;
        p.b[1] = 1.0f;
        v.push_back(p);
    }
    for (size_t i = v.size(); i > 1; i--) { const size_t j = (size_t)(rnd() * (float)i) % i; rt_object t = v[i - 1]; v[i - 1] = v[j]; v[j] = t; }
    return v;
}

int main()
{
    if (getenv("RT_SAN_SELFTEST")) {        // is the build instrumented at all?  (the test expects this run to die in ASan's hands)
        std::vector<int32_t> b(3, 0);
        volatile int past = 4;
        return rt_split_bands(nullptr, 100, past, b.data());      // writes bounds[0..4] into three ints
    }
    int bad = 0;
    for (int kind = 0; kind < 3; kind++) {
        const int32_t n = rt_make_scene(kind, 640, 360, 3, 240, nullptr, 0, nullptr);
        std::vector<rt_object> objs((size_t)n);
        rt_camera cam;
        if (n <= 0 || rt_make_scene(kind, 640, 360, 3, 240, objs.data(), n, &cam) != n) { fprintf(stderr, "rt_make_scene(%d) failed\n", kind); return 1; }
        if (rt_make_scene(kind, 640, 360, 0, 240, objs.data(), n - 1, &cam) >= 0 && n > 1) { /* a short buffer must not be overrun: ASan would say */ }
        bad += check_groups(objs, kind == 0 ? "default" : kind == 1 ? "grid1k" : "grid16k");
    }
    const int sizes[] = {1, 2, 7, 8, 9, 63, 64, 65, 513, 4097, 20000};
    for (int kind = 0; kind < 12; kind++)
        for (int n : sizes) {
            char what[64];
            snprintf(what, sizeof what, "random kind %d n %d", kind, n);
            bad += check_groups(random_scene(kind, n), what);
        }
    // bands
    for (int h : {1, 7, 768, 2160})
        for (int parts : {1, 2, 3, 8}) {
            std::vector<float> cost((size_t)h);
            for (float &c : cost) c = rnd() < 0.05f ? 1000.0f * rnd() : rnd();
            std::vector<int32_t> b((size_t)parts + 1, -1);
            if (rt_split_bands(cost.data(), h, parts, b.data()) != RT_OK || b[0] != 0 || b[parts] != h) { fprintf(stderr, "rt_split_bands %d/%d\n", h, parts); bad++; }
            for (int i = 0; i < parts; i++) if (b[i] > b[i + 1]) { fprintf(stderr, "rt_split_bands %d/%d not monotone\n", h, parts); bad++; }
            if (rt_split_bands(nullptr, h, parts, b.data()) != RT_OK) bad++;
        }
    // strips
    for (int w : {1, 8, 9, 1000, 3840})
        for (int parts : {1, 2, 3, 8}) {
            int total = 0;
            for (int ph = 0; ph < parts; ph++) {
                const int c = rt_interleaved_count(w, parts, ph);
                total += c;
                for (int k = 0; k < c; k++) { const int x = rt_interleaved_index(k, parts, ph); if (x < 0 || x >= w) { fprintf(stderr, "strip column %d of width %d\n", x, w); bad++; } }
            }
            if (total != w) { fprintf(stderr, "strips of width %d over %d parts cover %d columns\n", w, parts, total); bad++; }
        }
    // camera, BMP
    {
        rt_camera cam;
        const float pos[3] = {0, 6, 8}, dir[3] = {0, -.8f, -1};
        rt_camera_init(&cam, pos, dir, 60.0f, 1024, 768);
        if (!(cam.aspect_ratio > 1.3f && cam.aspect_ratio < 1.4f)) bad++;
        uint8_t hdr[54];
        if (rt_bmp_header(1023, 7, hdr) != RT_OK || hdr[0] != 'B' || hdr[1] != 'M') bad++;
        std::vector<uint8_t> rgb(3 * 1023 * 7, 200);
        char path[] = "/tmp/rt_host_san_XXXXXX";
        const int fd = mkstemp(path);
        if (fd >= 0) { if (rt_write_bmp(path, 1023, 7, rgb.data()) != RT_OK) bad++; remove(path); }
    }
    printf("%s\n", bad ? "FAILED" : "host entry points clean under ASan + UBSan");
    return bad ? 1 : 0;
}
