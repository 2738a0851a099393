/*
 * rt_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see rt_oracle.h).
 *
 * Plain-C restatement of the reference's trace loop.  Every function cites the reference lines it
 * follows (paths relative to /root/reference).  Must be built with -ffp-contract=off and without
 * -ffast-math: float operation ORDER is part of the contract.
 */
#include "rt_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } v3;

/* glm vec3 operators, glm/detail/type_vec3.inl:579-675: component-wise, scalar applied per lane */
static inline v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_from(const float *p) { return v3_make(p[0], p[1], p[2]); }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
/* glm/detail/func_geometric.inl:54-61: tmp = x*y; tmp.x + tmp.y + tmp.z */
static inline float v3_dot(v3 a, v3 b)
{
    v3 t = v3_mul(a, b);
    return t.x + t.y + t.z;
}
/* func_geometric.inl:74-85 */
static inline v3 v3_cross(v3 x, v3 y)
{
    return v3_make(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
/* func_geometric.inl:88-96 with func_exponential.inl:42-48: v * (1 / sqrt(dot(v,v))) */
static inline v3 v3_normalize(v3 v)
{
    float inv = 1.0f / sqrtf(v3_dot(v, v));
    return v3_scale(v, inv);
}
/* func_geometric.inl:110-116: I - N * dot(N,I) * 2 */
static inline v3 v3_reflect(v3 i, v3 n)
{
    float d = v3_dot(n, i);
    return v3_sub(i, v3_scale(v3_scale(n, d), 2.0f));
}

/* ---- camera ------------------------------------------------------------------------------- */

/* camera.h:18-45 */
void orc_camera_init(rt_camera *cam, const float pos[3], const float dir[3], float fov, int w, int h)
{
    v3 view = v3_from(dir);
    v3 right = v3_cross(view, v3_make(0.0f, 1.0f, 0.0f));       /* camera.h:28 */
    v3 up = v3_cross(right, view);                               /* camera.h:33 */
    up = v3_normalize(up);                                       /* camera.h:36-38 */
    right = v3_normalize(right);
    view = v3_normalize(view);
    cam->position[0] = pos[0]; cam->position[1] = pos[1]; cam->position[2] = pos[2];
    cam->view_direction[0] = view.x; cam->view_direction[1] = view.y; cam->view_direction[2] = view.z;
    cam->right[0] = right.x; cam->right[1] = right.y; cam->right[2] = right.z;
    cam->up[0] = up.x; cam->up[1] = up.y; cam->up[2] = up.z;
    cam->image_width = (float)w;                                 /* camera.h:24-25 */
    cam->image_height = (float)h;
    cam->aspect_ratio = (float)w / (float)h;                     /* camera.h:41 */
    /* camera.h:44 -- glm::radians(float) is a float multiply (func_trigonometric.inl:12-17),
     * the halving and tan are double */
    float rad = fov * 0.01745329251994329576923690768489f;
    cam->half_angle = (float)tan((double)rad / 2.0);
}

/* camera.h:48-62 */
void orc_world_ray(const rt_camera *cam, float sx, float sy, float out[3])
{
    v3 dir = v3_from(cam->view_direction);
    float x = ((sx * 2.0f) / cam->image_width) - 1.0f;
    float y = ((sy * 2.0f) / cam->image_height) - 1.0f;
    dir = v3_add(dir, v3_scale(v3_from(cam->right), cam->half_angle * cam->aspect_ratio * x));
    dir = v3_sub(dir, v3_scale(v3_from(cam->up), cam->half_angle * y));
    dir = v3_normalize(dir);
    out[0] = dir.x; out[1] = dir.y; out[2] = dir.z;
}

/* raytracer_win/camera.h:147-173 (thin-lens GetWorldRay) with the lens-plane offset handed in instead
 * of drawn from glm::circularRand(0.14f) (:154): origin = the lens point, direction = towards the
 * point of the pinhole ray at focus distance (focal_length - 1) / |dir|, focal_length =
 * glm::length(focalPoint - position) (:162).  filmWidth/filmHeight are image_width/image_height. */
void orc_lens_ray(const rt_camera *cam, float focal_length, float lens_x, float lens_y, float sx, float sy,
                  float out_o[3], float out_d[3])
{
    v3 dir = v3_from(cam->view_direction);                                  /* :156 */
    float x = ((sx * 2.0f) / cam->image_width) - 1.0f;                      /* :157 */
    float y = ((sy * 2.0f) / cam->image_height) - 1.0f;                     /* :158 */
    v3 right = v3_from(cam->right), up = v3_from(cam->up), pos = v3_from(cam->position);
    dir = v3_add(dir, v3_scale(right, cam->half_angle * cam->aspect_ratio * x));   /* :162 */
    dir = v3_sub(dir, v3_scale(up, cam->half_angle * y));                   /* :163 */
    float ft = (focal_length - 1.0f) / sqrtf(v3_dot(dir, dir));             /* :165, glm::length = sqrt(dot) */
    v3 focus = v3_add(pos, v3_scale(dir, ft));                              /* :166 */
    v3 lens = pos;                                                          /* :168 */
    lens = v3_add(lens, v3_scale(right, lens_x));                           /* :169 */
    lens = v3_add(lens, v3_scale(up, lens_y));                              /* :170 */
    dir = v3_normalize(v3_sub(focus, lens));                                /* :171 */
    out_o[0] = lens.x; out_o[1] = lens.y; out_o[2] = lens.z;
    out_d[0] = dir.x; out_d[1] = dir.y; out_d[2] = dir.z;
}

/* ---- primitives --------------------------------------------------------------------------- */

/* glm/gtx/intersect.inl:105-123, given the already normalised direction */
static inline int sphere_hit_n(v3 c, float r2, v3 orig, v3 dn, float *dist, orc_stats *st)
{
    const float eps = FLT_EPSILON;
    v3 diff = v3_sub(c, orig);
    float t0 = v3_dot(diff, dn);
    float dsq = v3_dot(diff, diff) - t0 * t0;
    if (st) st->sphere_tests++;
    if (dsq > r2) return 0;
    if (st) st->sphere_accepts++;
    float t1 = sqrtf(r2 - dsq);
    float d = t0 > t1 + eps ? t0 - t1 : t0 + t1;
    *dist = d;
    return d > eps;
}

/* glm/gtx/intersect.inl:7-24 */
static inline int plane_hit(v3 po, v3 pn, v3 orig, v3 dir, float *dist, orc_stats *st)
{
    const float eps = FLT_EPSILON;
    float d = v3_dot(dir, pn);
    if (st) st->plane_tests++;
    if (d < eps) {
        if (st) st->plane_accepts++;
        *dist = v3_dot(v3_sub(po, orig), pn) / d;
        return 1;
    }
    return 0;
}

int orc_sphere_intersects(const float center[3], float radius, const float orig[3], const float dir[3], float *dist)
{
    /* sceneobjects.h:69-73: normalize(rayDir), radius * radius */
    return sphere_hit_n(v3_from(center), radius * radius, v3_from(orig), v3_normalize(v3_from(dir)), dist, NULL);
}

int orc_plane_intersects(const float porig[3], const float pnormal[3], const float orig[3], const float dir[3], float *dist)
{
    return plane_hit(v3_from(porig), v3_from(pnormal), v3_from(orig), v3_from(dir), dist, NULL);
}

/* int(double) as g++/x86-64 executes it (cvttsd2si): out-of-range and NaN give INT_MIN.
 * The C standard leaves it undefined; the reference's images depend on it. */
static inline int x86_double_to_int(double d)
{
    if (!(d > -2147483649.0 && d < 2147483648.0)) return INT_MIN;
    return (int)d;
}

/* sceneobjects.h:106-115: floor is ::floor(double) under g++, the sum is double */
int orc_checker_white(const float pos[3])
{
    double s = floor((double)pos[0]) + floor((double)pos[2]);
    return (x86_double_to_int(s) & 1) == 0;
}

/* SceneObject virtuals, sceneobjects.h:49-73 and :106-130 */
static inline int obj_intersects(const rt_object *o, v3 orig, v3 dir, v3 dn, float *dist, orc_stats *st)
{
    if (o->type == RT_OBJ_SPHERE)
        return sphere_hit_n(v3_from(o->a), o->radius * o->radius, orig, dn, dist, st);
    return plane_hit(v3_from(o->a), v3_from(o->b), orig, dir, dist, st);
}

static inline const rt_material *obj_material(const rt_object *o, v3 pos)
{
    if (o->type == RT_OBJ_SPHERE) return &o->mat;
    float p[3] = {pos.x, pos.y, pos.z};
    return orc_checker_white(p) ? &o->mat : &o->mat2;
}

static inline v3 obj_normal(const rt_object *o, v3 pos)
{
    if (o->type == RT_OBJ_SPHERE) return v3_normalize(v3_sub(pos, v3_from(o->a))); /* :59-62 */
    return v3_from(o->b);                                                           /* :117-120 */
}

static inline v3 obj_ray_from(const rt_object *o, v3 from)
{
    return v3_normalize(v3_sub(v3_from(o->a), from));     /* :64-67 and :122-125 */
}

static inline int emissive_nonzero(const rt_material *m)
{
    /* main.cpp:131: !(emissive == vec3(0,0,0)) */
    return !(m->emissive[0] == 0.0f && m->emissive[1] == 0.0f && m->emissive[2] == 0.0f);
}

/* An object can light anything only if some material of it has non-zero emissive
 * ("emitter pruning is exact"). */
static inline int obj_may_emit(const rt_object *o)
{
    if (emissive_nonzero(&o->mat)) return 1;
    return o->type != RT_OBJ_SPHERE && emissive_nonzero(&o->mat2);
}

/* exported for the known-answer tests (tests/golden/vectors.json) */
void orc_sphere_normal(const float c[3], const float pos[3], float out[3])
{
    v3 n = v3_normalize(v3_sub(v3_from(pos), v3_from(c)));             /* sceneobjects.h:59-62 */
    out[0] = n.x; out[1] = n.y; out[2] = n.z;
}

void orc_ray_from(const float c[3], const float from[3], float out[3])
{
    v3 n = v3_normalize(v3_sub(v3_from(c), v3_from(from)));            /* sceneobjects.h:64-67 */
    out[0] = n.x; out[1] = n.y; out[2] = n.z;
}

void orc_reflect_dir(const float i[3], const float n[3], float out[3])
{
    v3 r = v3_normalize(v3_reflect(v3_from(i), v3_from(n)));           /* main.cpp:100 */
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* ---- TraceRay, main.cpp:84-188 ------------------------------------------------------------- */

static v3 trace(const rt_object *objs, int n, v3 orig, v3 dir, int depth, int max_depth, int verbatim, orc_stats *st)
{
    if (st) { st->calls++; if (depth < 16) st->calls_by_depth[depth]++; }

    /* FindNearestObject, main.cpp:65-82.  normalize(dir) is recomputed per sphere test in the
     * reference (sceneobjects.h:71); it is loop invariant, so hoisting it is bit-identical. */
    v3 dn = v3_normalize(dir);
    const rt_object *nearest = NULL;
    float nearest_d = FLT_MAX;
    for (int i = 0; i < n; i++) {
        float d;
        if (st) st->nearest_tests++;
        if (obj_intersects(&objs[i], orig, dir, dn, &d, st) && nearest_d > d) {
            nearest = &objs[i];
            nearest_d = d;
        }
    }
    if (!nearest) return v3_make(0.1f, 0.1f, 0.1f);                       /* :90-93 */
    if (st) st->hits++;

    v3 pos = v3_add(orig, v3_scale(dir, nearest_d));                      /* :94 */
    v3 normal = obj_normal(nearest, pos);                                 /* :95 */
    v3 out = v3_make(0.0f, 0.0f, 0.0f);
    const rt_material *mat = obj_material(nearest, pos);                  /* :98 */
    v3 refl = v3_normalize(v3_reflect(dir, normal));                      /* :100 */

    if (depth < max_depth && mat->reflectance > 0.0f) {                   /* :103-110 */
        if (st) st->reflection_rays++;
        v3 rc = trace(objs, n, v3_add(pos, v3_scale(refl, 0.001f)), refl, depth + 1, max_depth, verbatim, st);
        out = v3_scale(rc, mat->reflectance);
    }

    for (int e = 0; e < n; e++) {                                         /* :112 */
        const rt_object *em = &objs[e];
        if (st) st->shadow_rays_verbatim++;
        if (!verbatim && !obj_may_emit(em)) continue;
        if (st) st->shadow_rays_pruned++;
        v3 edir = obj_ray_from(em, pos);                                  /* :114 */
        v3 sorig = v3_add(pos, v3_scale(edir, 0.001f));
        v3 edn = v3_normalize(edir);
        float best = FLT_MAX;
        const rt_object *occluder = NULL;
        const rt_material *emat = NULL;
        for (int o = 0; o < n; o++) {                                     /* :119-151 */
            float d;
            if (st) st->shadow_tests++;
            if (obj_intersects(&objs[o], sorig, edir, edn, &d, st)) {
                if (o == e) {
                    if (best > d) {
                        best = d;
                        emat = obj_material(&objs[o], v3_add(pos, v3_scale(edir, d)));
                        if (!emissive_nonzero(emat)) emat = NULL;
                        else occluder = NULL;
                    }
                } else if (best > d) {
                    occluder = &objs[o];
                    emat = NULL;
                    best = d;
                }
            }
        }
        if (!emat || occluder) continue;                                  /* :154-157 */

        float diffuse_i = v3_dot(normal, edir);                           /* :162 */
        float spec_i = 0.0f;
        if (diffuse_i > 0.0f) {
            spec_i = v3_dot(refl, edir);                                  /* :166 */
            if (spec_i > 0.0f) {
                spec_i = (float)pow((double)spec_i, 10.0);                /* :169 ::pow(double,double) */
                spec_i = (0.0f < spec_i) ? spec_i : 0.0f;                 /* :170 std::max(0.0f, specI) */
            } else {
                spec_i = 0.0f;
            }
        } else {
            diffuse_i = 0.0f;
        }
        if (st) st->lit++;
        v3 l = v3_add(v3_scale(v3_mul(v3_from(emat->emissive), v3_from(mat->albedo)), diffuse_i),
                      v3_scale(v3_from(mat->specular), spec_i));          /* :183 */
        out = v3_add(out, l);
    }
    out = v3_scale(out, 1.f - mat->reflectance);                          /* :185 */
    out = v3_add(out, v3_from(mat->emissive));                            /* :186 */
    return out;
}

void orc_trace(const rt_object *objs, int n, const float orig[3], const float dir[3], int depth,
               int max_depth, int verbatim, float out[3], orc_stats *st)
{
    v3 c = trace(objs, n, v3_from(orig), v3_from(dir), depth, max_depth, verbatim, st);
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

/* ---- DrawScene, main.cpp:190-228 ----------------------------------------------------------- */

static const float k_ref_pattern[8] = {0.1f, 0.2f, 0.6f, 0.5f, 0.8f, 0.7f, 0.2f, 0.8f}; /* :202 */

/* uint8_t(float) as g++/x86-64 executes it: cvttss2si to 32 bits, low byte kept (NaN -> 0). */
static inline uint8_t x86_float_to_u8(float f)
{
    int i;
    if (!(f > -2147483904.0f && f < 2147483648.0f)) i = INT_MIN;
    else i = (int)f;
    return (uint8_t)(i & 0xff);
}

typedef struct {
    const rt_object *objs; int n; const rt_camera *cam; const rt_render_params *p;
    int verbatim, offset, stride; uint8_t *rgb8; float *f32x4; orc_stats st; int want_stats;
} job_t;

static void *render_rows(void *arg)
{
    job_t *j = (job_t *)arg;
    const rt_render_params *p = j->p;
    const float *pat = p->sample_offsets ? p->sample_offsets : k_ref_pattern;
    int y0 = p->row_begin, y1 = p->row_end;
    if (y0 == 0 && y1 == 0) y1 = p->height;
    orc_stats *st = j->want_stats ? &j->st : NULL;
    v3 cpos = v3_from(j->cam->position);
    for (int y = y0 + j->offset; y < y1; y += j->stride) {                /* :196 */
        for (int x = 0; x < p->width; x++) {
            v3 color = v3_make(0.0f, 0.0f, 0.0f);
            orc_stats px_st;
            memset(&px_st, 0, sizeof px_st);
            for (int s = 0; s < p->spp; s++) {                            /* :203-209 */
                float sx = (float)x + pat[2 * s], sy = (float)y + pat[2 * s + 1];
                float d[3], o[3] = {cpos.x, cpos.y, cpos.z};
                if (p->lens_offsets)
                    orc_lens_ray(j->cam, p->focal_length, p->lens_offsets[2 * s], p->lens_offsets[2 * s + 1], sx, sy, o, d);
                else
                    orc_world_ray(j->cam, sx, sy, d);
                px_st.primary_rays++;
                color = v3_add(color, trace(j->objs, j->n, v3_from(o), v3_from(d), 0, p->max_depth, j->verbatim, &px_st));
            }
            color = v3_scale(color, 1.0f / (float)p->spp);                /* :210 */
            size_t idx = (size_t)p->width * (size_t)y + (size_t)x;
            if (p->accumulate && j->f32x4) {
                /* raytracer_win/main.cpp:292-314: bufferVal = ((bufferVal * k1) + vec4(color, 1)) * k2 */
                const float k1 = (float)p->accum_index, k2 = 1.f / (k1 + 1.f);
                float *o = j->f32x4 + 4 * idx;
                const float in[4] = {color.x, color.y, color.z, 1.0f};
                /* sample 0 starts from the zero-initialised buffer of InitMaps (:105) */
                for (int k = 0; k < 4; k++) o[k] = (((p->accum_index == 0 ? 0.0f : o[k]) * k1) + in[k]) * k2;
                /* CopyTargetToBitmap (:44-72) shows the mean: clamp(0,1) then *255 -- the same bytes as
                 * the *255 / clamp(0,255) below for every input, NaN included */
                color = v3_make(o[0], o[1], o[2]);
            } else if (j->f32x4) {
                float *o = j->f32x4 + 4 * idx;
                /* w = rays this pixel cost: primary + reflection + pruned shadow rays */
                o[0] = color.x; o[1] = color.y; o[2] = color.z;
                o[3] = (float)(px_st.primary_rays + px_st.reflection_rays + px_st.shadow_rays_pruned);
            }
            if (st) {
                const uint64_t *src = (const uint64_t *)&px_st;
                uint64_t *dst = (uint64_t *)st;
                for (size_t k = 0; k < sizeof(orc_stats) / sizeof(uint64_t); k++) dst[k] += src[k];
            }
            if (j->rgb8) {
                v3 c = v3_scale(color, 255.0f);                           /* :213 */
                /* :214 glm::clamp = min(max(x,0),255), func_common.inl:15-28,279-285 */
                float ch[3] = {c.x, c.y, c.z};
                for (int k = 0; k < 3; k++) {
                    float v = ch[k] > 0.0f ? ch[k] : 0.0f;
                    v = v < 255.0f ? v : 255.0f;
                    j->rgb8[3 * idx + k] = x86_float_to_u8(v);            /* :216 */
                }
            }
        }
    }
    return NULL;
}

void orc_render(const rt_object *objs, int n, const rt_camera *cam, const rt_render_params *p,
                int verbatim, int threads, uint8_t *rgb8, float *f32x4, orc_stats *st)
{
    if (threads < 1) threads = 1;
    job_t *jobs = (job_t *)calloc((size_t)threads, sizeof(job_t));
    pthread_t *tid = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    for (int i = 0; i < threads; i++) {                                   /* :193-195 */
        job_t *j = &jobs[i];
        j->objs = objs; j->n = n; j->cam = cam; j->p = p; j->verbatim = verbatim;
        j->offset = i; j->stride = threads; j->rgb8 = rgb8; j->f32x4 = f32x4; j->want_stats = st != NULL;
        if (threads == 1) render_rows(j);
        else pthread_create(&tid[i], NULL, render_rows, j);
    }
    if (threads > 1)
        for (int i = 0; i < threads; i++) pthread_join(tid[i], NULL);     /* :224-227 */
    if (st) {
        memset(st, 0, sizeof(*st));
        for (int i = 0; i < threads; i++) {
            const uint64_t *src = (const uint64_t *)&jobs[i].st;
            uint64_t *dst = (uint64_t *)st;
            for (size_t k = 0; k < sizeof(orc_stats) / sizeof(uint64_t); k++) dst[k] += src[k];
        }
    }
    free(jobs);
    free(tid);
}

void orc_trace_samples_v(const rt_object *objs, int n, const rt_camera *cam, int max_depth, int verbatim,
                         const float *sxy, int n_samples, float *rgb_out)
{
    v3 cpos = v3_from(cam->position);
    for (int i = 0; i < n_samples; i++) {
        float d[3];
        orc_world_ray(cam, sxy[2 * i], sxy[2 * i + 1], d);
        v3 c = trace(objs, n, cpos, v3_from(d), 0, max_depth, verbatim, NULL);
        rgb_out[3 * i] = c.x; rgb_out[3 * i + 1] = c.y; rgb_out[3 * i + 2] = c.z;
    }
}

void orc_trace_samples(const rt_object *objs, int n, const rt_camera *cam, int max_depth,
                       const float *sxy, int n_samples, float *rgb_out)
{
    orc_trace_samples_v(objs, n, cam, max_depth, 0, sxy, n_samples, rgb_out);
}

/* ---- WriteBitmap, writebitmap.h:88-182 ----------------------------------------------------- */

static void put_u32le(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)(v & 0xff); p[1] = (uint8_t)((v >> 8) & 0xff);
    p[2] = (uint8_t)((v >> 16) & 0xff); p[3] = (uint8_t)((v >> 24) & 0xff);
}

int orc_write_bmp(const char *path, int w, int h, const uint8_t *rgb8)
{
    int extrabytes = 4 - ((w * 3) % 4);                                   /* :96-100 */
    if (extrabytes == 4) extrabytes = 0;
    int paddedsize = ((w * 3) + extrabytes) * h;                          /* :102 */
    uint8_t hdr[54];
    memset(hdr, 0, sizeof hdr);
    hdr[0] = 'B'; hdr[1] = 'M';                                           /* :132 */
    put_u32le(hdr + 2, (uint32_t)(paddedsize + 54));                      /* :107 */
    put_u32le(hdr + 6, 0);
    put_u32le(hdr + 10, 54);
    put_u32le(hdr + 14, 40);
    put_u32le(hdr + 18, (uint32_t)w);
    put_u32le(hdr + 22, (uint32_t)h);
    hdr[26] = 1; hdr[27] = 0; hdr[28] = 24; hdr[29] = 0;                  /* :144-147 */
    put_u32le(hdr + 30, 0);
    put_u32le(hdr + 34, (uint32_t)paddedsize);
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    fwrite(hdr, 1, sizeof hdr, f);
    size_t rowbytes = (size_t)w * 3 + (size_t)extrabytes;
    uint8_t *row = (uint8_t *)calloc(rowbytes ? rowbytes : 1, 1);
    for (int y = h - 1; y >= 0; y--) {                                    /* :161-178 */
        const uint8_t *src = rgb8 + (size_t)w * 3 * (size_t)y;
        for (int x = 0; x < w; x++) {
            row[3 * x] = src[3 * x + 2];
            row[3 * x + 1] = src[3 * x + 1];
            row[3 * x + 2] = src[3 * x];
        }
        fwrite(row, 1, rowbytes, f);
    }
    free(row);
    fclose(f);
    return 0;
}
