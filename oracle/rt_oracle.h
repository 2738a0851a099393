/*
 * rt_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of raytracer_cpp's per-pixel trace loop, used only as the checker by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under raytracer_amd/
 * may include, link or call it.
 *
 * Pinning: PINNED TO THE REFERENCE ITSELF.  oracle/Makefile compiles the reference's own trace loop
 * (raytracer_cpp/main.cpp's InitScene, FindNearestObject, TraceRay, DrawScene, from the file where it
 * lies, minus the CmdParser include :10 and main() :230-258, which the loop does not use) and its
 * headers into the libraries under oracle/_ref; tests/gen_golden.py records their outputs into tests/golden/ (image.bmp
 * md5s for -a 0 / -a 1, whole-frame radiance hashes, >1000 radiance values, call counters by depth, the
 * synthetic scenes at MAX_DEPTH 5 and 8, 14 edge scenes at MAX_DEPTH 0/1/5/15, BASELINE's configs 2 and
 * 3, ~1300 primitive known answers), and tests/test_oracle_golden.py holds this restatement to every
 * one of them bit for bit -- and, where oracle/_ref is present, to the reference's TraceRay live on
 * random scenes.
 *
 * It shares only the POD layouts of include/rt_api.h with the product.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include "../include/rt_api.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_stats {
    uint64_t primary_rays, reflection_rays, shadow_rays_pruned, shadow_rays_verbatim;
    uint64_t calls, hits, lit;
    uint64_t nearest_tests, shadow_tests;         /* object tests actually executed            */
    uint64_t sphere_tests, sphere_accepts;        /* executed; accepts passed dSq <= r^2       */
    uint64_t plane_tests, plane_accepts;
    uint64_t calls_by_depth[16];
} orc_stats;

/* camera.h:18-45 */
void orc_camera_init(rt_camera *cam, const float pos[3], const float dir[3], float fov, int w, int h);
/* camera.h:48-62 */
void orc_world_ray(const rt_camera *cam, float sx, float sy, float out[3]);
/* sceneobjects.h:69-73 + glm/gtx/intersect.inl:105-123 (normalises dir itself, like the reference) */
int orc_sphere_intersects(const float center[3], float radius, const float orig[3], const float dir[3], float *dist);
/* sceneobjects.h:127-130 + glm/gtx/intersect.inl:7-24 */
int orc_plane_intersects(const float porig[3], const float pnormal[3], const float orig[3], const float dir[3], float *dist);
/* sceneobjects.h:106-115; returns 1 for white */
int orc_checker_white(const float pos[3]);
void orc_sphere_normal(const float c[3], const float pos[3], float out[3]);   /* sceneobjects.h:59-62 */
void orc_ray_from(const float c[3], const float from[3], float out[3]);       /* sceneobjects.h:64-67 */
void orc_reflect_dir(const float i[3], const float n[3], float out[3]);       /* main.cpp:100 */
/* main.cpp:84-188.  verbatim != 0: every object is a candidate emitter (as written, O(N^2));
 * verbatim == 0: only objects with non-zero emissive (bit-identical). */
void orc_trace(const rt_object *objs, int n, const float orig[3], const float dir[3], int depth,
               int max_depth, int verbatim, float out[3], orc_stats *st);
/* main.cpp:190-228.  Either output may be NULL.  rgb8: Bitmap::pData layout; f32x4: radiance after
 * the 1/spp scale, w = number of rays the pixel cost
 * (primary + reflection + pruned shadow rays, all samples).  Rows [row_begin,row_end) are rendered into full-image-indexed buffers.
 * threads >= 1 interleaves rows like the reference's partitions. */
void orc_lens_ray(const rt_camera *cam, float focal_length, float lens_x, float lens_y, float sx, float sy,
                  float out_o[3], float out_d[3]);
void orc_render(const rt_object *objs, int n, const rt_camera *cam, const rt_render_params *p,
                int verbatim, int threads, uint8_t *rgb8, float *f32x4, orc_stats *st);
/* Radiance of individual samples (sx, sy in pixel units) -- for sparse-pixel fixtures. */
void orc_trace_samples(const rt_object *objs, int n, const rt_camera *cam, int max_depth,
                       const float *sxy, int n_samples, float *rgb_out);
/* The same with the emitter loop as written (verbatim != 0) or pruned (0). */
void orc_trace_samples_v(const rt_object *objs, int n, const rt_camera *cam, int max_depth, int verbatim,
                         const float *sxy, int n_samples, float *rgb_out);
/* writebitmap.h:88-182, byte-identical output */
int orc_write_bmp(const char *path, int w, int h, const uint8_t *rgb8);

#ifdef __cplusplus
}
#endif
#endif
