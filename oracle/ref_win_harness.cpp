// ref_win_harness.cpp -- TEST INFRASTRUCTURE.  Thin extern "C" driver over the reference's
// raytracer_win/camera.h (and its vendored GLM), compiled where it lies under /root/reference (never
// copied).  Output: oracle/_ref/libref_win.so (git-ignored).  tests/gen_golden.py records known
// answers of the thin-lens Camera::GetWorldRay (camera.h:147-173) from it.
//
// The lens sample is drawn inside GetWorldRay with glm::circularRand (rand() underneath): the
// harness seeds rand(), draws the same circularRand first to learn the offset, re-seeds and calls
// GetWorldRay, so every fixture row carries the offset the ray was built with.  The camera's basis
// is private in the reference; it is read here through the usual test-side access switch.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>

#define private public
#include "camera.h"
#undef private

extern "C" {

// cam_out: position(3) view(3) right(3) up(3) aspect half film_w film_h focal_length = 17 floats
void ref_win_lens_ray(const float pos[3], const float focal[3], float film_w, float film_h, unsigned seed, float sx, float sy,
                      float cam_out[17], float lens[2], float ray_o[3], float ray_d[3])
{
    Camera cam;
    cam.SetPositionAndFocalPoint(glm::vec3(pos[0], pos[1], pos[2]), glm::vec3(focal[0], focal[1], focal[2]));
    cam.SetFilmSize(film_w, film_h);
    cam.PreRender();
    srand(seed);
    const glm::vec2 l = glm::circularRand(0.14f);
    srand(seed);
    const Ray r = cam.GetWorldRay(glm::vec2(sx, sy));
    for (int i = 0; i < 3; i++) {
        cam_out[i] = cam.position[i];
        cam_out[3 + i] = cam.viewDirection[i];
        cam_out[6 + i] = cam.right[i];
        cam_out[9 + i] = cam.up[i];
        ray_o[i] = r.position[i];
        ray_d[i] = r.direction[i];
    }
    cam_out[12] = cam.aspectRatio;
    cam_out[13] = cam.halfAngle;
    cam_out[14] = cam.filmWidth;
    cam_out[15] = cam.filmHeight;
    cam_out[16] = glm::length(cam.focalPoint - cam.position);
    lens[0] = l.x;
    lens[1] = l.y;
}

} // extern "C"
