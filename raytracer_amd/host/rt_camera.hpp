// rt_camera.hpp -- struct Camera with the reference's fields, constructor arguments and
// GetWorldRay (raytracer_cpp/camera.h:4-63).  The constructor defers to the C ABI's rt_camera_init
// so that the host class and the device see the same bits.
#pragma once
#include "rt_api.h"
#include "rt_vec.hpp"

struct Camera {
    vec3 position;
    vec3 viewDirection;
    vec3 right;
    vec3 up;
    float aspectRatio;
    float halfAngle;
    float imageWidth;
    float imageHeight;

    Camera(const vec3 &cameraPosition, const vec3 &dir, float fov, int width, unsigned int height)
    {
        const float p[3] = {cameraPosition.x, cameraPosition.y, cameraPosition.z};
        const float d[3] = {dir.x, dir.y, dir.z};
        rt_camera c;
        rt_camera_init(&c, p, d, fov, width, (int32_t)height);
        position = vec3(c.position[0], c.position[1], c.position[2]);
        viewDirection = vec3(c.view_direction[0], c.view_direction[1], c.view_direction[2]);
        right = vec3(c.right[0], c.right[1], c.right[2]);
        up = vec3(c.up[0], c.up[1], c.up[2]);
        aspectRatio = c.aspect_ratio;
        halfAngle = c.half_angle;
        imageWidth = c.image_width;
        imageHeight = c.image_height;
    }

    vec3 GetWorldRay(const vec2 &imageSample) const
    {
        const float x = ((imageSample.x * 2.0f) / imageWidth) - 1.0f;
        const float y = ((imageSample.y * 2.0f) / imageHeight) - 1.0f;
        vec3 dir = viewDirection;
        dir += right * (halfAngle * aspectRatio * x);
        dir -= up * (halfAngle * y);
        return normalize(dir);
    }

    rt_camera Flatten() const
    {
        rt_camera c;
        c.position[0] = position.x; c.position[1] = position.y; c.position[2] = position.z;
        c.view_direction[0] = viewDirection.x; c.view_direction[1] = viewDirection.y; c.view_direction[2] = viewDirection.z;
        c.right[0] = right.x; c.right[1] = right.y; c.right[2] = right.z;
        c.up[0] = up.x; c.up[1] = up.y; c.up[2] = up.z;
        c.aspect_ratio = aspectRatio;
        c.half_angle = halfAngle;
        c.image_width = imageWidth;
        c.image_height = imageHeight;
        return c;
    }
};
