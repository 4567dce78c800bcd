// Per-frame constants of Camera::convertPixelToRay (camera.cpp:115-153), hoisted out of the per-ray path.
// Host-only, header-only; the arithmetic follows the reference step by step (float products, the vector
// length through double squares and a double sqrt, then float divisions) so that the basis is bit-identical
// to what the reference recomputes for every ray.
#ifndef RT_CAMERA_BASIS_H
#define RT_CAMERA_BASIS_H

#include <cmath>

#include "rt_api.h"

struct rt_camera_basis {
    float fwd[3], right[3], up[3];
    float sw_over_fl, sh_over_fl;     // sensor_width / focal_length, sensor_height / focal_length
    float lens_radius;                // focal_length / (2 * aperture), 0 when aperture <= 0
};

// Raytracer::normalise, raytracer.cpp:577-585: pow(x,2) promotes to double, sqrt stays double, result to float.
inline void rt_normalise_ref(float v[3]) {
    float len = (float)std::sqrt((double)v[0] * (double)v[0] + (double)v[1] * (double)v[1] + (double)v[2] * (double)v[2]);
    v[0] /= len; v[1] /= len; v[2] /= len;
}

inline rt_camera_basis rt_make_camera_basis(const rt_camera& c) {
    rt_camera_basis b{};
    float f[3] = { c.gaze[0], c.gaze[1], c.gaze[2] };
    rt_normalise_ref(f);                                              // camera.cpp:124
    const float wup[3] = { 0.0f, 0.0f, 1.0f };
    float r[3] = { f[1] * wup[2] - f[2] * wup[1], f[2] * wup[0] - f[0] * wup[2], f[0] * wup[1] - f[1] * wup[0] };   // camera.cpp:130-134
    float rlen = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (rlen < 1e-6f) {                                               // camera.cpp:138-144
        r[0] = wup[1] * f[2] - wup[2] * f[1];
        r[1] = wup[2] * f[0] - wup[0] * f[2];
        r[2] = wup[0] * f[1] - wup[1] * f[0];
    }
    rt_normalise_ref(r);                                              // camera.cpp:146
    float u[3] = { f[1] * r[2] - f[2] * r[1], f[2] * r[0] - f[0] * r[2], f[0] * r[1] - f[1] * r[0] };               // camera.cpp:149-153
    for (int i = 0; i < 3; ++i) { b.fwd[i] = f[i]; b.right[i] = r[i]; b.up[i] = u[i]; }
    b.sw_over_fl = c.sensor_width / c.focal_length;                   // camera.cpp:119
    b.sh_over_fl = c.sensor_height / c.focal_length;                  // camera.cpp:120
    b.lens_radius = c.aperture > 0.0f ? c.focal_length / (2.0f * c.aperture) : 0.0f;   // camera.cpp:182
    return b;
}

#endif
