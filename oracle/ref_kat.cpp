// Known-answer dumper for the reference's primitive tests.  TEST INFRASTRUCTURE ONLY.
//
// This file is this repository's own code: it includes the reference's headers where they lie
// (-I$(REF)/Code) and is linked against the reference's own, unmodified sources (oracle/Makefile, target
// `kat`), so every number it prints was computed by the reference's functions:
//   Sphere::intersect  sphere.cpp:118-174      Cube::intersect  cube.cpp:126-292
//   Plane::intersect   plane.cpp:143-201       AABB::intersect  aabb.cpp:31-63
//   Sphere/Cube/Plane::getAABB  sphere.cpp:177-188, cube.cpp:354-386, plane.cpp:324-340
// Inputs come from a small counter-based generator (not rand(): the reference never seeds it and this must not
// depend on libc).  Output: a binary stream of float32 records, `kind, n_in, n_out, in..., out...`, that
// tests/golden/make_golden.py stores as a fixture and tests/test_oracle.py replays through the C restatement.
#include "sphere.h"
#include "cube.h"
#include "plane.h"
#include "aabb.h"
#include "ray.h"
#include "config.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {

uint64_t g_state = 0x9E3779B97F4A7C15ull;
uint32_t next_u32() {                       // splitmix64, upper half
    uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    return (uint32_t)(z >> 32);
}
float uni(float lo, float hi) { return lo + (hi - lo) * (float)(next_u32() >> 8) * (1.0f / 16777216.0f); }

void emit(FILE* f, int kind, const std::vector<float>& in, const std::vector<float>& out) {
    float head[3] = { (float)kind, (float)in.size(), (float)out.size() };
    fwrite(head, 4, 3, f); fwrite(in.data(), 4, in.size(), f); fwrite(out.data(), 4, out.size(), f);
}

// A ray aimed at (or near) a target point so that a useful share of the tests hit.
Ray make_ray(const float target[3], float spread, std::vector<float>& in) {
    Ray r;
    r.origin = { uni(-8, 8), uni(-8, 8), uni(-8, 8) };
    float d[3] = { target[0] + uni(-spread, spread) - r.origin[0], target[1] + uni(-spread, spread) - r.origin[1],
                   target[2] + uni(-spread, spread) - r.origin[2] };
    uint32_t mode = next_u32() % 16;
    if (mode == 0) d[next_u32() % 3] = 0.0f;                     // axis-parallel component
    if (mode == 1) for (float& v : d) v = -v;                    // pointing away
    float len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (!(len > 0.0f)) { d[0] = 1.0f; d[1] = d[2] = 0.0f; len = 1.0f; }
    if (mode != 2) for (float& v : d) v /= len;                  // mode 2: unnormalised direction
    r.direction = { d[0], d[1], d[2] };
    r.time = (next_u32() & 1) ? uni(0, 1) : 0.0f;
    for (float v : r.origin) in.push_back(v);
    for (float v : r.direction) in.push_back(v);
    in.push_back(r.time);
    return r;
}

void hit_out(bool hit, const HitStructure& hs, bool tex, std::vector<float>& out) {
    out.push_back(hit ? 1.0f : 0.0f);
    if (!hit) { for (int i = 0; i < 9; ++i) out.push_back(0.0f); return; }
    out.push_back(hs.rayDistance);
    for (int i = 0; i < 3; ++i) out.push_back(hs.hitPoint[i]);
    for (int i = 0; i < 3; ++i) out.push_back(hs.normal[i]);
    out.push_back(tex ? hs.u : 0.0f); out.push_back(tex ? hs.v : 0.0f);
}

void set_material(Shape& s, bool tex) {
    s.diffuse = { 0.5f, 0.5f, 0.5f }; s.specular = { 0.5f, 0.5f, 0.5f };
    s.shininess = 32.0f; s.transparency = 0.0f; s.ior = 1.0f; s.hasTex = tex; s.texture = tex ? "kat.ppm" : "";
}

void box_out(const AABB& b, std::vector<float>& out) { for (float v : b.min) out.push_back(v); for (float v : b.max) out.push_back(v); }

} // namespace

int main(int argc, char** argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 2000;
    FILE* f = stdout;
    for (int i = 0; i < n; ++i) {                                 // kind 0: sphere
        std::vector<float> in, out;
        Sphere s; const bool tex = next_u32() & 1; set_material(s, tex);
        s.startLocation = { uni(-4, 4), uni(-4, 4), uni(-4, 4) };
        s.endLocation = { s.startLocation[0] + uni(-2, 2), s.startLocation[1] + uni(-2, 2), s.startLocation[2] + uni(-2, 2) };
        s.radius = uni(0.05f, 2.5f);
        Config cfg; cfg.motionBlur = next_u32() & 1; cfg.textures = tex;
        float tgt[3] = { s.startLocation[0], s.startLocation[1], s.startLocation[2] };
        Ray r = make_ray(tgt, s.radius * 1.5f, in);
        for (float v : s.startLocation) in.push_back(v);
        for (float v : s.endLocation) in.push_back(v);
        in.push_back(s.radius); in.push_back(cfg.motionBlur ? 1.0f : 0.0f); in.push_back(tex ? 1.0f : 0.0f);
        HitStructure hs; bool hit = s.intersect(r, hs, cfg);
        hit_out(hit, hs, tex, out); box_out(s.getAABB(), out);
        emit(f, 0, in, out);
    }
    for (int i = 0; i < n; ++i) {                                 // kind 1: cube
        std::vector<float> in, out;
        Cube c; const bool tex = next_u32() & 1; set_material(c, tex);
        c.startLocation = { uni(-4, 4), uni(-4, 4), uni(-4, 4) };
        c.endLocation = { c.startLocation[0] + uni(-2, 2), c.startLocation[1] + uni(-2, 2), c.startLocation[2] + uni(-2, 2) };
        const uint32_t rm = next_u32() % 4;
        c.rotation = { rm == 0 ? 0.0f : uni(-6.3f, 6.3f), rm <= 1 ? 0.0f : uni(-6.3f, 6.3f), rm == 0 ? 0.0f : uni(-6.3f, 6.3f) };
        c.scale = uni(0.05f, 2.0f);
        Config cfg; cfg.motionBlur = next_u32() & 1; cfg.textures = tex;
        float tgt[3] = { c.startLocation[0], c.startLocation[1], c.startLocation[2] };
        Ray r = make_ray(tgt, c.scale * 2.0f, in);
        for (float v : c.startLocation) in.push_back(v);
        for (float v : c.endLocation) in.push_back(v);
        for (float v : c.rotation) in.push_back(v);
        in.push_back(c.scale); in.push_back(cfg.motionBlur ? 1.0f : 0.0f); in.push_back(tex ? 1.0f : 0.0f);
        HitStructure hs; bool hit = c.intersect(r, hs, cfg);
        hit_out(hit, hs, tex, out); box_out(c.getAABB(), out);
        emit(f, 1, in, out);
    }
    for (int i = 0; i < n; ++i) {                                 // kind 2: plane (a fresh object per test, as a BVH leaf copy is)
        std::vector<float> in, out;
        Plane p; const bool tex = next_u32() & 1; set_material(p, tex);
        // a planar quadrilateral: centre + two in-plane axes, corners listed in a scrambled order like Blender's
        float c0[3] = { uni(-4, 4), uni(-4, 4), uni(-4, 4) };
        float a[3] = { uni(-1, 1), uni(-1, 1), uni(-1, 1) }, b[3] = { uni(-1, 1), uni(-1, 1), uni(-1, 1) };
        if (next_u32() % 4 == 0) { a[0] = uni(0.5f, 3); a[1] = a[2] = 0; b[1] = uni(0.5f, 3); b[0] = b[2] = 0; }   // axis-aligned floor
        const float sx[4] = { -1, 1, -1, 1 }, sy[4] = { -1, -1, 1, 1 };
        int order[4] = { 0, 1, 2, 3 };
        for (int k = 3; k > 0; --k) { int j = (int)(next_u32() % (uint32_t)(k + 1)); int t = order[k]; order[k] = order[j]; order[j] = t; }
        for (int k = 0; k < 4; ++k) {
            int q = order[k];
            p.vertices.push_back({ c0[0] + sx[q] * a[0] + sy[q] * b[0], c0[1] + sx[q] * a[1] + sy[q] * b[1], c0[2] + sx[q] * a[2] + sy[q] * b[2] });
        }
        p.calculateNormal();                                      // as parsePlaneDataFromJson does, plane.cpp:107
        Config cfg; cfg.textures = tex;
        Ray r = make_ray(c0, 2.5f, in);
        for (const auto& v : p.vertices) for (float x : v) in.push_back(x);
        in.push_back(tex ? 1.0f : 0.0f);
        AABB box = p.getAABB();                                   // before intersect() reorders the vertices
        HitStructure hs; bool hit = p.intersect(r, hs, cfg);
        hit_out(hit, hs, tex, out);
        for (float v : p.normal) out.push_back(v);
        box_out(box, out);
        emit(f, 2, in, out);
    }
    for (int i = 0; i < n; ++i) {                                 // kind 3: box slab test
        std::vector<float> in, out;
        float c0[3] = { uni(-4, 4), uni(-4, 4), uni(-4, 4) }, h[3] = { uni(0, 2), uni(0, 2), uni(0, 2) };
        if (next_u32() % 8 == 0) h[next_u32() % 3] = 0.0f;        // flat box (a plane's)
        Ray r = make_ray(c0, 2.5f, in);
        AABB box({ c0[0] - h[0], c0[1] - h[1], c0[2] - h[2] }, { c0[0] + h[0], c0[1] + h[1], c0[2] + h[2] });
        box_out(box, in);
        float tmin = 0.0f, tmax = 0.0f;
        bool hit = box.intersect(r, tmin, tmax);
        out.push_back(hit ? 1.0f : 0.0f); out.push_back(tmin); out.push_back(tmax);
        emit(f, 3, in, out);
    }
    return 0;
}
