// gfx950 kernels of the render path: extend (primary-ray generation fused in at level 0), shadow, shade/resolve,
// finalize, tile assembly.  One lane per ray / per hit; 64-lane wavefronts; work items are compacted in HBM
// queues between kernels so that every launched lane has work.
//
// Arithmetic contract (build with -ffp-contract=off, IEEE divide/sqrt): every expression keeps the operand
// types and the association the reference's C++ has, so that discrete decisions (which primitive is hit,
// whether a slab test passes, int(255*c)) agree with it.  Reference citations are relative to its Code/ dir.
#include "rt_types.h"

#include <float.h>
#include <math.h>

// This file is compiled once per PART (see the Makefile): every part holds the shared device code and instantiates one
// family of kernels together with its launcher, so the parts build side by side.
//   RT_PART_KIND 0: shade / finalize / assemble / sort keys / frame start, and the public launchers that pick a part
//                1: k_extend   2: k_primary_shaded   3: k_shadow          RT_PART_SUB: the kernels' SUB argument (0, 1, 2)
#ifndef RT_PART_KIND
#error "compile with -DRT_PART_KIND=<0..3> -DRT_PART_SUB=<0..2>"
#endif
#ifndef RT_PART_SUB
#define RT_PART_SUB 0
#endif
#define RT_CAT2(a, b) a##b
#define RT_CAT(a, b) RT_CAT2(a, b)

// Development builds (-DRT_WALK_STATS): counts of the walk's steps, per lane and per wavefront, added to
// DevScene::walk_stats[k] / [16 + k] (read back with rt_debug_walk_stats).  Expands to nothing otherwise.
#ifdef RT_WALK_STATS
#define RT_WS(sc, k) do { if ((sc).walk_stats) { const unsigned long long m__ = __ballot(1); \
    if ((int)__lane_id() == __ffsll((long long)m__) - 1) { atomicAdd(&(sc).walk_stats[k], (unsigned long long)__popcll(m__)); atomicAdd(&(sc).walk_stats[16 + (k)], 1ull); } } } while (0)
#else
#define RT_WS(sc, k) do { } while (0)
#endif
enum { WS_BOX = 0, WS_GROUP = 1, WS_STEP = 2, WS_PLANE = 3, WS_CUBE = 4, WS_CUBE_EXACT = 5, WS_SPHERE = 6, WS_RETRY = 7, WS_RUN = 8, WS_TRACE = 9 };

#define DEVI __device__ __forceinline__
#define RL_FN __device__ __forceinline__
#include "ref_libm.h"

// minimum wavefronts per SIMD the register allocator must leave room for (second __launch_bounds__ argument)
#ifndef RT_EXTEND_WAVES
#define RT_EXTEND_WAVES 4           // scenes without second-level hierarchies: a few spills, better co-residency
#endif
#ifndef RT_EXTEND_WAVES_SUB
#define RT_EXTEND_WAVES_SUB 3       // 168 registers: the walk plus the hit's shading state fit almost without spills (4: 45-85 spilled, 10 % slower; 5: 30 % slower)
#endif
#ifndef RT_SHADOW_WAVES
#define RT_SHADOW_WAVES 5           // each mode's kernel fits 96 registers without spilling
#endif
#ifndef RT_SHADOW_WAVES_SUB
#define RT_SHADOW_WAVES_SUB 3       // no spills at 132-144 registers
#endif

#ifndef RT_REFILL_LANES
#define RT_REFILL_LANES 16          // k_shadow's drain: lanes that must be waiting before a wavefront refills / walks the first level
#endif
#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 3            // 152 registers as the compiler allots them
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace {

// ---------------------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------------------
DEVI float smax(float a, float b) { return (a < b) ? b : a; }        // std::max: NaN in `a` survives
DEVI float smin(float a, float b) { return (b < a) ? b : a; }        // std::min
DEVI int f2i(float f) { return __float_as_int(f); }
DEVI float i2f(int i) { return __int_as_float(i); }

struct V3 { float x, y, z; };
DEVI V3 mk(float x, float y, float z) { V3 v; v.x = x; v.y = y; v.z = z; return v; }
DEVI V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DEVI V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DEVI V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
DEVI float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                    // raytracer.cpp:588-592
DEVI V3 cross(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
// raytracer.cpp:577-585: squares and their sum in double, double sqrt, length rounded to float, float divides
DEVI float len_ref(V3 v) { return (float)sqrt((double)v.x * (double)v.x + (double)v.y * (double)v.y + (double)v.z * (double)v.z); }
DEVI V3 normalise(V3 v) { float l = len_ref(v); return mk(v.x / l, v.y / l, v.z / l); }
// Rays whose direction is drawn at random (soft-shadow samples) are judged by their distribution, not bit for bit:
// their tests run the same algorithm with the same thresholds on the hardware's reciprocal / reciprocal square root
// (1 ulp) instead of IEEE division and the double-precision length.  FAST = false is the reference's arithmetic.
template <bool FAST> DEVI float qdiv(float a, float b) { return FAST ? a * __builtin_amdgcn_rcpf(b) : a / b; }
template <bool FAST> DEVI V3 normalise_q(V3 v) {
    if (FAST) {
        float l2 = dot(v, v);
        if (l2 > 1e-30f && l2 < 1e30f) return v * __builtin_amdgcn_rsqf(l2);   // anything else takes the exact route
    }
    return normalise(v);
}
DEVI float lerp_ref(float a, float b, float t) { return a + t * (b - a); }                   // raytracer.cpp:671-673
// std::pow(float,float) of the reference is glibc powf; a double-precision pow rounded once to float agrees
// with it except where powf itself is not correctly rounded.
DEVI float powf_ref(float x, float y) { return (float)pow((double)x, (double)y); }

struct Ray { V3 o, d; float time; };

// Counter-based random numbers: three decorrelated 32-bit values from a three-word counter, no state, any lane can
// draw any number.  This is the PCG3D hash (Jarzynski and Olano, "Hash Functions for GPU Rendering", JCGT 9(3), 2020):
// about twenty integer operations per call.  Counters are (pixel or path id mixed with the seed, sample index,
// stream), so a value is a pure function of what it is drawn for, whatever the rank count, chunking or scheduling.
DEVI uint3 pcg3d(uint32_t x, uint32_t y, uint32_t z) {
    x = x * 1664525u + 1013904223u; y = y * 1664525u + 1013904223u; z = z * 1664525u + 1013904223u;
    x += y * z; y += z * x; z += x * y;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16;
    x += y * z; y += z * x; z += x * y;
    return make_uint3(x, y, z);
}
DEVI float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }               // [0,1)
DEVI uint32_t mix32(uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; }
enum { STREAM_AA = 1, STREAM_DOF = 2, STREAM_MB = 3, STREAM_SHADOW = 4, STREAM_GLOSSY = 5 };

// raytracer.cpp:634-644 on three uniforms: theta = 2 pi u, phi = acos(2v-1), r = cbrt(w).  Sampling code, judged by
// its distribution: cos(phi) = 2v-1 and sin(phi) = sqrt(1-cos^2) replace the acos/sin/cos round trip, sin/cos(2 pi u)
// come from the hardware instructions that take their argument in revolutions, cbrt is exp2(log2(w)/3).
DEVI V3 unit_sphere(float u, float v, float w) {
    float ct = __builtin_amdgcn_cosf(u), st = __builtin_amdgcn_sinf(u);
    float cp = 2.0f * v - 1.0f;
    float sp = __builtin_amdgcn_sqrtf(fmaxf(0.0f, 1.0f - cp * cp));
    float r = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(w) * (1.0f / 3.0f));
    return mk(r * sp * ct, r * sp * st, r * cp);
}

// raytracer.cpp:647-668: z and phi come from 1000-level quantised draws; basis from (0,1,0) x dir
DEVI V3 cone_direction(V3 dir, float angle, uint32_t r0, uint32_t r1) {
    float cosTheta = cosf(angle);
    float z = cosTheta + (1.0f - cosTheta) * ((float)(int)((r0 >> 1) % 1000u) / 1000.0f);
    float rev = (float)(int)((r1 >> 1) % 1000u) / 1000.0f;             // phi = 2 pi k / 1000, taken in revolutions
    float r = sqrtf(1.0f - z * z);
    V3 rv = mk(r * __builtin_amdgcn_cosf(rev), r * __builtin_amdgcn_sinf(rev), z);
    V3 u = normalise(cross(mk(0.0f, 1.0f, 0.0f), dir));
    V3 v = cross(dir, u);
    V3 s = mk(rv.x * u.x + rv.y * v.x + rv.z * dir.x, rv.x * u.y + rv.y * v.y + rv.z * dir.y, rv.x * u.z + rv.y * v.z + rv.z * dir.z);
    return normalise(s);
}

// ---------------------------------------------------------------------------------------------------------
// primitive tests.  Each returns whether the primitive's intersect() would return true and, if so, the
// rayDistance it would report.  FULL additionally produces the HitStructure fields the shader needs.
// ---------------------------------------------------------------------------------------------------------
struct HitOut { V3 p, n; float u, v; int gid; };

// sphere.cpp:118-174
template <bool FULL, bool FAST = false>
DEVI bool sphere_test(const float4* __restrict__ rec, const Ray& r, bool mb, bool want_uv, float& t_out, HitOut* h) {
    float4 w0 = rec[0], w1 = rec[1];
    V3 loc = mk(w0.x, w0.y, w0.z);
    if (mb) loc = mk(lerp_ref(w0.x, w1.x, r.time), lerp_ref(w0.y, w1.y, r.time), lerp_ref(w0.z, w1.z, r.time));
    float radius = w0.w;
    V3 l = loc - r.o;
    float tca = l.x * r.d.x + l.y * r.d.y + l.z * r.d.z;
    if (tca < 0) return false;
    float d2 = l.x * l.x + l.y * l.y + l.z * l.z - tca * tca;
    if (d2 > radius * radius) return false;
    float thc = FAST ? __builtin_amdgcn_sqrtf(radius * radius - d2) : sqrtf(radius * radius - d2);
    float t0 = tca - thc, t1 = tca + thc;
    float t = (t0 > 0) ? t0 : t1;
    if (t < 0) return false;
    t_out = t;
    if (FULL) {
        V3 ip = mk(r.o.x + t * r.d.x, r.o.y + t * r.d.y, r.o.z + t * r.d.z);
        h->p = ip;
        h->n = mk((ip.x - loc.x) / radius, (ip.y - loc.y) / radius, (ip.z - loc.z) / radius);
        h->gid = f2i(w1.w);
        if (want_uv) {                                                   // sphere.cpp:152-162
            V3 nn = normalise(ip - loc);
            // atan2 / asin on floats are the C library's single-precision routines, reproduced bit for bit (ref_libm.h)
            h->u = (float)((double)0.5f + (double)rl_atan2f(nn.z, nn.x) / (2 * M_PI));
            h->v = (float)((double)0.5f - (double)rl_asinf(nn.y, [](float t) { return sqrtf(t); }) / M_PI);
        }
    }
    return true;
}

// cube.cpp:297-318 with the sines/cosines taken from the record (computed on the host)
DEVI V3 rot_fwd(V3 v, float4 w2, float4 w3) {
    float cx = w2.x, sx = w2.y, cy = w2.z, sy = w2.w, cz = w3.x, sz = w3.y;
    float y1 = v.y * cx - v.z * sx, z1 = v.y * sx + v.z * cx;
    float x2 = v.x * cy + z1 * sy, z2 = -v.x * sy + z1 * cy;
    return mk(x2 * cz - y1 * sz, x2 * sz + y1 * cz, z2);
}
// cube.cpp:320-351
DEVI V3 rot_inv(V3 v, float4 w4, float4 w5) {
    float cz = w4.x, sz = w4.y, cy = w4.z, sy = w4.w, cx = w5.x, sx = w5.y;
    float x1 = v.x * cz - v.y * sz, y1 = v.x * sz + v.y * cz;
    float x2 = x1 * cy + v.z * sy, z2 = -x1 * sy + v.z * cy;
    return mk(x2, y1 * cx - z2 * sx, y1 * sx + z2 * cx);
}

// cube.cpp:126-292
template <bool FULL, bool FAST = false>
DEVI bool cube_test(const float4* __restrict__ rec, const Ray& r, bool mb, bool want_uv, float& t_out, HitOut* h) {
    float4 w0 = rec[0], w1 = rec[1], w2 = rec[2], w3 = rec[3], w4 = rec[4], w5 = rec[5];
    V3 loc = mk(w0.x, w0.y, w0.z);
    if (mb) loc = mk(lerp_ref(w0.x, w1.x, r.time), lerp_ref(w0.y, w1.y, r.time), lerp_ref(w0.z, w1.z, r.time));
    float scale = w0.w, inv = w1.w;
    if (!FULL) {
        // Cheap certain-miss test before the exact one: the line's distance from the cube centre against the
        // circumscribed sphere.  The exact test treats a direction component below 1e-5 as "inside that slab
        // forever", so a reported hit can lie outside the true cube by up to 1e-5 x distance travelled; the
        // radius is inflated for that and by 1 %, and the squared distance is given slack for cancellation.
        // NaNs fail the comparison and fall through to the exact test.
        V3 l = loc - r.o;
        float D2 = dot(l, l), tca = dot(l, r.d);
        float as = fabsf(scale);
        float R = as * 1.7495f + 2e-5f * (__builtin_amdgcn_sqrtf(D2) + 3.5f * as);
        if (D2 - tca * tca > R * R + 1e-5f * D2) return false;
    }
    V3 o = rot_inv(r.o - loc, w4, w5);
    V3 d = rot_inv(r.d, w4, w5);
    V3 lo = o * inv;
    V3 ld = normalise_q<FAST>(d * inv);
    float tmin = -INFINITY, tmax = INFINITY;
    {
        float lov[3] = { lo.x, lo.y, lo.z }, ldv[3] = { ld.x, ld.y, ld.z };
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (fabsf(ldv[i]) < 1e-5f) {
                if (lov[i] < -1.0f || lov[i] > 1.0f) return false;
            } else {
                float t1, t2;
                if (FAST) { float rc = __builtin_amdgcn_rcpf(ldv[i]); t1 = (-1.0f - lov[i]) * rc; t2 = (1.0f - lov[i]) * rc; }
                else { t1 = (-1.0f - lov[i]) / ldv[i]; t2 = (1.0f - lov[i]) / ldv[i]; }
                if (t1 > t2) { float tt = t1; t1 = t2; t2 = tt; }
                tmin = smax(tmin, t1);
                tmax = smin(tmax, t2);
                if (tmin > tmax) return false;
            }
        }
    }
    const float T_EPS = 1e-4f;
    float t = -1.0f;
    if (tmin > T_EPS) t = tmin; else if (tmax > T_EPS) t = tmax;
    if (t < 0.0f) return false;
    if (FAST && !FULL) {
        // the hit point's distance along a unit-length ray: the local parameter times the cube's size (the
        // rotation keeps lengths), which is what the reference's dot product below evaluates to up to rounding
        t_out = smax(t * fabsf(scale), 1e-4f);
        return true;
    }
    V3 hl = mk(lo.x + ld.x * t, lo.y + ld.y * t, lo.z + ld.z * t);
    V3 hw = rot_fwd(hl * scale, w2, w3) + loc;
    float worldT = dot(hw - r.o, r.d);
    worldT = smax(worldT, 1e-4f);
    t_out = worldT;
    if (FULL) {
        const float eps = 1e-5f;
        float ax = fabsf(hl.x), ay = fabsf(hl.y), az = fabsf(hl.z);
        V3 nl = mk(0.0f, 0.0f, 0.0f);
        if (ax >= ay - eps && ax >= az - eps) nl.x = (hl.x > 0) ? 1.0f : -1.0f;
        else if (ay >= ax - eps && ay >= az - eps) nl.y = (hl.y > 0) ? 1.0f : -1.0f;
        else nl.z = (hl.z > 0) ? 1.0f : -1.0f;
        h->p = hw;
        h->n = normalise(rot_fwd(nl, w2, w3));
        h->gid = f2i(w3.z);
        if (want_uv) {                                                   // cube.cpp:197-263
            float u, v;
            float maxA = smax(ax, smax(ay, az));
            if (ax >= maxA - eps) {
                if (hl.x > 0.0f) { u = (hl.z + 1.0f) * 0.5f; v = (hl.y + 1.0f) * 0.5f; }
                else { u = (1.0f - hl.z) * 0.5f; v = (hl.y + 1.0f) * 0.5f; }
            } else if (ay >= maxA - eps) {
                if (hl.y > 0.0f) { u = (hl.x + 1.0f) * 0.5f; v = (hl.z + 1.0f) * 0.5f; }
                else { u = (hl.x + 1.0f) * 0.5f; v = (1.0f - hl.z) * 0.5f; }
            } else {
                if (hl.z > 0.0f) { u = (hl.x + 1.0f) * 0.5f; v = (hl.y + 1.0f) * 0.5f; }
                else { u = (1.0f - hl.x) * 0.5f; v = (hl.y + 1.0f) * 0.5f; }
            }
            u = fmodf(u, 1.0f); v = fmodf(v, 1.0f);
            if (u < 0.0f) u += 1.0f;
            if (v < 0.0f) v += 1.0f;
            if (u < 0.0f) u = 0.0f;
            if (u > 1.0f) u = 1.0f;
            if (v < 0.0f) v = 0.0f;
            if (v > 1.0f) v = 1.0f;
            h->u = u; h->v = v;
        }
    }
    return true;
}

// plane.cpp:143-251.  The polygon's vertices were winding-sorted once on the host (the reference re-sorts them
// on every call with the same result); the vertex used for the plane offset is the one stored in the record.
template <bool FULL, bool FAST = false>
DEVI bool plane_test(const float4* __restrict__ rec, const float4* __restrict__ edges, const float4* __restrict__ plane_uv,
                     const Ray& r, bool want_uv, float& t_out, HitOut* h) {
    float4 w0 = rec[0], w1 = rec[1];
    float denom = w0.x * r.d.x + w0.y * r.d.y + w0.z * r.d.z;
    if ((double)fabsf(denom) < 1e-10) return false;
    float t = qdiv<FAST>(w0.x * (w1.x - r.o.x) + w0.y * (w1.y - r.o.y) + w0.z * (w1.z - r.o.z), denom);
    if (t < 0) return false;
    V3 ip = mk(r.o.x + t * r.d.x, r.o.y + t * r.d.y, r.o.z + t * r.d.z);
    int first = f2i(w0.w), packed = f2i(w1.w);
    int n = packed & 0xffff, drop = packed >> 16;
    // plane.cpp:205-251: even-odd crossing test in the projection that drops the dominant normal axis
    float Px = (drop == 0) ? ip.y : ip.x;
    float Py = (drop == 2) ? ip.y : ip.z;
    // one record per edge (vertex i and its predecessor j): (Pi.x, Pi.y, Pj.y, (Pj.x - Pi.x) / (Pj.y - Pi.y)), the
    // quotient formed on the host with the same IEEE division; NaN marks an edge the reference skips
    bool inside = false;
    for (int i = 0; i < n; ++i) {
        const float4 e = edges[first + i];
        if ((e.y > Py) != (e.z > Py)) {
            float xi = e.w * (Py - e.y) + e.x;
            if (Px < xi) inside = !inside;
        }
    }
    if (!inside) return false;
    t_out = t;
    if (FULL) {
        h->p = ip;
        h->n = mk(w0.x, w0.y, w0.z);
        int gid = f2i(rec[2].x);
        h->gid = gid;
        if (want_uv) {                                                   // plane.cpp:163-184 (axes and lengths from the host)
            float4 ua = plane_uv[3 * gid], va = plane_uv[3 * gid + 1], s0 = plane_uv[3 * gid + 2];
            V3 local = ip - mk(s0.x, s0.y, s0.z);
            float u = dot(local, mk(ua.x, ua.y, ua.z)) / ua.w;
            float v = dot(local, mk(va.x, va.y, va.z)) / va.w;
            h->u = fmodf(smax(u, 0.0f), 1.0f);
            h->v = fmodf(smax(v, 0.0f), 1.0f);
        }
    }
    return true;
}

// aabb.cpp:31-63: true divisions, ">= 0" picks the near/far plane (so -0.0 takes the positive branch), and the
// incoming range is ignored.
DEVI bool slab_exact(const Ray& r, float4 a, float4 b) {
    float tminx, tmaxx, tminy, tmaxy, tminz, tmaxz;
    if (r.d.x >= 0) { tminx = (a.x - r.o.x) / r.d.x; tmaxx = (b.x - r.o.x) / r.d.x; }
    else            { tminx = (b.x - r.o.x) / r.d.x; tmaxx = (a.x - r.o.x) / r.d.x; }
    if (r.d.y >= 0) { tminy = (a.y - r.o.y) / r.d.y; tmaxy = (b.y - r.o.y) / r.d.y; }
    else            { tminy = (b.y - r.o.y) / r.d.y; tmaxy = (a.y - r.o.y) / r.d.y; }
    if (r.d.z >= 0) { tminz = (a.z - r.o.z) / r.d.z; tmaxz = (b.z - r.o.z) / r.d.z; }
    else            { tminz = (b.z - r.o.z) / r.d.z; tmaxz = (a.z - r.o.z) / r.d.z; }
    float tMin = smax(tminx, smax(tminy, tminz));
    float tMax = smin(tmaxx, smin(tmaxy, tmaxz));
    return tMin <= tMax;
}

// Same decision as slab_exact at a third of the instructions: quotients by reciprocal-multiply (each within
// 2.5 ulp of the correctly rounded quotient), the decision taken only when the two bounds are further apart than
// that error allows; anything closer, and any zero / denormal / huge / NaN direction component, goes to slab_exact.
template <bool FAST>
DEVI bool slab(const Ray& r, float4 a, float4 b, V3 rcp, bool fast_ok) {
    if (FAST) {                                      // randomly drawn rays: the reciprocal-multiply decision stands
        float x0 = (a.x - r.o.x) * rcp.x, x1 = (b.x - r.o.x) * rcp.x;
        float y0 = (a.y - r.o.y) * rcp.y, y1 = (b.y - r.o.y) * rcp.y;
        float z0 = (a.z - r.o.z) * rcp.z, z1 = (b.z - r.o.z) * rcp.z;
        float tMin = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
        float tMax = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
        return tMin <= tMax;
    }
    if (fast_ok) {
        float x0 = (a.x - r.o.x) * rcp.x, x1 = (b.x - r.o.x) * rcp.x;
        float y0 = (a.y - r.o.y) * rcp.y, y1 = (b.y - r.o.y) * rcp.y;
        float z0 = (a.z - r.o.z) * rcp.z, z1 = (b.z - r.o.z) * rcp.z;
        float tMin = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
        float tMax = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
        float big = fmaxf(fmaxf(fmaxf(fabsf(x0), fabsf(x1)), fmaxf(fabsf(y0), fabsf(y1))), fmaxf(fabsf(z0), fabsf(z1)));
        float E = 1e-6f * big;                       // > 8 ulp of the largest quotient
        if (tMin + E < tMax - E) return true;
        if (tMin - E > tMax + E) return false;
    }
    return slab_exact(r, a, b);
}

// Result of a trace: class 0 plane / 1 cube / 2 sphere, index inside the set, which set it came from.
struct Best { float t; int cls; int idx; int fileset; };

// Closest hit over ranges of one primitive set, planes then cubes then spheres, strict '<' (first wins ties).
// leaf_rule: bvh.cpp:65-95 (start at +inf, require rayDistance >= 0); otherwise raytracer.cpp:202-247 (start at FLT_MAX).
// stop_below: the caller only asks whether the closest distance is below this value (soft shadows compare it with the
// distance to the light sample), so the search ends at the first candidate that is; -inf = full search.
template <bool ANY, bool FAST>
DEVI bool test_prims(const DevPrimSet& S, const Ray& r, bool mb, int p0, int pn, int c0, int cn, int s0, int sn,
                     bool leaf_rule, float stop_below, Best& best) {
    float closest = leaf_rule ? INFINITY : FLT_MAX;
    bool hit = false;
    for (int i = p0; i < p0 + pn; ++i) {
        float t;
        if (plane_test<false, FAST>(S.planes + 3 * i, S.edges, nullptr, r, false, t, nullptr) && (!leaf_rule || t >= 0.0f) && t < closest) {
            hit = true; closest = t; best.t = t; best.cls = 0; best.idx = i;
            if (ANY || t < stop_below) return true;
        }
    }
    for (int i = c0; i < c0 + cn; ++i) {
        float t;
        if (cube_test<false, FAST>(S.cubes + 6 * i, r, mb, false, t, nullptr) && (!leaf_rule || t >= 0.0f) && t < closest) {
            hit = true; closest = t; best.t = t; best.cls = 1; best.idx = i;
            if (ANY || t < stop_below) return true;
        }
    }
    for (int i = s0; i < s0 + sn; ++i) {
        float t;
        if (sphere_test<false, FAST>(S.spheres + 2 * i, r, mb, false, t, nullptr) && (!leaf_rule || t >= 0.0f) && t < closest) {
            hit = true; closest = t; best.t = t; best.cls = 2; best.idx = i;
            if (ANY || t < stop_below) return true;
        }
    }
    return hit;
}

// ---------------------------------------------------------------------------------------------------------
// Second-level hierarchy.  The reference's tree stops at depth 8, so a leaf of a large scene holds thousands of
// primitives, and its leaf rule (and the brute-force rule) is "closest reported distance, the earlier primitive
// wins an exact tie".  That result does not depend on the order of testing, so groups of more than a few
// primitives get a conventional BVH over their TRUE bounds, walked front to back with an LDS stack:
//   * a candidate replaces the best one if it is nearer, or exactly as near and earlier in the reference's
//     testing order (planes, cubes, spheres; file order inside a class) - every run's record keeps that order per entry;
//   * a subtree is skipped only when the ray provably cannot report a hit in it that would win: its box,
//     enlarged by `pad`, is missed or entered beyond the best distance, all with slack for rounding.  `pad`
//     covers what a primitive test may report outside the primitive's true bounds (a cube test treats a
//     direction component below 1e-5 as zero, which moves the reported point by up to 1e-5 x distance).
// The builders make a perfect binary tree per group (rt_build.hip); what the walk reads is a 4-ary view of it
// (rt_build_wide_records): one record per node of every SECOND binary level, holding the boxes of that node's four
// grandchildren (or, at the bottom of a tree of odd depth, of its two children) plane by plane:
//   words 0-2 = min x / y / z of children 0..3, words 3-5 = max x / y / z of children 0..3      (128 bytes, one cache line)
// and, in scenes with moving primitives, words 8-13 = how far each of those planes moves until time 1 (256 bytes):
// primitives move on straight lines, so the interpolated box bounds them at the ray's time.  Records are numbered as a
// 4-ary heap per group (record c of the group sits at root + c, its children at 4c + 1 .. 4c + 4; the children of the
// last level are runs of up to RT_SUB_LEAF entries, run j of the group owning the entry slots 4 * (root + j) .. + 3).
// ---------------------------------------------------------------------------------------------------------
// Per-ray constants of the second-level walk.  A box is entered at max over axes of (near plane - origin) / d and left
// at min over axes of (far plane - origin) / d.  Which plane is the near one depends only on the sign of d, so the ray
// keeps, per axis, the BYTE OFFSET of its near and of its far plane word inside a record and loads exactly those: no
// select per box.  With the reciprocal and the products origin * reciprocal formed once per ray, a plane of four
// children costs four fused multiply-adds.  Conservative by construction, so that a box is only ever skipped when the
// exact arithmetic would skip it too:
//   * the origin is moved by `pad` away from the box on either side (what a primitive test may report outside the
//     primitive's true bounds);
//   * the near reciprocal is scaled by 1 - 4e-6 and the far one by 1 + 4e-6 (positive entry distances shrink, positive
//     exit distances grow, by more than the rounding of the reciprocal and of the multiply-add; negative ones only
//     matter through their sign, which scaling keeps);
//   * the constants carry 1e-6 of their own magnitude (the two products cancel when the origin lies near a plane).
struct SubRay { V3 rn, rf, cn, cf; unsigned on_x, on_y, on_z, of_x, of_y, of_z; };

DEVI void sub_axis(float o, float d, float pad, unsigned off_min, unsigned off_max, float& rn, float& rf, float& cn, float& cf, unsigned& on, unsigned& of) {
    const bool pos = d >= 0.0f;
    on = pos ? off_min : off_max; of = pos ? off_max : off_min;
    if (fabsf(d) > 1e-12f) {
        const float rcp = __builtin_amdgcn_rcpf(d);
        rn = rcp * 0.999996f; rf = rcp * 1.000004f;
        // d > 0: entered through the min plane, seen from origin + pad; left through the max plane, from origin - pad
        const float o_n = pos ? o + pad : o - pad, o_f = pos ? o - pad : o + pad;
        cn = -o_n * rn; cf = -o_f * rf;
        cn -= 1e-6f * fabsf(cn) + 1e-30f; cf += 1e-6f * fabsf(cf) + 1e-30f;
        if (!(cn == cn) || !(cf == cf)) { rn = rf = 0.0f; cn = -INFINITY; cf = INFINITY; }     // NaN origin or pad: no constraint
    } else {
        // zero, denormal-small or NaN component: this axis constrains nothing (a ray parallel to the slab and outside it
        // is then visited needlessly, never skipped wrongly)
        rn = rf = 0.0f; cn = -INFINITY; cf = INFINITY;
    }
}

DEVI SubRay make_sub_ray(const DevScene& sc, const Ray& r) {
    SubRay q;
    V3 dc = r.o - mk(sc.center[0], sc.center[1], sc.center[2]);
    float pad = 2e-5f * (__builtin_amdgcn_sqrtf(dot(dc, dc)) + sc.radius) + 1e-6f * sc.radius;
    if (!(pad >= 0.0f)) pad = INFINITY;          // NaN origin: nothing may be skipped
    sub_axis(r.o.x, r.d.x, pad, 0u, 48u, q.rn.x, q.rf.x, q.cn.x, q.cf.x, q.on_x, q.of_x);
    sub_axis(r.o.y, r.d.y, pad, 16u, 64u, q.rn.y, q.rf.y, q.cn.y, q.cf.y, q.on_y, q.of_y);
    sub_axis(r.o.z, r.d.z, pad, 32u, 80u, q.rn.z, q.rf.z, q.cn.z, q.cf.z, q.on_z, q.of_z);
    return q;
}

// max / min of three that ignore NaN operands (a non-finite plane times a zero reciprocal): one instruction each
DEVI float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
DEVI float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }

// Records of a group (rt_build_wide_records): c < n_rec = (4^levels - 1) / 3 are the 4-ary view of the binary tree,
// levels = ceil(depth / 2); the children of the last of those levels are the group's runs, whose records follow from
// n_rec: run r at n_rec + r, holding the boxes of its own (up to four) primitives, their entries (word 6) and their
// places in the reference's testing order (word 7).  So a run is searched like any other node - one load round, four
// box tests - and only the primitives whose box the ray enters before the best hit so far are tested, nearest box first.
// State of one ray's walk through one group's records, kept by the caller between steps.
struct SubWalk {
    unsigned c;                     // record inside the group that the next step reads
    int sp;                         // entries on this thread's stack
    int root;                       // the group's first record
    unsigned n_rec, first_last;     // first run record; first record of the last level above the runs
    bool pairs;                     // odd depth: a record of that level has two runs below it, not four
    float best_t, prune_t;          // closest reported distance so far; what a subtree must be entered below to be worth a visit
    unsigned best_order; bool has;  // the best hit's place in the reference's testing order; whether there is a hit at all
};

DEVI void sub_begin(SubWalk& s, int root, int depth, bool leaf_rule) {
    const int levels = (depth + 1) >> 1;                      // record levels above the runs: a step moves two binary levels down
    s.c = 0; s.sp = 0; s.root = root;
    s.n_rec = ((1u << (2 * levels)) - 1u) / 3u;
    s.first_last = ((1u << (2 * (levels - 1))) - 1u) / 3u;
    s.pairs = (depth & 1) != 0;
    s.best_t = leaf_rule ? INFINITY : FLT_MAX;
    // prune_t: the best distance so far - or, once a caller who only asks "is the closest distance below stop_below" has a
    // hit at or beyond that value, stop_below itself (nothing entered later than that can change the answer; which
    // primitive is reported is then not the caller's concern)
    s.prune_t = s.best_t;
    s.best_order = 0; s.has = false;
}

enum { SUB_CONTINUE = 0, SUB_EXHAUSTED = 1, SUB_STOPPED = 2 };   // SUB_STOPPED: a hit that ends the search (ANY, or below stop_below)

// First half of a step: the record the walk stands on - four boxes - against the ray.  key[k] = entry distance of child
// k (never below 0; NaN counts as 0) with the child number in the two lowest bits, or all ones for a child that is
// skipped; a child is skipped only on a certain miss (any NaN makes the comparison false and the child is visited) or
// when it is entered beyond what is still worth a visit.  Returns whether the record is a run's (its children are
// primitives: `ev` = their entries, `off` = the record's byte offset).
template <bool MOTION>
DEVI bool sub_boxes(const DevScene& sc, const SubWalk& s, const Ray& r, const SubRay& q, bool mb, unsigned key[4], int4& ev, unsigned& off) {
    // Scenes with moving primitives keep boxes per time; rays at time 0 - every secondary ray, and every ray of a frame
    // without motion blur, where primitives stay at their first location - read the first half of a record only.
    const float tm = (MOTION && mb) ? r.time : 0.0f;
    constexpr unsigned REC = MOTION ? 256u : 128u;
    const char* __restrict__ base = (const char*)sc.sub_nodes;
    RT_WS(sc, WS_STEP);
    const unsigned c = s.c;
    off = ((unsigned)s.root + c) * REC;                       // (the upload keeps every record below 4 GiB)
    const bool is_run = c >= s.n_rec;
    float4 nx = *(const float4*)(base + (off + q.on_x)), ny = *(const float4*)(base + (off + q.on_y)), nz = *(const float4*)(base + (off + q.on_z));
    float4 fx = *(const float4*)(base + (off + q.of_x)), fy = *(const float4*)(base + (off + q.of_y)), fz = *(const float4*)(base + (off + q.of_z));
    ev = make_int4(-1, -1, -1, -1);
    if (is_run) ev = *(const int4*)(base + (off + 96u));
    if (MOTION && tm != 0.0f) {
        // boxes per time: planes at time 0 plus time x (their change until time 1)
        const float4 dnx = *(const float4*)(base + (off + 128u + q.on_x)), dny = *(const float4*)(base + (off + 128u + q.on_y)), dnz = *(const float4*)(base + (off + 128u + q.on_z));
        const float4 dfx = *(const float4*)(base + (off + 128u + q.of_x)), dfy = *(const float4*)(base + (off + 128u + q.of_y)), dfz = *(const float4*)(base + (off + 128u + q.of_z));
#define RT_MOVE(p, dp) p.x = __builtin_fmaf(dp.x, tm, p.x); p.y = __builtin_fmaf(dp.y, tm, p.y); p.z = __builtin_fmaf(dp.z, tm, p.z); p.w = __builtin_fmaf(dp.w, tm, p.w)
        RT_MOVE(nx, dnx); RT_MOVE(ny, dny); RT_MOVE(nz, dnz); RT_MOVE(fx, dfx); RT_MOVE(fy, dfy); RT_MOVE(fz, dfz);
#undef RT_MOVE
    }
    const bool last = c >= s.first_last;                      // the children are runs (or, of a run, primitives)
#define RT_CHILD(k, m) do { \
        const float tn = max3f(__builtin_fmaf(nx.m, q.rn.x, q.cn.x), __builtin_fmaf(ny.m, q.rn.y, q.cn.y), __builtin_fmaf(nz.m, q.rn.z, q.cn.z)); \
        const float tf = min3f(__builtin_fmaf(fx.m, q.rf.x, q.cf.x), __builtin_fmaf(fy.m, q.rf.y, q.cf.y), __builtin_fmaf(fz.m, q.rf.z, q.cf.z)); \
        const float d = fmaxf(tn, 0.0f), t = fminf(tf, s.prune_t); \
        const bool go = !(d > t) && !(last && !is_run && s.pairs && k >= 2); \
        key[k] = go ? ((__float_as_uint(d) & ~3u) | (unsigned)k) : 0xffffffffu; } while (0)
    RT_CHILD(0, x); RT_CHILD(1, y); RT_CHILD(2, z); RT_CHILD(3, w);
#undef RT_CHILD
    return is_run;
}

// Next record of the walk: the stack's top, or none.
DEVI int sub_pop(SubWalk& s, const int* __restrict__ stack) {
    if (s.sp == 0) return SUB_EXHAUSTED;
    --s.sp; s.c = (unsigned)stack[s.sp * RT_BLOCK];
    return SUB_CONTINUE;
}

// Second half of a step on a record above the runs: the children worth a visit, nearest first.
DEVI int sub_descend(SubWalk& s, unsigned key[4], int* __restrict__ stack) {
#define RT_CSWAP(i, j) do { const unsigned lo = min(key[i], key[j]), hi = max(key[i], key[j]); key[i] = lo; key[j] = hi; } while (0)
    RT_CSWAP(0, 1); RT_CSWAP(2, 3); RT_CSWAP(0, 2); RT_CSWAP(1, 3); RT_CSWAP(1, 2);
#undef RT_CSWAP
    const unsigned c = s.c;
    const bool last = c >= s.first_last;
    const unsigned i0 = c - s.first_last;
    const unsigned kid0 = last ? s.n_rec + (s.pairs ? 2u * i0 : 4u * i0) : 4u * c + 1u;
    if (key[0] != 0xffffffffu) {
        if (key[3] != 0xffffffffu) { stack[s.sp * RT_BLOCK] = (int)(kid0 + (key[3] & 3u)); ++s.sp; }
        if (key[2] != 0xffffffffu) { stack[s.sp * RT_BLOCK] = (int)(kid0 + (key[2] & 3u)); ++s.sp; }
        if (key[1] != 0xffffffffu) { stack[s.sp * RT_BLOCK] = (int)(kid0 + (key[1] & 3u)); ++s.sp; }
        s.c = kid0 + (key[0] & 3u);
        return SUB_CONTINUE;
    }
    return sub_pop(s, stack);
}

// One step of the walk: one record - four boxes - and, for a run's record, the tests of its primitives.
template <bool ANY, bool FAST, bool MOTION>
DEVI int sub_step(const DevScene& sc, const DevPrimSet& S, SubWalk& s, const Ray& r, const SubRay& q, bool mb, bool leaf_rule,
                  float stop_below, int* __restrict__ stack, Best& best) {
    unsigned key[4]; int4 ev; unsigned off;
    const char* __restrict__ base = (const char*)sc.sub_nodes;
    if (!sub_boxes<MOTION>(sc, s, r, q, mb, key, ev, off)) return sub_descend(s, key, stack);
    // the run's primitives whose box is worth a visit, nearest box first (one copy of the primitive tests)
    RT_WS(sc, WS_RUN);
    for (;;) {
        const unsigned m = min(min(key[0], key[1]), min(key[2], key[3]));
        if (m == 0xffffffffu) break;
        if (__uint_as_float(m & ~3u) > s.prune_t) break;                       // a primitive tested meanwhile brought the bound down
        const unsigned k = m & 3u;
        key[0] = k == 0u ? 0xffffffffu : key[0]; key[1] = k == 1u ? 0xffffffffu : key[1];
        key[2] = k == 2u ? 0xffffffffu : key[2]; key[3] = k == 3u ? 0xffffffffu : key[3];
        const unsigned entry = (unsigned)(k == 0u ? ev.x : k == 1u ? ev.y : k == 2u ? ev.z : ev.w);
        if (entry == 0xffffffffu) continue;
        const int cls = (int)(entry >> 30), idx = (int)(entry & 0x3fffffffu);
        float t; bool ok;
        if (cls == 0) { RT_WS(sc, WS_PLANE); ok = plane_test<false, FAST>(S.planes + 3 * idx, S.edges, nullptr, r, false, t, nullptr); }
        else if (cls == 1) { RT_WS(sc, WS_CUBE); ok = cube_test<false, FAST>(S.cubes + 6 * idx, r, mb, false, t, nullptr); }
        else { RT_WS(sc, WS_SPHERE); ok = sphere_test<false, FAST>(S.spheres + 2 * idx, r, mb, false, t, nullptr); }
        if (ok && (!leaf_rule || t >= 0.0f) && !(t > s.best_t)) {
            // exact tie between two reported distances: the primitive the reference tests first wins (strict '<'
            // in its loops); word 7 of the run's record keeps every primitive's place in that order
            unsigned ord = 0;
            if (s.has && t == s.best_t) {
                const int4 ov = *(const int4*)(base + (off + 112u));
                ord = (unsigned)(k == 0u ? ov.x : k == 1u ? ov.y : k == 2u ? ov.z : ov.w);
                if (!(ord < s.best_order)) continue;
            } else if (!(t < s.best_t)) continue;                                // NaN
            else if (!ANY) {
                const int4 ov = *(const int4*)(base + (off + 112u));
                ord = (unsigned)(k == 0u ? ov.x : k == 1u ? ov.y : k == 2u ? ov.z : ov.w);
            }
            s.has = true; s.best_t = t; s.best_order = ord; best.t = t; best.cls = cls; best.idx = idx;
            if (ANY || t < stop_below) return SUB_STOPPED;
            s.prune_t = (stop_below > -INFINITY) ? stop_below : t;             // t >= stop_below here
        }
    }
    return sub_pop(s, stack);
}

template <bool ANY, bool FAST, bool MOTION>
DEVI bool sub_trace(const DevScene& sc, const DevPrimSet& S, int root, int depth, const Ray& r, const SubRay& q, bool mb, bool leaf_rule,
                    float stop_below, int* __restrict__ stack, Best& best) {
    SubWalk s;
    sub_begin(s, root, depth, leaf_rule);
    for (;;) {
        const int st = sub_step<ANY, FAST, MOTION>(sc, S, s, r, q, mb, leaf_rule, stop_below, stack, best);
        if (st != SUB_CONTINUE) return st == SUB_STOPPED || s.has;
    }
}

// Closest hit of a group of primitives: through its second-level hierarchy when it has one, else one by one.
// SUB = the scene has at least one second-level hierarchy (kernels for small scenes are compiled without that code):
// 1 = node records of 4 words, 2 = of 8 words (boxes per time, scenes with moving primitives).
template <bool ANY, int SUB, bool FAST>
DEVI bool test_group(const DevScene& sc, const DevPrimSet& S, const Ray& r, const SubRay& q, bool mb, int p0, int pn, int c0, int cn, int s0, int sn,
                     int sub_root, int sub_depth, bool leaf_rule, float stop_below, int* __restrict__ stack, Best& best) {
    if (SUB && sub_root >= 0) return sub_trace<ANY, FAST, SUB == 2>(sc, S, sub_root, sub_depth, r, q, mb, leaf_rule, stop_below, stack, best);
    return test_prims<ANY, FAST>(S, r, mb, p0, pn, c0, cn, s0, sn, leaf_rule, stop_below, best);
}

// BVHNode::intersect (bvh.cpp:35-105) or Raytracer::unacceleratedIntersection (raytracer.cpp:202-256).
// The reference visits both children of every node whose box is hit and keeps the LAST leaf hit it pushed;
// walking the nodes right child first and stopping at the first leaf that yields a hit gives the same answer.
// The node array is laid out in that order with skip links, so the walk needs no stack.
// with_fallback: raytracer.cpp:424,484 - a secondary ray that misses the BVH is retried over every primitive.
// stop_below (see test_prims) ends the search inside the one group whose result the reference would return: a leaf
// that yields any hit ends the reference's walk too, so nothing is skipped that could change "closest < stop_below".
// TOGETHER (see the first-level loop): for wavefronts of unrelated rays; primary rays, whose lanes meet the same leaves at
// the same time anyway, keep the plain loop (and its register allocation).
template <bool ANY, int SUB, bool FAST = false, bool TOGETHER = true>
DEVI bool trace(const DevScene& sc, const DevFrame& f, const float4* __restrict__ nodes, const int4* __restrict__ leaves,
                int* __restrict__ stack, const Ray& r, bool with_fallback, Best& best, float stop_below = -INFINITY) {
    const bool mb = f.motion_blur != 0;
    RT_WS(sc, WS_TRACE);
    SubRay q; q.rn = q.rf = q.cn = q.cf = mk(0.0f, 0.0f, 0.0f); q.on_x = q.on_y = q.on_z = q.of_x = q.of_y = q.of_z = 0u;
    if (SUB) q = make_sub_ray(sc, r);
    // the file-order set as its hierarchy reads it (records in hierarchy order): what best.idx of a file-order hit indexes
    const DevPrimSet& FS = SUB ? sc.walkset : sc.fileset;
    if (f.use_bvh) {
        int i = 0;
        const int n = sc.n_nodes;
        // reciprocal-multiply slab tests are allowed when every direction component is an ordinary number
        const float ax = fabsf(r.d.x), ay = fabsf(r.d.y), az = fabsf(r.d.z);
        const bool fast_ok = ax > 1e-18f && ax < 1e18f && ay > 1e-18f && ay < 1e18f && az > 1e-18f && az < 1e18f;
        const V3 rcp = mk(__builtin_amdgcn_rcpf(r.d.x), __builtin_amdgcn_rcpf(r.d.y), __builtin_amdgcn_rcpf(r.d.z));
        if (SUB && TOGETHER) {
            // Leaves cost a walk through their own hierarchy, boxes a few instructions.  Every lane first advances to the
            // next leaf whose box it pierces (lanes differ in how many boxes that takes), then the lanes search their
            // leaves side by side: a wavefront of unrelated rays makes as many searches as its busiest lane, not one per
            // position of the node array at which any of its lanes meets a leaf.
            for (;;) {
                int leaf = -1;
                while (i < n) {
                    const float4 a = nodes[2 * i], b = nodes[2 * i + 1];
                    RT_WS(sc, WS_BOX);
                    if (!slab<FAST>(r, a, b, rcp, fast_ok)) { i = f2i(a.w); continue; }
                    ++i;
                    leaf = f2i(b.w);
                    if (leaf >= 0) break;
                }
                if (leaf < 0) break;
                const int4 l0 = leaves[2 * leaf], l1 = leaves[2 * leaf + 1];
                RT_WS(sc, WS_GROUP);
                if (test_group<ANY, SUB, FAST>(sc, sc.leafset, r, q, mb, l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w, true, stop_below, stack, best)) { best.fileset = 0; return true; }
            }
        } else {
            while (i < n) {
                float4 a = nodes[2 * i], b = nodes[2 * i + 1];
                RT_WS(sc, WS_BOX);
                if (!slab<FAST>(r, a, b, rcp, fast_ok)) { i = f2i(a.w); continue; }
                int leaf = f2i(b.w);
                ++i;
                if (leaf >= 0) {
                    int4 l0 = leaves[2 * leaf], l1 = leaves[2 * leaf + 1];
                    RT_WS(sc, WS_GROUP);
                    if (test_group<ANY, SUB, FAST>(sc, sc.leafset, r, q, mb, l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w, true, stop_below, stack, best)) { best.fileset = 0; return true; }
                }
            }
        }
        if (!with_fallback) return false;
        RT_WS(sc, WS_RETRY);
        best.fileset = 1;
        return test_group<false, SUB, FAST>(sc, FS, r, q, mb, 0, FS.n_planes, 0, FS.n_cubes, 0, FS.n_spheres,
                                 sc.file_sub_root, sc.file_sub_depth, false, stop_below, stack, best);
    }
    best.fileset = 1;
    return test_group<ANY, SUB, FAST>(sc, FS, r, q, mb, 0, FS.n_planes, 0, FS.n_cubes, 0, FS.n_spheres,
                           sc.file_sub_root, sc.file_sub_depth, false, stop_below, stack, best);
}

template <int SUB>
DEVI void full_hit(const DevScene& sc, const DevFrame& f, const Ray& r, const Best& best, HitOut& h) {
    const DevPrimSet& S = best.fileset ? (SUB ? sc.walkset : sc.fileset) : sc.leafset;
    const bool mb = f.motion_blur != 0;
    float t;
    h.u = 0.0f; h.v = 0.0f;
    if (best.cls == 0) {
        int gid = f2i(S.planes[3 * best.idx + 2].x);
        bool uv = f.textures && f2i(sc.mats[3 * gid + 2].y) >= 0;
        plane_test<true>(S.planes + 3 * best.idx, S.edges, sc.plane_uv, r, uv, t, &h);
    } else if (best.cls == 1) {
        int gid = f2i(S.cubes[6 * best.idx + 3].z);
        bool uv = f.textures && f2i(sc.mats[3 * gid + 2].y) >= 0;
        cube_test<true>(S.cubes + 6 * best.idx, r, mb, uv, t, &h);
    } else {
        int gid = f2i(S.spheres[2 * best.idx + 1].w);
        bool uv = f.textures && f2i(sc.mats[3 * gid + 2].y) >= 0;
        sphere_test<true>(S.spheres + 2 * best.idx, r, mb, uv, t, &h);
    }
}

// Copy the BVH nodes and leaf records into LDS with coalesced 16-byte loads (one pass per workgroup).
DEVI void stage_bvh(const DevScene& sc, float4* s_nodes, int4* s_leaves, bool with_leaves) {
    for (int i = threadIdx.x; i < 2 * sc.n_nodes; i += blockDim.x) s_nodes[i] = sc.nodes[i];
    if (with_leaves) for (int i = threadIdx.x; i < 2 * sc.n_leaves; i += blockDim.x) s_leaves[i] = sc.leaves[i];
    __syncthreads();
}

// Tiny scenes (a handful of primitives, no second-level hierarchy): every workgroup also copies the primitive records
// of both orderings, the polygon vertices and the materials into LDS and redirects the scene's pointers there, so a
// primitive test costs LDS reads instead of a chain of global-memory round trips.  Words in LDS: rt_lds_prim_words().
DEVI void stage_prims(DevScene& sc, float4* s_prim) {
    int o = 0;
    auto put = [&](const float4* src, int words) -> const float4* {
        for (int i = threadIdx.x; i < words; i += blockDim.x) s_prim[o + i] = src[i];
        const float4* p = s_prim + o; o += words; return p;
    };
    const int n_mats = 3 * (sc.fileset.n_planes + sc.fileset.n_cubes + sc.fileset.n_spheres);
    const float4* edges = put(sc.fileset.edges, sc.n_verts);
    sc.fileset.planes = put(sc.fileset.planes, 3 * sc.fileset.n_planes);
    sc.fileset.cubes = put(sc.fileset.cubes, 6 * sc.fileset.n_cubes);
    sc.fileset.spheres = put(sc.fileset.spheres, 2 * sc.fileset.n_spheres);
    sc.leafset.planes = put(sc.leafset.planes, 3 * sc.leafset.n_planes);
    sc.leafset.cubes = put(sc.leafset.cubes, 6 * sc.leafset.n_cubes);
    sc.leafset.spheres = put(sc.leafset.spheres, 2 * sc.leafset.n_spheres);
    sc.mats = put(sc.mats, n_mats);
    sc.fileset.edges = edges; sc.leafset.edges = edges;
    __syncthreads();
}

// Sum a per-thread count over the workgroup and add it to a 64-bit statistic with one atomic.
DEVI void block_stat_add(unsigned long long* stat, unsigned long long v, unsigned long long* s_tmp) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_tmp[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (unsigned k = 0; k < (blockDim.x >> 6); ++k) t += s_tmp[k];
        if (t) atomicAdd(stat, t);
    }
}

// Pixel of slot `in` (0 .. RT_TILE_PIX - 1) inside its tile: the 64 slots of a wavefront cover an 8 x 8 block of pixels
// (blocks in row-major order), not two rows of 32 - neighbouring rays, hence neighbouring walks through large scenes.
DEVI int slot_col(int in) { return (((in >> 6) & 3) << 3) | (in & 7); }
DEVI int slot_row(int in) { return ((in >> 8) << 3) | ((in >> 3) & 7); }

// Camera::convertPixelToRay (camera.cpp:109-199) and the sample loops around it (raytracer.cpp:120-141).
// i = index inside the batch; one batch = one sample index for a range of this rank's pixel slots.
DEVI bool generate_primary(const DevFrame& f, int i, Ray& r, uint32_t& path, int& slot) {
    slot = f.pix_begin + i;
    int tile_local = slot / RT_TILE_PIX, in = slot % RT_TILE_PIX;
    // a wavefront's 64 slots lie in one tile (batches start at multiples of 64): one scalar load, which does not
    // queue behind the vector stores of the previous round
    int gt = f.local_tiles[__builtin_amdgcn_readfirstlane(tile_local)];
    int x = (gt % f.tiles_x) * RT_TILE + slot_col(in);
    int y = (gt / f.tiles_x) * RT_TILE + slot_row(in);
    if (x >= f.width || y >= f.height) return false;
    uint32_t pixel = (uint32_t)(y * f.width + x);
    int s = f.sample;
    int s_mb = s % f.mb_n, s_dof = (s / f.mb_n) % f.dof_n, s_aa = s / (f.mb_n * f.dof_n);
    float u = (float)x, v = (float)y;
    if (f.anti_aliasing) {                                              // raytracer.cpp:123-127
        uint3 q = pcg3d(pixel ^ (f.seed * 0x9E3779B1u), (uint32_t)s_aa, STREAM_AA);
        u = (float)x + u01(q.x);
        v = (float)y + u01(q.y);
    }
    float un = (u + 0.5f) / f.res_x;
    float vn = (v + 0.5f) / f.res_y;
    float cam_x = -(un - 0.5f) * f.sw_over_fl;
    float cam_y = (0.5f - vn) * f.sh_over_fl;
    float cam_z = -1.0f;
    V3 d = mk(cam_x * f.right[0] + cam_y * f.up[0] + cam_z * f.fwd[0],
              cam_x * f.right[1] + cam_y * f.up[1] + cam_z * f.fwd[1],
              cam_x * f.right[2] + cam_y * f.up[2] + cam_z * f.fwd[2]);
    float dlen = sqrtf(d.x * d.x + d.y * d.y + d.z * d.z);
    if (dlen < 1e-6f) d = mk(0.0f, 0.0f, 1.0f);
    d = normalise(d);
    d = mk(-d.x, -d.y, -d.z);
    V3 o = mk(f.cam_loc[0], f.cam_loc[1], f.cam_loc[2]);
    if (f.lens_radius > 0.0f) {                                         // camera.cpp:181-196, 286-312
        uint3 q = pcg3d(pixel ^ (f.seed * 0x9E3779B1u), (uint32_t)(s_aa * f.dof_n + s_dof), STREAM_DOF);
        float lx = 2.0f * u01(q.x) - 1.0f, ly = 2.0f * u01(q.y) - 1.0f;
        float sx = 0.0f, sy = 0.0f;
        if (!(lx == 0.0f && ly == 0.0f)) {
            // camera.cpp:299-311: theta = (pi/4)(y/x) or pi/2 - (pi/4)(x/y); here in revolutions for v_sin/v_cos
            float rev, rr;
            if (fabsf(lx) > fabsf(ly)) { rr = lx; rev = 0.125f * (ly / lx); }
            else { rr = ly; rev = 0.25f - 0.125f * (lx / ly); }
            sx = rr * __builtin_amdgcn_cosf(rev); sy = rr * __builtin_amdgcn_sinf(rev);
        }
        sx *= f.lens_radius; sy *= f.lens_radius;
        V3 right = mk(f.right[0], f.right[1], f.right[2]), up = mk(f.up[0], f.up[1], f.up[2]);
        V3 lens = right * sx + up * sy;
        V3 focus = o + d * f.focal_distance;
        o = o + lens;
        d = normalise(focus - o);
    }
    float time = 0.0f;
    if (f.motion_blur) {                                                // raytracer.cpp:141 (only read when -m is on)
        uint3 q = pcg3d(pixel ^ (f.seed * 0x9E3779B1u), (uint32_t)s, STREAM_MB);
        time = ((float)s_mb + u01(q.x)) / (float)f.mb_n;
    }
    r.o = o; r.d = d; r.time = time;
    path = mix32(pixel * 0x9E3779B1u ^ mix32((uint32_t)s * 0x85EBCA6Bu + f.seed));
    return true;
}

// ---------------------------------------------------------------------------------------------------------
// extend: trace one ray per lane to the hit the reference would report, append a hit-tree node, and spawn the
// reflection / refraction rays of the next level (raytracer.cpp:144-164 for primaries, :373-498 for the spawn).
// ---------------------------------------------------------------------------------------------------------
// STATS: the kernel keeps the six per-thread ray counts (twelve registers it otherwise has to spill)
template <bool PRIMARY, int SUB, bool LDSP, bool STATS>
__global__ __launch_bounds__(RT_BLOCK, SUB ? RT_EXTEND_WAVES_SUB : RT_EXTEND_WAVES) void k_extend(DevScene sc_arg, DevFrame f, DevWork w, int level) {
    extern __shared__ float4 s_mem[];
    DevScene sc = sc_arg;
    float4* s_nodes = s_mem;
    // with second-level hierarchies the LDS budget goes to the traversal stacks (occupancy): leaf records, read once
    // per leaf visit, then stay in global memory
    const int4* s_leaves = SUB ? sc.leaves : (const int4*)(s_mem + 2 * sc.n_nodes);
    int* s_stack = (int*)(s_mem + 2 * sc.n_nodes) + threadIdx.x;                      // [depth][thread], only with SUB
    __shared__ unsigned long long s_tmp[RT_BLOCK / 64];
    // The launch's work = chunks of RT_BLOCK rays: consecutive pixel slots for primary rays, consecutive queue slots
    // of one region otherwise.  Workgroup b takes chunks b, b + gridDim.x, ...
    __shared__ int s_pre[RT_REGIONS + 1], s_cnt[RT_REGIONS];           // first chunk and ray count of every region
    if (!PRIMARY) {
        if (threadIdx.x == 0) {
            int acc = 0, n[RT_REGIONS];
#pragma unroll
            for (int rg = 0; rg < RT_REGIONS; ++rg) n[rg] = w.ctr[RT_CTR_LINK(level, rg)];      // unused regions' counters stay 0
#pragma unroll
            for (int rg = 0; rg < RT_REGIONS; ++rg) {
                const int m = min(n[rg], w.link_rcap[level]);
                s_cnt[rg] = m; s_pre[rg] = acc; acc += (m + RT_BLOCK - 1) / RT_BLOCK;
            }
            s_pre[RT_REGIONS] = acc;
        }
        __syncthreads();
    }
    const int n_chunks = PRIMARY ? (f.pix_count + RT_BLOCK - 1) / RT_BLOCK : s_pre[RT_REGIONS];
    if ((int)blockIdx.x >= n_chunks) return;                           // whole workgroup idle
    if (f.use_bvh) stage_bvh(sc, s_nodes, (int4*)(s_mem + 2 * sc.n_nodes), !SUB);
    if (LDSP) stage_prims(sc, s_mem + 2 * sc.n_nodes + 2 * sc.n_leaves);

    const int link_in = PRIMARY ? 0 : w.link_off[level];
    const int node_base = w.node_off[level], node_cap = w.node_rcap[level];
    const int link_out = w.link_off[level + 1], link_cap = w.link_rcap[level + 1];
    // queue slot of this thread's ray in chunk g, or -1 past the end of the chunk's region (region returned in rg)
    int seg = 0;
    // (with w.perm: the k-th ray traced of a region is the one in slot perm[k of that region], see k_sort_keys)
    auto locate = [&](int g, int& rg) -> int {
        while (seg < w.regions - 1 && g >= __builtin_amdgcn_readfirstlane(s_pre[seg + 1])) ++seg;
        rg = seg;
        const int local = (g - __builtin_amdgcn_readfirstlane(s_pre[seg])) * RT_BLOCK + (int)threadIdx.x;
        if (!(local < __builtin_amdgcn_readfirstlane(s_cnt[seg]))) return -1;
        const int at = seg * w.link_rcap[level] + local;
        return (SUB && w.perm) ? (int)w.perm[at] : at;
    };
    const float4* __restrict__ qa_in = w.q_a[level & 1];
    const float4* __restrict__ qb_in = w.q_b[level & 1];
    float4* __restrict__ qa_out = w.q_a[(level + 1) & 1];
    float4* __restrict__ qb_out = w.q_b[(level + 1) & 1];
    const bool spawn = level < f.max_depth;
    unsigned long long n_primary = 0, n_refl = 0, n_refr = 0, n_shadow_ref = 0, n_refl_ref = 0, n_refr_ref = 0;
    const unsigned long long shadow_per_hit = (unsigned long long)sc.n_lights * (f.soft_shadows ? f.ss_samples : 1);

    // Order inside a round: loads first (this round's queue entry was fetched during the previous round), then
    // the trace and all arithmetic, then ONE returning atomic per wavefront that allocates the hit's node and its
    // children's queue slots together, then every store.  Loads, the atomic's return and stores share one in-order
    // counter, so anything that waits for a load also waits for all older stores: nothing is loaded after a store
    // of the same round, and only the atomic stands between the arithmetic and the stores.
    float4 qa_next = make_float4(0, 0, 0, 0), qb_next = qa_next;
    int q_next = -1, rg_next = 0;
    if (!PRIMARY) { q_next = locate((int)blockIdx.x, rg_next); if (q_next >= 0) { qa_next = qa_in[q_next]; qb_next = qb_in[q_next]; } }
    for (int g = (int)blockIdx.x; g < n_chunks; g += (int)gridDim.x) {
        const int i = PRIMARY ? g * RT_BLOCK + (int)threadIdx.x : q_next;        // pixel slot of the batch / queue slot
        const int rg = PRIMARY ? (g & (w.regions - 1)) : rg_next;                // region this round allocates in
        bool active = PRIMARY ? i < f.pix_count : i >= 0;
        Ray r; uint32_t path = 0, mult = 1; int slot = 0;
        bool replicated = false;            // an ancestor's reflection was already traced gr_samples times (soft shadows)
        int copies = 1;                     // identical hits of the reference this ray's node stands for (f.bundle)
        r.o = mk(0, 0, 0); r.d = mk(0, 0, 1); r.time = 0.0f;
        if (active) {
            if (PRIMARY) active = generate_primary(f, i, r, path, slot);
            else {
                float4 a = qa_next, b = qb_next;
                r.o = mk(a.x, a.y, a.z); r.d = mk(a.w, b.x, b.y); r.time = 0.0f;   // secondary rays carry time 0 (ray.h:13)
                path = (uint32_t)f2i(b.z); mult = (uint32_t)f2i(b.w) & 0x7fffffffu; replicated = (f2i(b.w) >> 31) & 1;
                if (f.bundle && replicated) copies = f.gr_samples;
            }
        }
        if (!PRIMARY && g + (int)gridDim.x < n_chunks) {
            q_next = locate(g + (int)gridDim.x, rg_next);
            if (q_next >= 0) { qa_next = qa_in[q_next]; qb_next = qb_in[q_next]; }
        }
        if (STATS && PRIMARY && active) ++n_primary;
        Best best; best.t = 0.0f; best.cls = 0; best.idx = 0; best.fileset = 0;
        bool hit = active && trace<false, SUB, false, !PRIMARY>(sc, f, s_nodes, s_leaves, s_stack, r, !PRIMARY && f.fallback_full, best);

        int n_children = 0, n_reflect = 0, n_entries = 0; bool refract = false;   // n_reflect: child colours summed at resolve; n_entries: rays queued for them
        HitOut h; V3 N = mk(0, 0, 1), rd = mk(0, 0, 1), fd = mk(0, 0, 1); float weight = 0.0f, gloss_angle = 0.0f; bool glossy = false;
        if (hit) {
            if (STATS) n_shadow_ref += shadow_per_hit * mult * (unsigned)copies;
            full_hit<SUB>(sc, f, r, best, h);
            if (spawn) {
                float4 m0 = sc.mats[3 * h.gid], m1 = sc.mats[3 * h.gid + 1], m2 = sc.mats[3 * h.gid + 2];
                float reflectivity = m2.z;                               // raytracer.cpp:374-375, evaluated on the host
                float ior = m2.x > 0.0f ? m2.x : 1.0f;
                float transparency = m1.w;
                V3 cam = mk(f.cam_loc[0], f.cam_loc[1], f.cam_loc[2]);
                V3 V = normalise(cam - h.p);                             // raytracer.cpp:381-382: always from the camera
                N = h.n;
                if (dot(r.d, N) > 0.0f) N = mk(-N.x, -N.y, -N.z);        // raytracer.cpp:386-390
                if (reflectivity > 0) {                                  // raytracer.cpp:395-413
                    float incoming = dot(r.d, N);
                    rd = normalise(mk(r.d.x - 2.0f * incoming * N.x, r.d.y - 2.0f * incoming * N.y, r.d.z - 2.0f * incoming * N.z));
                    glossy = f.glossy && m0.w > 0.0f;
                    if (glossy) gloss_angle = (float)((double)smax(0.01f, (1.0f - smin(1.0f, m0.w / 128.0f))) * (M_PI / 6.0));
                    // Without glossy perturbation all gr_samples rays are identical: trace one, replay the sum at resolve.
                    // With soft shadows the reference's identical rays still differ in what follows them: each hit draws
                    // its own shadow samples (raytracer.cpp:330-332 under :400-432) and is clamped on its own.  The first
                    // reflection of a path is therefore traced gr_samples times for real (distinct path ids, so distinct
                    // sample sets); below it, where a contribution carries reflectivity^2 or less, identical rays are
                    // traced once again (the reference would trace gr_samples^depth of them).
                    // f.bundle: the gr_samples rays are geometrically identical, so ONE queue entry is traced and its node
                    // stands for gr_samples copies (with their own path ids, hence their own shadow samples, shadow counts
                    // and colours); everything below a bundle stays a bundle of the same size, copy s under copy s.
                    n_reflect = (glossy || (f.replicate && !replicated)) ? f.gr_samples : 1;
                    n_entries = (f.bundle && !glossy) ? 1 : n_reflect;
                }
                if (transparency > 0) {                                  // raytracer.cpp:444-497
                    float n1 = 1.0f, n2 = ior;
                    float cos_i = dot(r.d, N);
                    if (cos_i > 0) { float tt = n1; n1 = n2; n2 = tt; }
                    float eta = n1 / n2;
                    cos_i = -smax(-1.0f, smin(1.0f, cos_i));
                    float k = 1.0f - eta * eta * (1.0f - cos_i * cos_i);
                    if (!(k < 0.0f)) {
                        float sq = sqrtf(k);
                        fd = normalise(mk((r.d.x * eta) + (N.x * (eta * cos_i - sq)), (r.d.y * eta) + (N.y * (eta * cos_i - sq)),
                                          (r.d.z * eta) + (N.z * (eta * cos_i - sq))));
                        refract = true;
                        float cosTheta = fabsf(dot(V, N));
                        float q = (n1 - n2) / (n1 + n2);
                        float r0 = (float)((double)q * (double)q);
                        float fresnel = (float)((double)r0 + (double)(1.0f - r0) * pow((double)(1.0f - cosTheta), 5.0));
                        weight = transparency * (1.0f - fresnel);
                    }
                }
                n_children = n_entries + (refract ? 1 : 0);
            }
        }

        // one node per hit and n_children queue slots, allocated per wavefront with one 64-bit atomic
        int node = -1, qslot = 0;
        {
            const int lane = __lane_id();
            const unsigned long long hm = __ballot(hit);
            int incl = n_children;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { int v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
            const int total_children = __shfl(incl, 63, 64);
            const int total_hits = __popcll(hm), before = __popcll(hm & ((1ull << lane) - 1ull));
            unsigned long long pair = 0;
            if (lane == 0 && (total_hits | total_children))
                pair = atomicAdd((unsigned long long*)&w.ctr[RT_CTR_NODE(level, rg)],
                                 ((unsigned long long)(unsigned)total_children << 32) | (unsigned long long)(unsigned)total_hits);
            const int base_node = __shfl((int)(unsigned)(pair & 0xffffffffull), 0, 64), base_link = __shfl((int)(unsigned)(pair >> 32), 0, 64);
            if (hit) node = base_node + before;
            qslot = base_link + incl - n_children;
        }
        if (hit && node + 1 > node_cap) { w.stats[RT_S_ERR] = 1; hit = false; }
        node = hit ? node_base + rg * node_cap + node : -1;
        if (!PRIMARY && active) w.link[link_in + i] = node;
        if (hit) {
            if (n_children > 0 && qslot + n_children > link_cap) {
                w.stats[RT_S_ERR] = 1; n_children = 0; n_reflect = 0; n_entries = 0; refract = false;
            }
            qslot += rg * link_cap;                                      // the region's part of the next level's queue
            const int first_link = link_out + qslot;
            // one node for all `copies` of the hit: what differs per copy (shadow counts, colours) lives in the node's slots
            {
                w.n_a[node] = make_float4(h.p.x, h.p.y, h.p.z, r.time);
                w.n_b[node] = make_float4(h.n.x, h.n.y, h.n.z, i2f(h.gid | (copies > 1 ? (int)0x80000000 : 0)));
                if (f.textures) w.n_uv[node] = make_float2(h.u, h.v);
                // the third word - children, refraction weight, path id - has readers only where a hit can have children
                // (k_shade) or draws shadow samples (k_shadow, soft shadows): hits of the last level of a frame with hard
                // shadows, i.e. every hit of a frame without recursion, do not write it
                if (spawn || f.soft_shadows) {
                    const int packed = f.bundle ? (n_reflect | (copies << 13) | (1 << 29) | (refract ? (1 << 30) : 0)) : (n_reflect | (refract ? (1 << 30) : 0));
                    w.n_c[node] = make_float4(i2f(n_children ? first_link : -1), weight, i2f((int)path), i2f(packed));
                }
            }
            if (PRIMARY) w.n_pix[node] = slot;
            const float EPS = 1e-4f;
            // multiplicity of the children: each of the reference's gr_samples copies of a deduplicated ray
            unsigned long long cm = (unsigned long long)mult * (unsigned)(n_reflect == 1 ? f.gr_samples : 1);
            const uint32_t flag_refl = (replicated || (f.replicate && !glossy)) ? 0x80000000u : 0u;
            const uint32_t child_mult = (cm > 0x7fffffffull ? 0x7fffffffu : (uint32_t)cm) | flag_refl;
            for (int j = 0; j < n_entries; ++j) {                        // raytracer.cpp:415-421
                V3 dir = rd;
                uint32_t cpath = mix32(path * 0x9E3779B1u + (uint32_t)j + 1u);
                if (glossy) {
                    uint3 q = pcg3d(path ^ (f.seed * 0x9E3779B1u), (uint32_t)j, (uint32_t)level * 16u + STREAM_GLOSSY);
                    dir = cone_direction(rd, gloss_angle, q.x, q.y);
                }
                V3 org = h.p + N * EPS;
                qa_out[qslot + j] = make_float4(org.x, org.y, org.z, dir.x);
                qb_out[qslot + j] = make_float4(dir.y, dir.z, i2f((int)cpath), i2f((int)child_mult));
            }
            if (refract) {                                               // raytracer.cpp:474-479
                V3 org = h.p + fd * EPS;
                uint32_t cpath = mix32(path * 0x9E3779B1u + 0x7FFFFFFFu);
                qa_out[qslot + n_entries] = make_float4(org.x, org.y, org.z, fd.x);
                qb_out[qslot + n_entries] = make_float4(fd.y, fd.z, i2f((int)cpath), i2f((int)(mult | (replicated ? 0x80000000u : 0u))));
                if (STATS) { ++n_refr; n_refr_ref += (unsigned long long)mult * (unsigned)copies; }
            }
            if (STATS) {
                n_refl += (unsigned)n_entries;
                if (n_reflect) n_refl_ref += (unsigned long long)mult * (unsigned)copies * (unsigned)f.gr_samples;
            }
        }
    }
    if (!STATS) return;                                                // ray statistics only for frames that ask for them
    if (PRIMARY) block_stat_add(&w.stats[RT_S_PRIMARY], n_primary, s_tmp);
    block_stat_add(&w.stats[RT_S_SHADOW_REF], n_shadow_ref, s_tmp);
    if (spawn) {
        block_stat_add(&w.stats[RT_S_REFLECT], n_refl, s_tmp);
        block_stat_add(&w.stats[RT_S_REFRACT], n_refr, s_tmp);
        block_stat_add(&w.stats[RT_S_REFLECT_REF], n_refl_ref, s_tmp);
        block_stat_add(&w.stats[RT_S_REFRACT_REF], n_refr_ref, s_tmp);
    }
}

// Soft shadows of a scene with a handful of primitives: what the sample rays of a (hit, light) pair can hit, decided
// once per pair.  The sample rays lie on lines through the hit point p (they start 0.001 along their own direction,
// raytracer.cpp:549) and aim at points of the ball of radius R around the light (raytracer.cpp:541), so their
// directions lie in the cone around a = (light - p) / D with sin(half angle) = R / D.
//   * MAY: a primitive can only be hit if that cone, widened for what the float tests may report outside the geometry
//     (and swept over the motion), reaches it; a polygon can also be ruled out when every ray of the cone leaves its
//     plane behind.  When nothing can be hit, every sample is unblocked whatever the random numbers are.
//   * SURE: a sphere or cube at rest that every line of the cone pierces well inside, starting clearly outside it and
//     reaching it clearly before the nearest point a sample may aim at, is reported by every sample ray at a distance
//     below the sample's (raytracer.cpp:564): when the search is certain to get to it, every sample is blocked.
// In both cases no ray is traced and the count written is exactly the one tracing would give.  Conservative
// throughout (a "maybe" traces).
DEVI bool cone_may_hit_sphere(V3 p, V3 a, float sin_t, float cos_t, V3 c, float r) {
    const V3 v = c - p;
    const float dv2 = dot(v, v);
    const float re = r * 1.02f + 2e-3f * (__builtin_amdgcn_sqrtf(dv2) + 1.0f);
    if (!(dv2 > re * re * 1.0001f)) return true;                        // the apex is inside (or near) the sphere, or NaN
    const float inv = __builtin_amdgcn_rsqf(dv2);
    const float sin_b = re * inv, cos_b = __builtin_amdgcn_sqrtf(fmaxf(0.0f, 1.0f - sin_b * sin_b));
    const float cos_g = dot(v, a) * inv;                                 // angle between the axis and the centre
    // reachable iff gamma <= theta + beta; theta + beta >= 90 degrees is never ruled out
    const float cos_tb = cos_t * cos_b - sin_t * sin_b;
    return !(cos_tb > 0.0f && cos_g < cos_tb - 1e-3f);
}

// Every line through p whose direction is within the cone passes the centre c at a distance of at most
// |c - p| sin(gamma + theta) (gamma: angle between the axis and the centre).  Below 0.98 r, with p clearly outside,
// sphere.cpp:127-141 finds tca > 0, d2 < r^2 and reports the near root, which is at most |c - o| <= |c - p| + 0.001.
// `reach` = what a reported distance must stay below: the nearest a light sample can be, less a margin.
DEVI bool cone_surely_hits_sphere(V3 p, V3 a, float sin_t, float cos_t, V3 c, float r, float reach) {
    const V3 v = c - p;
    const float dist = __builtin_amdgcn_sqrtf(dot(v, v));
    if (!(dist > r * 1.02f + 0.01f) || !(r > 1e-3f) || !(dist < 1e3f * r)) return false;
    const float cos_g = dot(v, a) / dist;
    if (!(cos_g > 0.0f)) return false;
    const float sin_g = __builtin_amdgcn_sqrtf(fmaxf(0.0f, 1.0f - cos_g * cos_g));
    const float sin_sum = sin_g * cos_t + cos_g * sin_t, cos_sum = cos_g * cos_t - sin_g * sin_t;
    if (!(cos_sum > 0.05f)) return false;
    return dist * sin_sum * 1.001f + 1e-3f < 0.98f * r && dist + 0.002f < reach;
}

// own_gid, N: the primitive the hit lies on and its geometric normal there (as the hit record holds it).  A sphere or a
// cube is convex, so a ray that starts on its surface and heads outwards cannot meet it again; in the reference's
// arithmetic: the sample ray starts at p + 0.001 d (raytracer.cpp:549), so with N.d >= m > 0 over the whole cone
//   sphere.cpp:127-129: tca = (c - o).d = -r N.d - 0.001 +- rounding (a few 1e-6 for coordinates below 1000) < 0: "return false";
//   cube.cpp:170-215: on the axis of the hit's face the local origin is 1 + e + 0.001 N.d / scale (e: the rounding of the
//   hit point and of the two rotations, and the 1e-5 by which the face choice of cube.cpp:270-280 may prefer a
//   neighbouring face at an edge), so the far slab distance is -e / N.d - 0.001 / scale <= 1e-4 as long as
//   e <= N.d (1e-4 + 0.001 / scale): tmax then never exceeds the test's own threshold of 1e-4 and it reports no hit.
// Such a primitive is left out of the search below.
// The same questions against a cube itself rather than the sphere around it, in the cube's own frame (cube.cpp:126-168:
// translate by the location at the ray's time, rotate back, divide by the scale; the cube is then [-1, 1]^3 and
// distances are in units of its half edge).  Seen from a point outside it, a cube of half edge h is the convex polygon
// of its silhouette edges - the edges between a face that looks at the point and one that does not - and a line through
// the point meets the cube exactly when its direction x lies on the inner side of every plane through the point and a
// silhouette edge: n_e . x >= 0 with n_e the plane's unit normal turned towards the cube's centre.  For the cone's
// axis al and half angle theta:
//   min_e n_e . al < -sin(theta): every direction of the cone lies outside one of the planes - no line of the cone meets the cube;
//   min_e n_e . al >  sin(theta): every direction lies inside all of them - every line of the cone passes through the cube.
// An edge along axis k at (s_i h, s_j h) of the other two axes: with d = corner - point, the plane's normal is d x e_k =
// (d_j, -d_i) on the axes (i, j) (cyclic order), so an edge costs a dozen operations.  Returns that minimum;
// `degenerate`: the point lies on the line of a silhouette edge, or inside the cube.
DEVI float cube_silhouette_margin(V3 pl, V3 al, float h, bool& degenerate) {
    const float pv[3] = { pl.x, pl.y, pl.z }, av[3] = { al.x, al.y, al.z };
    float mn = INFINITY;
    degenerate = !(fmaxf(fmaxf(fabsf(pl.x), fabsf(pl.y)), fabsf(pl.z)) > h);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int i = (k + 1) % 3, j = (k + 2) % 3;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float si = (e & 1) ? 1.0f : -1.0f, sj = (e & 2) ? 1.0f : -1.0f;
            const bool fi = si * pv[i] > h, fj = sj * pv[j] > h;      // does the face on either side of the edge look at the point?
            const float di = si * h - pv[i], dj = sj * h - pv[j];
            float ni = dj, nj = -di;
            if (ni * pv[i] + nj * pv[j] > 0.0f) { ni = -ni; nj = -nj; }
            const float n2 = ni * ni + nj * nj;
            const float m = (ni * av[i] + nj * av[j]) * __builtin_amdgcn_rsqf(n2);
            if (fi != fj) { mn = fminf(mn, m); degenerate = degenerate || !(n2 > 1e-12f) || !(m == m); }
        }
    }
    return mn;
}

// (eps: the float test treats a direction component below 1e-5 as zero, which moves a reported point by up to 1e-5 of
// the distance travelled, and its slab distances carry a few 1e-7 of their size in rounding; taken ten times over.)
DEVI bool cone_may_hit_cube(const float4* __restrict__ rec, bool mb, float time, V3 p, V3 a, float sin_t, float cos_t) {
    const float4 w0 = rec[0], w1 = rec[1], w4 = rec[4], w5 = rec[5];
    V3 loc = mk(w0.x, w0.y, w0.z);
    if (mb) loc = mk(lerp_ref(w0.x, w1.x, time), lerp_ref(w0.y, w1.y, time), lerp_ref(w0.z, w1.z, time));
    const float inv = fabsf(w1.w);                                       // 1 / scale
    if (!(inv > 1e-6f && inv < 1e6f)) return true;
    const V3 pl = rot_inv(p - loc, w4, w5) * inv, al = rot_inv(a, w4, w5);
    const float dist = __builtin_amdgcn_sqrtf(dot(pl, pl));
    const float eps = 2e-3f + 1e-4f * (dist + 1.0f);
    bool degenerate;
    const float mn = cube_silhouette_margin(pl, al, (1.0f + eps) * 1.001f, degenerate);
    // (a NaN anywhere makes the comparison false: "may hit")
    return degenerate || !(mn < -(sin_t * 1.002f + 1e-4f));
}

// A cube at rest, shrunk to 0.97: every line of the cone passes through a point well inside the cube, so cube.cpp:170-215
// finds tmin < tmax.  With p outside the slab of some axis by more than 0.02 half edges the ray needs at least 0.02 of
// local distance to get in, so tmin > 1e-4 and is what the test reports - at most the distance to the cube's farthest
// corner (the direction component on that axis is then far above the 1e-5 below which the test treats the ray as parallel).
DEVI bool cone_surely_hits_cube(const float4* __restrict__ rec, V3 p, V3 a, float sin_t, float cos_t, float reach) {
    const float4 w0 = rec[0], w1 = rec[1], w4 = rec[4], w5 = rec[5];
    const float inv = fabsf(w1.w), scale = fabsf(w0.w);
    if (!(inv > 1e-3f && inv < 1e3f)) return false;
    const V3 pl = rot_inv(p - mk(w0.x, w0.y, w0.z), w4, w5) * inv, al = rot_inv(a, w4, w5);
    if (!(fmaxf(fmaxf(fabsf(pl.x), fabsf(pl.y)), fabsf(pl.z)) > 1.02f + 0.002f * inv)) return false;
    const float dist = __builtin_amdgcn_sqrtf(dot(pl, pl));
    if (!(dist < 500.0f) || !((dist + 1.7321f) * scale * 1.001f + 0.003f < reach)) return false;
    bool degenerate;
    const float mn = cube_silhouette_margin(pl, al, 0.97f, degenerate);
    return !degenerate && mn > sin_t * 1.01f + 2e-4f;
}

// Bit g of `may` / `sure`: the primitive with global id g (planes, then cubes, then spheres, each in file order) may
// be hit by a sample ray / is certain to block every sample ray, if the search gets to it.  At most 32 primitives.
struct ConeSets { unsigned may, sure; };
DEVI ConeSets cone_sets(const DevScene& sc, bool mb, float time, V3 p, V3 lc, float R, int own_gid, V3 N, bool want_sure) {
    ConeSets out; out.may = 0xffffffffu; out.sure = 0u;
    const V3 ax = lc - p;
    const float D2 = dot(ax, ax);
    if (!(D2 > 4.0f * R * R) || !(D2 < 1e30f)) return out;              // inside or near the light ball: no useful cone
    out.may = 0u;
    const float invD = __builtin_amdgcn_rsqf(D2);
    const V3 a = ax * invD;
    const float sin_t = R * invD * 1.001f + 1e-4f, cos_t = __builtin_amdgcn_sqrtf(fmaxf(0.0f, 1.0f - sin_t * sin_t));
    const float reach = (D2 * invD - R) * 0.999f - 0.01f;                // every light sample is farther than this
    want_sure = want_sure && !mb;
    const DevPrimSet& S = sc.fileset;
    // smallest N.d over the cone (N is of unit length up to rounding); NaN compares false everywhere below
    const float n_a = dot(N, a);
    const float out_least = n_a * cos_t - __builtin_amdgcn_sqrtf(fmaxf(0.0f, 1.0f - n_a * n_a)) * sin_t;
    const float p_size = fabsf(p.x) + fabsf(p.y) + fabsf(p.z);
    for (int i = 0; i < S.n_spheres; ++i) {
        const float4 w0 = S.spheres[2 * i], w1 = S.spheres[2 * i + 1];
        const int gid = S.n_planes + S.n_cubes + i;
        if (gid == own_gid && out_least > 0.02f && p_size + fabsf(w0.x) + fabsf(w0.y) + fabsf(w0.z) + fabsf(w1.x) + fabsf(w1.y) + fabsf(w1.z) < 1000.0f)
            continue;                                                    // the hit's own sphere, every sample ray leaving it
        V3 c = mk(w0.x, w0.y, w0.z); float r = fabsf(w0.w);
        if (mb) { const V3 e = mk(w1.x, w1.y, w1.z); const V3 h = (e - c) * 0.5f; c = c + h; r += __builtin_amdgcn_sqrtf(dot(h, h)) * 1.001f; }
        if (!cone_may_hit_sphere(p, a, sin_t, cos_t, c, r)) continue;
        out.may |= 1u << gid;
        if (want_sure && w0.w > 0.0f && cone_surely_hits_sphere(p, a, sin_t, cos_t, c, r, reach)) out.sure |= 1u << gid;
    }
    for (int i = 0; i < S.n_cubes; ++i) {
        const float4 w0 = S.cubes[6 * i], w1 = S.cubes[6 * i + 1];
        const int gid = S.n_planes + i;
        if (gid == own_gid && out_least > 0.02f && w0.w > 1e-3f) {
            // e: 1e-5 (face choice) + rounding of the hit point and the rotations, in units of the cube's half edge, doubled
            const float scale = w0.w;
            const float e = 1.5e-5f + 2.0f * (2e-7f * (p_size + fabsf(w0.x) + fabsf(w0.y) + fabsf(w0.z) + fabsf(w1.x) + fabsf(w1.y) + fabsf(w1.z)) + 1e-6f * scale) / scale;
            if (e <= out_least * (1e-4f + 0.001f / scale)) continue;     // the hit's own cube, every sample ray leaving it
        }
        V3 c = mk(w0.x, w0.y, w0.z); float r = fabsf(w0.w) * 1.7325f;    // half edge x sqrt(3)
        if (mb) { const V3 e = mk(w1.x, w1.y, w1.z); const V3 h = (e - c) * 0.5f; c = c + h; r += __builtin_amdgcn_sqrtf(dot(h, h)) * 1.001f; }
        if (!(cone_may_hit_sphere(p, a, sin_t, cos_t, c, r) && cone_may_hit_cube(S.cubes + 6 * i, mb, time, p, a, sin_t, cos_t))) continue;
        out.may |= 1u << gid;
        if (want_sure && w0.w > 0.0f && cone_surely_hits_cube(S.cubes + 6 * i, p, a, sin_t, cos_t, reach)) out.sure |= 1u << gid;
    }
    for (int i = 0; i < S.n_planes; ++i) {
        const float4 w0 = S.planes[3 * i], w1 = S.planes[3 * i + 1];
        // Every ray of the cone moves away from the plane, and starts on it or beyond it: plane.cpp:153-156 then computes
        // t < 0 and rejects.  n.d >= (n.a cos - sqrt(1 - (n.a)^2) sin) |n| over the cone; the start is p + 0.001 d.
        V3 n = mk(w0.x, w0.y, w0.z);
        const float nl2 = dot(n, n);
        bool ruled_out = false;
        if (nl2 > 1e-20f && nl2 < 1e20f) {
            const float inl = __builtin_amdgcn_rsqf(nl2);
            // signed distance of the plane ahead of p, measured to both vertices an offset may come from: the winding-sorted
            // first vertex (this record, what -u tests) and the file-order first vertex (third word; what the leaves' copies
            // test, plane.cpp:149 on bvh.cpp's copies).  They lie in one plane only for a planar polygon.
            const float4 w2 = S.planes[3 * i + 2];
            float na = dot(n, a) * inl, h = dot(n, mk(w1.x - p.x, w1.y - p.y, w1.z - p.z)) * inl, h2 = dot(n, mk(w2.y - p.x, w2.z - p.y, w2.w - p.z)) * inl;
            if (na < 0.0f) { na = -na; h = -h; h2 = -h2; }
            const float least = na * cos_t - __builtin_amdgcn_sqrtf(fmaxf(0.0f, 1.0f - na * na)) * sin_t;    // smallest n.d over the cone
            const float scale = fmaxf(fmaxf(fmaxf(fmaxf(fabsf(p.x), fabsf(p.y)), fmaxf(fabsf(p.z), 1.0f)), fmaxf(fmaxf(fabsf(w1.x), fabsf(w1.y)), fabsf(w1.z))),
                                      fmaxf(fmaxf(fabsf(w2.y), fabsf(w2.z)), fabsf(w2.w)));
            // t = h / (n.d) - 0.001 < 0 as long as h < 0.001 x least; half of that against the rounding of h
            ruled_out = least > 0.02f && fmaxf(h, h2) < 5e-4f * least - 2e-6f * scale;
        }
        if (ruled_out) continue;
        const float4 bs = sc.plane_bound[i];
        if (cone_may_hit_sphere(p, a, sin_t, cos_t, mk(bs.x, bs.y, bs.z), bs.w)) out.may |= 1u << i;
    }
    return out;
}

// What a pair's sample rays need: CONE_NONE = nothing can be hit (count SS), CONE_BLOCKED = every one is blocked
// (count 0), CONE_TRACE = trace them.  Without the BVH a sample is blocked when the closest reported distance over all
// primitives is below the light sample's (raytracer.cpp:202-256, 564): one certain blocker decides.  With it the
// reference returns the result of the last leaf that reported anything (bvh.cpp:35-105) - in this walk's order: of
// the first; so the leaves are taken in that order, and the first one holding anything that may be hit decides: if it
// holds a certain blocker, the leaf's closest distance is below the light sample's whatever else it holds.  (A ray that
// passes well inside a primitive passes well inside every box around it: aabb.cpp:31-63 lets it into the leaf.)
enum { CONE_NONE = 0, CONE_BLOCKED = 1, CONE_TRACE = 2 };
DEVI int cone_outcome(const DevScene& sc, const float4* __restrict__ nodes, const int4* __restrict__ leaves, bool use_bvh, ConeSets cs) {
    if (cs.may == 0u) return CONE_NONE;
    if (cs.sure == 0u) return CONE_TRACE;
    if (!use_bvh) return CONE_BLOCKED;
    const DevPrimSet& S = sc.leafset;
    for (int i = 0; i < sc.n_nodes; ++i) {
        const int leaf = f2i(nodes[2 * i + 1].w);
        if (leaf < 0) continue;
        const int4 l0 = leaves[2 * leaf], l1 = leaves[2 * leaf + 1];
        unsigned held = 0u;
        for (int k = l0.x; k < l0.x + l0.y; ++k) held |= 1u << (f2i(S.planes[3 * k + 2].x) & 31);
        for (int k = l0.z; k < l0.z + l0.w; ++k) held |= 1u << (f2i(S.cubes[6 * k + 3].z) & 31);
        for (int k = l1.x; k < l1.x + l1.y; ++k) held |= 1u << (f2i(S.spheres[2 * k + 1].w) & 31);
        if (held & cs.may) return (held & cs.sure) ? CONE_BLOCKED : CONE_TRACE;
    }
    return CONE_TRACE;                                                   // (not reached: every primitive sits in a leaf)
}

// ---------------------------------------------------------------------------------------------------------
// shadow: one lane per (hit, light, sample).  Hard shadows (raytracer.cpp:509-530): any hit at any distance
// blocks.  Soft shadows (raytracer.cpp:534-571): ss_samples rays to points in a ball around the light, blocked
// when the reported hit is nearer than the sampled point.  Output = number of unblocked samples per (hit, light).
// ---------------------------------------------------------------------------------------------------------
// MODE: 0 = hard shadows, 1 = soft shadows in the reference's arithmetic, 2 = soft shadows on reciprocals (separate
// kernels: each needs fewer registers than one kernel holding both searches, which decides how many wavefronts fit)
// PHASE (see `phase` below): compiled apart, so that each launch carries the registers of its own part only
template <int SUB, bool LDSP, int MODE, int PHASE>
__global__ __launch_bounds__(RT_BLOCK, SUB ? RT_SHADOW_WAVES_SUB : (PHASE == 1 ? RT_SHADOW_WAVES - 1 : RT_SHADOW_WAVES)) void k_shadow(DevScene sc_arg, DevFrame f, DevWork w, int lv_begin, int lv_end, int sq_k) {
    constexpr int phase = PHASE;
    extern __shared__ float4 s_mem[];
    DevScene sc = sc_arg;
    float4* s_nodes = s_mem;
    // with second-level hierarchies the LDS budget goes to the traversal stacks (occupancy): leaf records, read once
    // per leaf visit, then stay in global memory
    const int4* s_leaves = SUB ? sc.leaves : (const int4*)(s_mem + 2 * sc.n_nodes);
    int* s_stack = (int*)(s_mem + 2 * sc.n_nodes) + threadIdx.x;                      // [depth][thread], only with SUB
    __shared__ unsigned long long s_tmp[RT_BLOCK / 64];
    // The launch's work = chunks of RT_BLOCK items inside segments; a segment = the hits of one region of one of the
    // levels lv_begin..lv_end; an item = a (slot, light) pair - slot = one copy of a hit, see DevWork - or, for sample
    // counts that do not tile a wavefront, a (slot, light, sample) ray.  Workgroup b takes chunks b, b + gridDim.x, ...
    constexpr int MAX_SEG = RT_MAX_LEVELS * RT_REGIONS;
    __shared__ int s_pre[MAX_SEG + 1];                                  // first chunk of every segment
    __shared__ unsigned s_items[MAX_SEG];                               // shadow rays of the segment
    __shared__ int s_nbase[MAX_SEG];                                    // node index of the segment's first hit
    __shared__ int s_sbase[MAX_SEG];                                    // slot of that node's copy 0
    __shared__ int s_cstride[MAX_SEG];                                  // slots between two copies of a node
    __shared__ int s_mul[MAX_SEG];                                      // slots per node in the segment
    const int L = sc.n_lights;
    constexpr bool FAST = MODE == 2;
    const int SS = MODE ? f.ss_samples : 1;
    // grouped: the SS sample rays of a (hit, light) pair sit side by side inside one wavefront and are counted with a
    // ballot.  The launch's items are then the PAIRS (see the two phases below), otherwise the rays.
    const bool grouped = (SS <= 64) && ((SS & (SS - 1)) == 0);
    const int ss_shift = 31 - __clz(SS);
    const int pairs_per_pass = RT_BLOCK >> ss_shift;
    // phase: 0 = the launch classifies its pairs and traces what needs tracing; 1 = it classifies and hands the pairs to
    // trace to queue sq_k in memory; 2 = it traces that queue (segments = the queue's regions, chunks = pairs_per_pass
    // pairs: every workgroup pass is full whatever part of the image the pairs came from).
    const unsigned per = grouped ? (unsigned)L : (unsigned)(L * SS);    // the host keeps a segment's rays below 2^31
    const int n_seg = phase == 2 ? w.regions : (lv_end - lv_begin + 1) * w.regions;
    for (int sg = threadIdx.x; sg < n_seg; sg += blockDim.x) {
        if (phase == 2) { s_items[sg] = (unsigned)min(w.ctr[RT_CTR_SHADOWQ(sq_k, sg)], w.sq_rcap); continue; }
        const int lv = lv_begin + sg / w.regions, rg = sg % w.regions;
        // (phase 1 decides per (node, light): the copies of a node share the answer)
        s_items[sg] = (unsigned)min(w.ctr[RT_CTR_NODE(lv, rg)], w.node_rcap[lv]) * (phase == 1 ? (unsigned)L : (unsigned)w.slot_mul[lv] * per);
        s_nbase[sg] = w.node_off[lv] + rg * w.node_rcap[lv];
        s_sbase[sg] = w.slot_off[lv] + rg * w.node_rcap[lv] * w.slot_mul[lv];
        s_mul[sg] = w.slot_mul[lv];
        s_cstride[sg] = w.node_rcap[lv];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned per_chunk = phase == 2 ? (unsigned)pairs_per_pass : (unsigned)RT_BLOCK;
        int acc = 0;
        for (int sg = 0; sg < n_seg; ++sg) { s_pre[sg] = acc; acc += (int)((s_items[sg] + per_chunk - 1) / per_chunk); }
        s_pre[n_seg] = acc;
    }
    __syncthreads();
    const int n_chunks = s_pre[n_seg];
    if ((int)blockIdx.x >= n_chunks) return;
    if (f.use_bvh) stage_bvh(sc, s_nodes, (int4*)(s_mem + 2 * sc.n_nodes), !SUB);
    if (LDSP) stage_prims(sc, s_mem + 2 * sc.n_nodes + 2 * sc.n_leaves);

    // the cone test loops over every primitive: for scenes with a handful of them
    const bool cull = f.cull_cones && sc.fileset.n_planes + sc.fileset.n_cubes + sc.fileset.n_spheres <= 8;
    int* __restrict__ out = (int*)w.shadow;
    unsigned long long n_rays = 0;
    const float4 light0 = sc.lights[0];                                 // count > 0 implies a light; the usual scene has one

    // One sample ray: hard shadows (raytracer.cpp:509-530) or sample j of a soft shadow (raytracer.cpp:534-571).  Returns
    // the distance the reported hit is compared with (the light sample's; -inf for hard shadows: any hit blocks).
    // (path: the hit's path id, n_c[node].z, when the caller has it already; `node` is then not read)
    auto make_sample_ray = [&](int node, bool have_path, uint32_t path, int cp, int l, int j, const float4 a, const float4 lt, Ray& sr) __attribute__((always_inline)) -> float {
        const V3 p = mk(a.x, a.y, a.z);
        sr.time = a.w;
        if (MODE == 0) {
            const float4 b = w.n_b[node];
            sr.d = normalise(mk(lt.x - p.x, lt.y - p.y, lt.z - p.z));
            sr.o = p + mk(b.x, b.y, b.z) * 1e-2f;                        // raytracer.cpp:515: unflipped geometric normal
            return -INFINITY;
        }
        if (!have_path) path = (uint32_t)f2i(w.n_c[node].z);
        if (cp) path = mix32(path + 0x9E3779B1u * (uint32_t)cp);        // every copy of a hit draws its own samples
        const uint3 q = pcg3d(path ^ (f.seed * 0x9E3779B1u), (uint32_t)j, (uint32_t)l * 16u + STREAM_SHADOW);
        const V3 jit = unit_sphere(u01(q.x), u01(q.y), u01(q.z)) * f.light_radius;
        const V3 h2l = mk(lt.x + jit.x, lt.y + jit.y, lt.z + jit.z) - p;
        V3 ld; float lightDist;                                          // raytracer.cpp:547,553: unit direction and distance
        const float l2 = dot(h2l, h2l);
        if (FAST && l2 > 1e-30f && l2 < 1e30f) { const float rs = __builtin_amdgcn_rsqf(l2); ld = h2l * rs; lightDist = l2 * rs; }
        else { ld = normalise(h2l); lightDist = len_ref(h2l); }
        sr.o = p + ld * 0.001f;                                          // raytracer.cpp:549
        sr.d = ld;
        return lightDist;
    };
    auto sample_unblocked = [&](int node, bool have_path, uint32_t path, int cp, int l, int j, const float4 a, const float4 lt) __attribute__((always_inline)) -> bool {
        Ray sr;
        Best best; best.t = 0.0f; best.cls = 0; best.idx = 0; best.fileset = 0;
        const float lightDist = make_sample_ray(node, have_path, path, cp, l, j, a, lt, sr);
        if (MODE == 0) return !trace<true, SUB>(sc, f, s_nodes, s_leaves, s_stack, sr, false, best);
        // only "closest reported distance < lightDist" is asked for (raytracer.cpp:564)
        const bool hit = trace<false, SUB, FAST>(sc, f, s_nodes, s_leaves, s_stack, sr, false, best, lightDist);
        return !(hit && best.t < lightDist);
    };
    // raytracer.cpp:322-336: the shadow factor only ever multiplies max(0, N.L).  Where the surface clearly faces away from
    // the light that product is zero whatever the rays would find, so they are not traced (the count written is 0).
    // "Clearly": beyond any rounding of the shader's normalised N.L; NaNs fail the test and trace.
    auto faces_light = [&](const float4 a, const float4 b, const float4 lt) -> bool {
        if (!f.skip_unlit) return true;
        const V3 nn = mk(b.x, b.y, b.z), Lr = mk(lt.x - a.x, lt.y - a.y, lt.z - a.z);
        const float d = dot(nn, Lr);
        return !(d < 0.0f && d * d > 1e-6f * dot(nn, nn) * dot(Lr, Lr));
    };

    int seg = 0;
    if (grouped && phase == 2) {
        // the queued pairs, pairs_per_pass at a time: SS neighbouring lanes per pair, counted with a ballot.  An entry
        // holds what a sample ray needs - (index into shadow[], copy, path id, hit time), (hit point) - and the
        // next chunk's entries are requested before this chunk's rays are traced.
        const int j = (int)(threadIdx.x & (unsigned)(SS - 1));
        auto fetch = [&](int g, int4& e0, float4& e1) -> bool {
            if (g >= n_chunks) return false;
            while (seg < n_seg - 1 && g >= __builtin_amdgcn_readfirstlane(s_pre[seg + 1])) ++seg;
            const int k = (g - __builtin_amdgcn_readfirstlane(s_pre[seg])) * pairs_per_pass + (int)(threadIdx.x >> ss_shift);
            if (k >= __builtin_amdgcn_readfirstlane((int)s_items[seg])) return false;
            const int4* __restrict__ e = w.sq + 2 * ((size_t)(sq_k * w.regions + seg) * (size_t)w.sq_rcap + (size_t)k);
            e0 = e[0]; e1 = *(const float4*)(e + 1);
            return true;
        };
        int4 c0 = make_int4(0, 0, 0, 0), n0 = c0; float4 c1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), n1 = c1;
        bool active = fetch((int)blockIdx.x, c0, c1);
        for (int g = (int)blockIdx.x; g < n_chunks; g += (int)gridDim.x) {
            const bool next_active = fetch(g + (int)gridDim.x, n0, n1);
            bool unblocked = false;
            if (active) {
                const int l = (L == 1) ? 0 : c0.x % L;
                ++n_rays;
                unblocked = sample_unblocked(0, true, (uint32_t)c0.z, c0.y, l, j, make_float4(c1.x, c1.y, c1.z, i2f(c0.w)), (L == 1) ? light0 : sc.lights[l]);
            }
            const unsigned long long m = __ballot(active && unblocked);
            if (active && j == 0) {
                const int lane0 = __lane_id();
                out[c0.x] = __popcll((SS == 64) ? m : ((m >> lane0) & ((1ull << SS) - 1ull)));
            }
            active = next_active; c0 = n0; c1 = n1;
        }
        if (f.count_stats) block_stat_add(&w.stats[RT_S_SHADOW], n_rays, s_tmp);
        return;
    }
    if (grouped && !SUB && MODE != 0 && phase == 1) {
        // One lane per (node, light): what its pairs - one per copy of the node, all with the same hit point - need.
        // Nothing when the surface faces away from the light (count 0), when no sample ray can reach any primitive
        // (count SS) or when every sample ray is certain to be blocked (count 0; see cone_sets): the counts are written
        // here.  Otherwise an entry per copy goes to the queue in memory: one atomic per wavefront that queues anything
        // (the regions spread those over several counters), no workgroup-wide step.  What a full region cannot take is
        // traced here, lane by lane.
        float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), b = a, a_next = a, b_next = a;
        struct NodeAt { bool valid; int item0, node, l, mul, cstep; };     // cstep: from one copy's count to the next
        auto locate = [&](int g, float4& ra, float4& rb) -> NodeAt {
            NodeAt at; at.valid = false; at.item0 = at.node = at.l = at.cstep = 0; at.mul = 1;
            if (g >= n_chunks) return at;
            while (seg < n_seg - 1 && g >= __builtin_amdgcn_readfirstlane(s_pre[seg + 1])) ++seg;
            const unsigned i = (unsigned)(g - __builtin_amdgcn_readfirstlane(s_pre[seg])) * RT_BLOCK + threadIdx.x;   // (node, light) of the segment
            if (i >= (unsigned)__builtin_amdgcn_readfirstlane((int)s_items[seg])) return at;
            int node_local;
            if (L == 1) { node_local = (int)i; at.l = 0; } else { node_local = (int)(i / (unsigned)L); at.l = (int)(i % (unsigned)L); }
            at.mul = __builtin_amdgcn_readfirstlane(s_mul[seg]);
            at.node = __builtin_amdgcn_readfirstlane(s_nbase[seg]) + node_local;
            at.item0 = (__builtin_amdgcn_readfirstlane(s_sbase[seg]) + node_local) * L + at.l;
            at.cstep = __builtin_amdgcn_readfirstlane(s_cstride[seg]) * L;
            ra = w.n_a[at.node]; rb = w.n_b[at.node];
            at.valid = true;
            return at;
        };
        const int lane = (int)__lane_id();
        const int rg = (int)((blockIdx.x * (RT_BLOCK / 64) + (threadIdx.x >> 6)) % (unsigned)w.regions);
        int4* __restrict__ qbase = w.sq + 2 * (size_t)(sq_k * w.regions + rg) * (size_t)w.sq_rcap;
        NodeAt cur = locate((int)blockIdx.x, a, b);
        for (int g = (int)blockIdx.x; g < n_chunks; g += (int)gridDim.x) {
            const NodeAt nxt = locate(g + (int)gridDim.x, a_next, b_next);
            int n_push = 0;
            // a node of a level with several slots per node has either one copy or all of them (top bit of the id word)
            const int ncp = (cur.valid && cur.mul > 1 && f2i(b.w) < 0) ? cur.mul : 1;
            const float4 lt = (L == 1) ? light0 : sc.lights[cur.l];
            if (cur.valid) {
                int what = CONE_TRACE;
                if (!faces_light(a, b, lt)) what = CONE_BLOCKED;
                else if (cull) {
                    const ConeSets cs = cone_sets(sc, f.motion_blur != 0, a.w, mk(a.x, a.y, a.z), mk(lt.x, lt.y, lt.z), f.light_radius,
                                                  f.cull_own ? (f2i(b.w) & 0x7fffffff) : -1, mk(b.x, b.y, b.z), f.cull_umbra != 0);
                    what = cone_outcome(sc, s_nodes, s_leaves, f.use_bvh != 0, cs);
                }
                if (what == CONE_TRACE) n_push = ncp;
                else for (int cp = 0; cp < ncp; ++cp) out[cur.item0 + cp * cur.cstep] = what == CONE_NONE ? SS : 0;
            }
            const unsigned long long m1 = __ballot(n_push == 1), mm = __ballot(n_push > 1);
            if (m1 | mm) {
                const int mulw = __builtin_amdgcn_readfirstlane(cur.mul);       // (lanes with n_push > 1 push their segment's mul; a wavefront's chunk lies in one segment)
                const int total = __popcll(m1) + mulw * __popcll(mm);
                const unsigned long long below = (1ull << lane) - 1ull;
                int at = 0;
                const int first_lane = __ffsll((long long)(m1 | mm)) - 1;
                if (lane == first_lane) at = atomicAdd(&w.ctr[RT_CTR_SHADOWQ(sq_k, rg)], total);
                at = __shfl(at, first_lane, 64) + __popcll(m1 & below) + mulw * __popcll(mm & below);
                if (n_push) {
                    const int path = f2i(w.n_c[cur.node].z);
                    for (int cp = 0; cp < n_push; ++cp) {
                        if (at + cp < w.sq_rcap) {
                            int4* __restrict__ e = qbase + 2 * (size_t)(at + cp);
                            e[0] = make_int4(cur.item0 + cp * cur.cstep, cp, path, f2i(a.w));
                            *(float4*)(e + 1) = make_float4(a.x, a.y, a.z, 0.0f);
                        } else {
                            int cnt = 0;
                            for (int j = 0; j < SS; ++j) { ++n_rays; cnt += sample_unblocked(cur.node, true, (uint32_t)path, cp, cur.l, j, a, lt) ? 1 : 0; }
                            out[cur.item0 + cp * cur.cstep] = cnt;
                        }
                    }
                }
            }
            cur = nxt; a = a_next; b = b_next;
        }
        if (f.count_stats) block_stat_add(&w.stats[RT_S_SHADOW], n_rays, s_tmp);
        return;
    }
    if (grouped) {
        // Two phases per round.  (1) One lane per (hit, light) PAIR of a chunk of RT_BLOCK pairs decides what the pair
        // needs: nothing when the surface faces away from the light (count 0), when no sample ray can reach any primitive
        // (count SS) or when every sample ray is certain to be blocked (count 0; see cone_sets), otherwise its SS
        // sample rays, and then queues the pair in LDS.  (2) Queued
        // pairs are expanded to rays, RT_BLOCK at a time, SS neighbouring lanes per pair, and counted with a ballot.
        // The per-pair work is so done once, not once per sample, and every wavefront that traces is full of rays that
        // need tracing (the workgroup's last one aside).
        // (scenes with second-level hierarchies collect more pairs before they trace: see the drain below)
        constexpr int QCAP = 2 * RT_BLOCK;
        __shared__ int s_q[QCAP], s_qnode[QCAP];                      // queued pairs: index into `out`; node
        __shared__ unsigned short s_qcp[QCAP];                        // ... and which of the node's copies (< 4096)
        __shared__ int s_qn, s_next;
        __shared__ unsigned s_qcnt[SUB ? QCAP / 2 : 1];              // the drain's unblocked samples per queued pair, 16 bits each
        if (threadIdx.x == 0) { s_qn = 0; s_next = 0; }
        if (SUB) for (int k = threadIdx.x; k < QCAP / 2; k += blockDim.x) s_qcnt[k] = 0u;
        __syncthreads();
        // which pair of chunk g this lane decides, and its hit record (requested one chunk ahead of its use)
        struct PairAt { bool valid; int item, node, cp, l; };
        auto locate = [&](int g, float4& a, float4& b) -> PairAt {
            PairAt at; at.valid = false; at.item = at.node = at.cp = at.l = 0;
            if (g >= n_chunks) return at;
            while (seg < n_seg - 1 && g >= __builtin_amdgcn_readfirstlane(s_pre[seg + 1])) ++seg;
            const unsigned i = (unsigned)(g - __builtin_amdgcn_readfirstlane(s_pre[seg])) * RT_BLOCK + threadIdx.x;   // pair of the segment
            if (i >= (unsigned)__builtin_amdgcn_readfirstlane((int)s_items[seg])) return at;
            int dense;
            if (L == 1) { dense = (int)i; at.l = 0; } else { dense = (int)(i / (unsigned)L); at.l = (int)(i % (unsigned)L); }
            const int mul = __builtin_amdgcn_readfirstlane(s_mul[seg]);
            int node_local = dense;
            if (mul > 1) {
                if ((mul & (mul - 1)) == 0) { node_local = dense >> (31 - __clz(mul)); at.cp = dense & (mul - 1); }
                else { node_local = dense / mul; at.cp = dense - node_local * mul; }
            }
            at.node = __builtin_amdgcn_readfirstlane(s_nbase[seg]) + node_local;
            at.item = (__builtin_amdgcn_readfirstlane(s_sbase[seg]) + at.cp * __builtin_amdgcn_readfirstlane(s_cstride[seg]) + node_local) * L + at.l;
            a = w.n_a[at.node]; b = w.n_b[at.node];
            at.valid = true;
            return at;
        };
        float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), b = a, a_next = a, b_next = a;
        PairAt cur = locate((int)blockIdx.x, a, b);
        for (int g = (int)blockIdx.x; g < n_chunks; g += (int)gridDim.x) {
            const PairAt nxt = locate(g + (int)gridDim.x, a_next, b_next);
            bool queue = false; const int item = cur.item, qnode = cur.node; int qcp = cur.cp;
            if (cur.valid) {
                const int l = cur.l;
                // a node of a level with several slots per node has either one copy or all of them (top bit of the id word)
                if (cur.cp == 0 || f2i(b.w) < 0) {
                    const float4 lt = (L == 1) ? light0 : sc.lights[l];
                    if (!faces_light(a, b, lt)) out[item] = 0;
                    else if (!SUB && MODE != 0 && cull) {
                        const ConeSets cs = cone_sets(sc, f.motion_blur != 0, a.w, mk(a.x, a.y, a.z), mk(lt.x, lt.y, lt.z), f.light_radius,
                                                      f.cull_own ? (f2i(b.w) & 0x7fffffff) : -1, mk(b.x, b.y, b.z), f.cull_umbra != 0);
                        const int what = cone_outcome(sc, s_nodes, s_leaves, f.use_bvh != 0, cs);
                        if (what == CONE_TRACE) queue = true; else out[item] = what == CONE_NONE ? SS : 0;
                    }
                    else queue = true;
                }
            }
            cur = nxt; a = a_next; b = b_next;
            {
                const unsigned long long m = __ballot(queue);
                int base = 0;
                if (__lane_id() == 0 && m) base = atomicAdd(&s_qn, __popcll(m));
                base = __shfl(base, 0, 64);
                if (queue) { const int at = base + __popcll(m & ((1ull << __lane_id()) - 1ull)); s_q[at] = item; s_qnode[at] = qnode; s_qcp[at] = (unsigned short)qcp; }
            }
            __syncthreads();
            int qn = s_qn;
            __syncthreads();                                             // everyone has read the count before anyone appends again
            const bool last = g + (int)gridDim.x >= n_chunks;
            if (SUB && MODE != 0) {
                // (Hard shadows send one ray per pair: a queue's worth of them is two rays per lane, too few for what follows to
                // pay; they keep the passes below.)
                // Walks through second-level hierarchies differ widely in length, and a wavefront whose lanes each trace one
                // ray lasts as long as its longest walk (measured on 10^7 primitives: 26 of 64 lanes busy per step).  So the
                // queued pairs' rays are not dealt out 256 at a time: once no further chunk of pairs fits the queue (or after
                // the last chunk) every wavefront runs a loop in which a lane that has finished its ray takes the next one
                // of the queue at once - one step of a walk per iteration, whatever ray the lane holds.  Unblocked samples
                // are added to the pair's count with an atomic (the lanes of a pair no longer sit side by side).
                if ((qn > QCAP - RT_BLOCK || last) && qn > 0) {
                    // What a lane does next: nothing (no ray), the first-level walk to its next leaf, a record of a leaf's
                    // hierarchy, or the test of one primitive of a run - by class, so that a pass runs one test for all its
                    // lanes.  Each kind of work is a branch with its own trip to memory; a wavefront runs a kind when at
                    // least RT_REFILL_LANES of its lanes wait for it (the others keep their state), or, when no kind has
                    // that many, the one most lanes wait for.
                    enum { K_IDLE = 0, K_FIRST = 1, K_STEP = 2, K_PLANE = 3, K_CUBE = 4, K_SPHERE = 5 };
                    const int n_queued_rays = qn << ss_shift;
                    const bool mb = f.motion_blur != 0;
                    const int lane = (int)__lane_id();
                    const int n_first = sc.n_nodes;
                    const bool leaf_rule = f.use_bvh != 0;
                    const DevPrimSet& S = f.use_bvh ? sc.leafset : sc.walkset;
                    int kind = K_IDLE, pr = 0, i = 0;
                    bool started = false;
                    Ray sr; sr.o = mk(0, 0, 0); sr.d = mk(0, 0, 1); sr.time = 0.0f;
                    SubRay q = make_sub_ray(sc, sr);
                    float lightDist = -INFINITY; V3 rcp = mk(0, 0, 0); bool fast_ok = false;
                    SubWalk sw; sub_begin(sw, 0, 1, true);
                    unsigned key[4] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu }; int4 ev = make_int4(-1, -1, -1, -1); unsigned cand = 0;
                    bool more = true;                                // (wavefront-uniform) the queue may still hold rays
                    // a sample is unblocked unless the group that ends the walk reports a hit nearer than the light sample
                    // (raytracer.cpp:564; any hit for hard shadows, raytracer.cpp:520-529)
                    // (counted in LDS: an atomic on the pair's count in memory would stand, in the wavefront's in-order memory
                    // counter, before the loads of its next step)
                    auto finish = [&](bool blocked) { if (!blocked) atomicAdd(&s_qcnt[pr >> 1], 1u << (16 * (pr & 1))); kind = K_IDLE; };
                    // the walk of a group ended without a hit that decides: a hit at or beyond the light sample ends the
                    // reference's search too (bvh.cpp:97-101), no hit sends the ray on to its next leaf
                    auto exhausted = [&]() { if (sw.has || !f.use_bvh) finish(false); else kind = K_FIRST; };
                    auto pop = [&]() { if (sub_pop(sw, s_stack) == SUB_EXHAUSTED) exhausted(); else kind = K_STEP; };
                    // the run's next primitive whose box is entered before what still matters, nearest box first
                    auto next_candidate = [&]() {
                        for (;;) {
                            const unsigned m = min(min(key[0], key[1]), min(key[2], key[3]));
                            if (m == 0xffffffffu || __uint_as_float(m & ~3u) > sw.prune_t) { pop(); return; }
                            const unsigned k = m & 3u;
                            key[0] = k == 0u ? 0xffffffffu : key[0]; key[1] = k == 1u ? 0xffffffffu : key[1];
                            key[2] = k == 2u ? 0xffffffffu : key[2]; key[3] = k == 3u ? 0xffffffffu : key[3];
                            const unsigned entry = (unsigned)(k == 0u ? ev.x : k == 1u ? ev.y : k == 2u ? ev.z : ev.w);
                            if (entry == 0xffffffffu) continue;
                            cand = entry & 0x3fffffffu; kind = K_PLANE + (int)(entry >> 30);
                            return;
                        }
                    };
                    auto tested = [&](bool ok, float t) {
                        if (ok && (!leaf_rule || t >= 0.0f) && t < sw.best_t) {
                            sw.has = true; sw.best_t = t;
                            if (MODE == 0 || t < lightDist) { finish(true); return; }
                            sw.prune_t = lightDist;                  // only a hit nearer than the light sample still matters
                        }
                        next_candidate();
                    };
                    for (;;) {
                        const int n_idle = __popcll(__ballot(kind == K_IDLE));
                        if (more && (n_idle >= RT_REFILL_LANES || n_idle == 64)) {
                            // the idle lanes take the next rays of the queue: one LDS atomic per wavefront
                            const unsigned long long idle = __ballot(kind == K_IDLE);
                            const int first_idle = __ffsll((long long)idle) - 1;
                            int base = 0;
                            if (lane == first_idle) base = atomicAdd(&s_next, n_idle);
                            base = __shfl(base, first_idle, 64);
                            more = base + n_idle < n_queued_rays;
                            if (kind == K_IDLE) {
                                const int my = base + __popcll(idle & ((1ull << lane) - 1ull));
                                if (my < n_queued_rays) {
                                    pr = my >> ss_shift;
                                    const int j = my & (SS - 1), it = s_q[pr];
                                    const int node = s_qnode[pr], cp = (int)(s_qcp[pr] & 0xffffu);
                                    const int l = (L == 1) ? 0 : it % L;
                                    ++n_rays;
                                    lightDist = make_sample_ray(node, false, 0u, cp, l, j, w.n_a[node], (L == 1) ? light0 : sc.lights[l], sr);
                                    q = make_sub_ray(sc, sr);
                                    const float ax = fabsf(sr.d.x), ay = fabsf(sr.d.y), az = fabsf(sr.d.z);
                                    fast_ok = ax > 1e-18f && ax < 1e18f && ay > 1e-18f && ay < 1e18f && az > 1e-18f && az < 1e18f;
                                    rcp = mk(__builtin_amdgcn_rcpf(sr.d.x), __builtin_amdgcn_rcpf(sr.d.y), __builtin_amdgcn_rcpf(sr.d.z));
                                    kind = K_FIRST; started = false; i = 0;
                                }
                            }
                        }
                        const int n1 = __popcll(__ballot(kind == K_FIRST)), n2 = __popcll(__ballot(kind == K_STEP)), n3 = __popcll(__ballot(kind == K_PLANE)),
                                  n4 = __popcll(__ballot(kind == K_CUBE)), n5 = __popcll(__ballot(kind == K_SPHERE));
                        const int most = max(max(n1, n2), max(n3, max(n4, n5)));
                        if (most == 0) { if (!more) break; continue; }      // every ray of this wavefront is done (and the queue empty)
                        const int need = min(most, RT_REFILL_LANES);
                        if (n2 >= need && kind == K_STEP) {
                            unsigned off;
                            if (sub_boxes<SUB == 2>(sc, sw, sr, q, mb, key, ev, off)) { RT_WS(sc, WS_RUN); next_candidate(); }
                            else if (sub_descend(sw, key, s_stack) == SUB_EXHAUSTED) exhausted();
                        }
                        if (n4 >= need && kind == K_CUBE) {
                            float t; RT_WS(sc, WS_CUBE);
                            const bool ok = cube_test<false, FAST>(S.cubes + 6 * cand, sr, mb, false, t, nullptr);
                            tested(ok, t);
                        }
                        if (n5 >= need && kind == K_SPHERE) {
                            float t; RT_WS(sc, WS_SPHERE);
                            const bool ok = sphere_test<false, FAST>(S.spheres + 2 * cand, sr, mb, false, t, nullptr);
                            tested(ok, t);
                        }
                        if (n3 >= need && kind == K_PLANE) {
                            float t; RT_WS(sc, WS_PLANE);
                            const bool ok = plane_test<false, FAST>(S.planes + 3 * cand, S.edges, nullptr, sr, false, t, nullptr);
                            tested(ok, t);
                        }
                        if (n1 >= need && kind == K_FIRST) {
                            if (f.use_bvh) {
                                // on to the next leaf whose box the ray pierces (aabb.cpp:31-63, in the reference's visiting order)
                                int leaf = -1;
                                while (i < n_first) {
                                    const float4 a = s_nodes[2 * i], b = s_nodes[2 * i + 1];
                                    RT_WS(sc, WS_BOX);
                                    if (!slab<FAST>(sr, a, b, rcp, fast_ok)) { i = f2i(a.w); continue; }
                                    ++i;
                                    leaf = f2i(b.w);
                                    if (leaf >= 0) break;
                                }
                                if (leaf < 0) finish(false);
                                else {
                                    const int4 l0 = s_leaves[2 * leaf], l1 = s_leaves[2 * leaf + 1];
                                    RT_WS(sc, WS_GROUP);
                                    if (l1.z >= 0) { sub_begin(sw, l1.z, l1.w, true); kind = K_STEP; }
                                    else {
                                        Best best; best.t = 0.0f; best.cls = 0; best.idx = 0; best.fileset = 0;
                                        if (test_prims<MODE == 0, FAST>(sc.leafset, sr, mb, l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, true, lightDist, best))
                                            finish(MODE == 0 || best.t < lightDist);
                                    }
                                }
                            } else if (!started) {
                                // raytracer.cpp:202-256: every primitive of the scene file
                                started = true;
                                if (sc.file_sub_root >= 0) { sub_begin(sw, sc.file_sub_root, sc.file_sub_depth, false); kind = K_STEP; }
                                else {
                                    const DevPrimSet& FS = sc.walkset;
                                    Best best; best.t = 0.0f; best.cls = 0; best.idx = 0; best.fileset = 0;
                                    const bool hit = test_prims<MODE == 0, FAST>(FS, sr, mb, 0, FS.n_planes, 0, FS.n_cubes, 0, FS.n_spheres, false, lightDist, best);
                                    finish(hit && (MODE == 0 || best.t < lightDist));
                                }
                            } else finish(false);
                        }
                    }
                    // every wavefront's rays are done: the pairs' counts go to memory, and the accumulators back to zero
                    __syncthreads();
                    for (int k = threadIdx.x; k < qn; k += blockDim.x) out[s_q[k]] = (int)((s_qcnt[k >> 1] >> (16 * (k & 1))) & 0xffffu);
                    __syncthreads();
                    for (int k = threadIdx.x; k < (qn + 1) / 2; k += blockDim.x) s_qcnt[k] = 0u;
                    qn = 0;
                }
                __syncthreads();
                if (threadIdx.x == 0) { s_qn = qn; s_next = 0; }
                __syncthreads();
                continue;
            }
            while (qn >= pairs_per_pass || (last && qn > 0)) {           // after the last chunk the queue is emptied
                const int take = min(qn, pairs_per_pass), first = qn - take;
                const int k = (int)(threadIdx.x >> ss_shift), j = (int)(threadIdx.x & (unsigned)(SS - 1));
                const bool active = k < take;
                bool unblocked = false; int it = 0;
                if (active) {
                    it = s_q[first + k];
                    const int node = s_qnode[first + k];
                    const int cp = (int)(s_qcp[first + k] & 0xffffu);
                    const int l = (L == 1) ? 0 : it % L;
                    ++n_rays;
                    unblocked = sample_unblocked(node, false, 0u, cp, l, j, w.n_a[node], (L == 1) ? light0 : sc.lights[l]);
                }
                const unsigned long long m = __ballot(active && unblocked);
                if (active && j == 0) {
                    const int lane0 = __lane_id();
                    const unsigned long long grp = (SS == 64) ? m : ((m >> lane0) & ((1ull << SS) - 1ull));
                    out[it] = __popcll(grp);
#ifdef RT_SHADOW_HIST
                    // development: how many traced pairs end fully blocked / partly / fully unblocked (level 0; deeper levels)
                    { const int cnt = __popcll(grp); atomicAdd(&w.stats[RT_S_WALK + (lv_begin ? 20 : 16) + (cnt == 0 ? 0 : cnt == SS ? 2 : 1)], 1ull); }
#endif
                }
                qn = first;
            }
            __syncthreads();
            if (threadIdx.x == 0) s_qn = qn;
            __syncthreads();
        }
        if (f.count_stats) block_stat_add(&w.stats[RT_S_SHADOW], n_rays, s_tmp);
        return;
    }
    // sample counts that do not tile a wavefront: one lane per ray, unblocked samples added with atomics
    for (int g = (int)blockIdx.x; g < n_chunks; g += (int)gridDim.x) {
        while (seg < n_seg - 1 && g >= __builtin_amdgcn_readfirstlane(s_pre[seg + 1])) ++seg;
        const unsigned i = (unsigned)(g - __builtin_amdgcn_readfirstlane(s_pre[seg])) * RT_BLOCK + threadIdx.x;   // ray of the segment
        if (i >= (unsigned)__builtin_amdgcn_readfirstlane((int)s_items[seg])) continue;
        const unsigned grp = i / (unsigned)SS; const int j = (int)(i % (unsigned)SS);
        int dense, l;
        if (L == 1) { dense = (int)grp; l = 0; } else { dense = (int)(grp / (unsigned)L); l = (int)(grp % (unsigned)L); }
        const int mul = __builtin_amdgcn_readfirstlane(s_mul[seg]);
        int node_local = dense, cp = 0;
        if (mul > 1) { node_local = dense / mul; cp = dense - node_local * mul; }
        const int node = __builtin_amdgcn_readfirstlane(s_nbase[seg]) + node_local;
        const float4 a = w.n_a[node], b = w.n_b[node];
        if (cp != 0 && f2i(b.w) >= 0) continue;                          // a node with one copy in a level of bundles
        const float4 lt = (L == 1) ? light0 : sc.lights[l];
        if (!faces_light(a, b, lt)) continue;                            // the host zeroed the counts for this layout
        ++n_rays;
        if (sample_unblocked(node, false, 0u, cp, l, j, a, lt)) atomicAdd(&out[(__builtin_amdgcn_readfirstlane(s_sbase[seg]) + cp * __builtin_amdgcn_readfirstlane(s_cstride[seg]) + node_local) * L + l], 1);
    }
    if (f.count_stats) block_stat_add(&w.stats[RT_S_SHADOW], n_rays, s_tmp);
}

// raytracer.cpp:173-178: truncate a colour to 0..255 and store it at the pixel of a local slot (row-major image, or
// this rank's tile-major block).  A wavefront's slots that reach here together lie in one tile only in k_finalize.
DEVI void store_pixel(const DevFrame& f, unsigned char* __restrict__ out, int tile_major, int slot, float cr, float cg, float cb) {
    int r = (int)(255.0f * cr), g = (int)(255.0f * cg), b = (int)(255.0f * cb);
    const int tile_local = slot / RT_TILE_PIX, in = slot % RT_TILE_PIX;
    if (tile_major) {
        // a rank's block holds its tiles one after the other, each row-major
        const int o = 3 * (tile_local * RT_TILE_PIX + slot_row(in) * RT_TILE + slot_col(in));
        out[o] = (unsigned char)r; out[o + 1] = (unsigned char)g; out[o + 2] = (unsigned char)b;
    } else {
        int gt = f.local_tiles[tile_local];
        int x = (gt % f.tiles_x) * RT_TILE + slot_col(in), y = (gt / f.tiles_x) * RT_TILE + slot_row(in);
        if (x < f.width && y < f.height) {
            size_t o = ((size_t)y * f.width + x) * 3;
            out[o] = (unsigned char)r; out[o + 1] = (unsigned char)g; out[o + 2] = (unsigned char)b;
        }
    }
}

#if RT_PART_KIND == 0
// ---------------------------------------------------------------------------------------------------------
// shade + resolve for one recursion level, deepest level first: Blinn-Phong at the hit (raytracer.cpp:262-347),
// then the parent's combination of its children's colours with both clamps (raytracer.cpp:436-503).  Level 0
// adds the result to the pixel's running sum (raytracer.cpp:154).
// ---------------------------------------------------------------------------------------------------------
// GROUP: copies a lane has in flight together: 4 for levels whose nodes have copies, 1 (a leaner kernel: more wavefronts
// resident) for levels of one slot per node
template <int GROUP>
__global__ __launch_bounds__(RT_BLOCK, GROUP > 1 ? RT_SHADE_WAVES : RT_SHADE_WAVES + 1) void k_shade(DevScene sc, DevFrame f, DevWork w, int level) {
    __shared__ unsigned long long s_tmp[RT_BLOCK / 64];
    // chunks of RT_BLOCK hits inside the level's regions, as in k_extend
    __shared__ int s_pre[RT_REGIONS + 1], s_cnt[RT_REGIONS], s_per_slot;
    if (threadIdx.x == 0) {
        int acc = 0, n[RT_REGIONS];
#pragma unroll
        for (int rg = 0; rg < RT_REGIONS; ++rg) n[rg] = w.ctr[RT_CTR_NODE(level, rg)];          // unused regions' counters stay 0
        // A level whose nodes have several copies: a lane per node that walks its copies forms the hit's lighting terms
        // once, but waits for its copies' loads group after group; a lane per SLOT repeats those terms for every copy and
        // has nothing to wait for twice.  The second wins while the whole level is one round of the launch's lanes.
        int total = 0;
#pragma unroll
        for (int rg = 0; rg < RT_REGIONS; ++rg) total += min(n[rg], w.node_rcap[level]);
        const int per_slot = (w.slot_mul[level] > 1 && (long long)total * w.slot_mul[level] <= (long long)gridDim.x * RT_BLOCK) ? w.slot_mul[level] : 1;
        s_per_slot = per_slot;
#pragma unroll
        for (int rg = 0; rg < RT_REGIONS; ++rg) {
            const int m = min(n[rg], w.node_rcap[level]) * per_slot;
            s_cnt[rg] = m; s_pre[rg] = acc; acc += (m + RT_BLOCK - 1) / RT_BLOCK;
        }
        s_pre[RT_REGIONS] = acc;
    }
    __syncthreads();
    const int n_chunks = s_pre[RT_REGIONS];
    if ((int)blockIdx.x >= n_chunks) return;
    const int L = sc.n_lights;
    const int* __restrict__ sh = (const int*)w.shadow;
    const V3 cam = mk(f.cam_loc[0], f.cam_loc[1], f.cam_loc[2]);
    const int mul = w.slot_mul[level], mul_child = w.slot_mul[level + 1];
    const int cstride = w.node_rcap[level], cstride_child = w.node_rcap[level + 1];
    const bool slot_lanes = s_per_slot > 1;
    const float inv_gr = 1.0f / (float)f.gr_samples;                     // raytracer.cpp:436-439
    unsigned long long n_shaded = 0;
    int seg = 0;
    for (int g = (int)blockIdx.x; g < n_chunks; g += (int)gridDim.x) {
        while (seg < w.regions - 1 && g >= __builtin_amdgcn_readfirstlane(s_pre[seg + 1])) ++seg;
        const int k = (g - __builtin_amdgcn_readfirstlane(s_pre[seg])) * RT_BLOCK + (int)threadIdx.x;
        if (k >= __builtin_amdgcn_readfirstlane(s_cnt[seg])) continue;
        int k_node = k, my_cp = 0;
        if (slot_lanes) {
            if ((mul & (mul - 1)) == 0) { k_node = k >> (31 - __clz(mul)); my_cp = k & (mul - 1); }
            else { k_node = k / mul; my_cp = k - k_node * mul; }
        }
        const int node = w.node_off[level] + seg * w.node_rcap[level] + k_node;
        // slot of the node's copy 0; copy cp sits cstride slots further (a region's slots: copy by copy, each node by node)
        const int slot0 = w.slot_off[level] + seg * mul * cstride + k_node;
        const bool has_children = level < f.max_depth;
        float4 a = w.n_a[node], b = w.n_b[node];
        const float4 c = (has_children || f.soft_shadows) ? w.n_c[node] : make_float4(i2f(-1), 0.0f, 0.0f, 0.0f);   // (not written otherwise, see k_extend)
        const int gid = f2i(b.w) & 0x7fffffff;
        float4 m0 = sc.mats[3 * gid], m1 = sc.mats[3 * gid + 1], m2 = sc.mats[3 * gid + 2];
        V3 p = mk(a.x, a.y, a.z), N = mk(b.x, b.y, b.z);
        const float ka = 0.25f, shiny = m0.w;
        V3 mat = mk(m0.x, m0.y, m0.z);
        const int tex = f2i(m2.y);
        if (f.textures && tex >= 0) {                                    // raytracer.cpp:271-297
            float2 uv = w.n_uv[node];
            int4 td = sc.tex_desc[tex];
            float u = fmodf(uv.x, 1.0f); if (uv.x < 0) u += 1.0f;
            float v = fmodf(uv.y, 1.0f); if (uv.y < 0) v += 1.0f;
            int xi = (int)(u * (float)(td.x - 1)); if (td.x - 1 < xi) xi = td.x - 1;
            int yi = (int)((1.0f - v) * (float)(td.y - 1)); if (td.y - 1 < yi) yi = td.y - 1;
            if (xi < 0) xi = 0;                                          // the reference reads out of bounds here
            if (yi < 0) yi = 0;
            float4 tx = sc.texels[td.z + yi * td.x + xi];
            mat = mk(tx.x, tx.y, tx.z);
        }
        // what the node's children contribute (raytracer.cpp:369-503): the same structure for every copy
        const int first = f2i(c.x), packed = f2i(c.w);
        const bool bundle = (packed >> 29) & 1;
        const int n_reflect = bundle ? (packed & 0x1fff) : (packed & 0x1fffffff);
        const int copies = bundle ? ((packed >> 13) & 0xffff) : 1;
        if (my_cp >= copies) continue;                                   // (a node with one copy in a level of several slots per node)
        const bool refract = (packed >> 30) & 1;
        const float reflectivity = m2.z;
        n_shaded += slot_lanes ? 1u : (unsigned)copies;
        // slot of copy 0 of a child node (children live one level down)
        auto child_slot = [&](int ch) {
            const int x = ch - w.node_off[level + 1];
            if (mul_child <= 1) return w.slot_off[level + 1] + x;
            const int rg_c = x / cstride_child;
            return w.slot_off[level + 1] + rg_c * mul_child * cstride_child + (x - rg_c * cstride_child);
        };
        int refl_slot = -1, refr_slot = -1;                              // bundles and single reflections: one child node
        if (has_children) {
            if ((n_reflect == 1 || bundle) && n_reflect > 0) { const int ch = w.link[first]; if (ch >= 0) refl_slot = child_slot(ch); }
            if (refract) { const int ch = w.link[first + (bundle ? (n_reflect ? 1 : 0) : n_reflect)]; if (ch >= 0) refr_slot = child_slot(ch); }
        }

        const V3 V = normalise(cam - p);
        // The shading of a hit (raytracer.cpp:262-347) depends on the copy only through the shadow factor: per light the
        // two terms are formed once, the copies' colours are built up in their slots (one pass for the usual one light).
        for (int l = 0; l < (L > 0 ? L : 1); ++l) {
            V3 diff = mk(0.0f, 0.0f, 0.0f), spec = mk(0.0f, 0.0f, 0.0f); float N_L = 0.0f, irr = 0.0f;
            if (L > 0) {
                float4 lt = sc.lights[l];
                V3 Lv = mk(lt.x - p.x, lt.y - p.y, lt.z - p.z);
                float L_len = len_ref(Lv);
                Lv = normalise(Lv);
                irr = (float)((double)lt.w / ((double)1.0f + (double)L_len * (double)L_len));   // raytracer.cpp:320
                V3 H = normalise(Lv + V);
                N_L = smax(0.0f, dot(N, Lv));
                float N_H = smax(0.0f, dot(N, H));
                float pw = powf_ref(N_H, shiny);
                diff = mat; spec = mk(0.5f * m1.x * pw * irr, 0.5f * m1.y * pw * irr, 0.5f * m1.z * pw * irr);
            }
            const bool first_light = l == 0, last_light = l >= L - 1;
            // copies four at a time: everything a group reads is loaded before anything is stored, so the loads of a
            // group are in flight together (a lane with 16 copies otherwise waits for sixteen round trips in a row)
            // (a lane per slot: this lane's copy only)
            const int cp_end = slot_lanes ? my_cp + 1 : copies;
            for (int cp0 = my_cp; cp0 < cp_end; cp0 += GROUP) {
                int cnt4[GROUP]; float4 own4[GROUP], refl4[GROUP], refr4[GROUP];
#pragma unroll
                for (int u = 0; u < GROUP; ++u) {
                    const int cp = cp0 + u, slot = slot0 + cp * cstride;
                    const bool on = cp < cp_end;
                    cnt4[u] = (on && L > 0) ? sh[slot * L + l] : 0;
                    own4[u] = (on && !first_light) ? w.n_col[slot] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    refl4[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); refr4[u] = refl4[u];
                    if (on && last_light && has_children) {
                        if (n_reflect == 1 && refl_slot >= 0) refl4[u] = w.n_col[refl_slot + cp * cstride_child];
                        if (refr_slot >= 0) refr4[u] = w.n_col[refr_slot + cp * cstride_child];
                    }
                }
#pragma unroll
                for (int u = 0; u < GROUP; ++u) {
                    const int cp = cp0 + u, slot = slot0 + cp * cstride;
                    if (cp >= cp_end) break;
                    V3 col = first_light ? mk(ka * mat.x, ka * mat.y, ka * mat.z) : mk(own4[u].x, own4[u].y, own4[u].z);
                    if (L > 0) {
                        const int cnt = cnt4[u];
                        const float shadow = f.soft_shadows ? (float)cnt / (float)f.ss_samples : (cnt ? 1.0f : 0.0f);
                        col.x += shadow * diff.x * N_L * irr + spec.x;
                        col.y += shadow * diff.y * N_L * irr + spec.y;
                        col.z += shadow * diff.z * N_L * irr + spec.z;
                    }
                    if (last_light) {
                        col.x = smin(1.0f, smax(col.x, 0.0f));           // raytracer.cpp:342-344
                        col.y = smin(1.0f, smax(col.y, 0.0f));
                        col.z = smin(1.0f, smax(col.z, 0.0f));
                        if (has_children) {
                            V3 rc = mk(0.0f, 0.0f, 0.0f);
                            if (n_reflect == 1) {
                                // copy cp of the child under copy cp of the parent
                                const V3 cc = mk(refl4[u].x, refl4[u].y, refl4[u].z);
                                for (int s2 = 0; s2 < f.gr_samples; ++s2) rc = rc + cc;     // the gr_samples identical rays of raytracer.cpp:400-432
                            } else if (bundle) {
                                // one traced ray whose node holds this hit's n_reflect children as its copies
                                for (int s2 = 0; s2 < n_reflect; ++s2) {
                                    if (refl_slot >= 0) { const float4 q = w.n_col[refl_slot + (cp * n_reflect + s2) * cstride_child]; rc = rc + mk(q.x, q.y, q.z); }
                                    else rc = rc + mk(0.0f, 0.0f, 0.0f);
                                }
                            } else {
                                for (int s2 = 0; s2 < n_reflect; ++s2) {
                                    const int ch = w.link[first + s2];
                                    if (ch >= 0) { const float4 q = w.n_col[child_slot(ch)]; rc = rc + mk(q.x, q.y, q.z); }
                                    else rc = rc + mk(0.0f, 0.0f, 0.0f);
                                }
                            }
                            rc = rc * inv_gr;
                            rc = rc * reflectivity;
                            col = col + rc;
                            if (refract) col = col + mk(refr4[u].x, refr4[u].y, refr4[u].z) * c.y;     // raytracer.cpp:494-497
                            col.x = smin(1.0f, smax(0.0f, col.x));       // raytracer.cpp:501-503
                            col.y = smin(1.0f, smax(0.0f, col.y));
                            col.z = smin(1.0f, smax(0.0f, col.z));
                        }
                        if (level == 0) {
                            const int pslot = w.n_pix[node];
                            if (f.direct_out) {
                                // one sample per pixel in one batch: the sum is this colour and the average divides by one, so the
                                // pixel can be written now (the frame start zeroed the image instead of the sums)
                                store_pixel(f, f.direct_out, f.direct_tile_major, pslot, col.x * 1.0f, col.y * 1.0f, col.z * 1.0f);
                            } else {
                                float4 acc = w.accum[pslot];
                                w.accum[pslot] = make_float4(acc.x + col.x, acc.y + col.y, acc.z + col.z, 0.0f);
                            }
                        }
                    }
                    if (!(last_light && level == 0)) w.n_col[slot] = make_float4(col.x, col.y, col.z, 0.0f);
                }
            }
        }
    }
    if (f.count_stats) block_stat_add(&w.stats[RT_S_NODES], n_shaded, s_tmp);
}

#endif  // RT_PART_KIND == 0

// ---------------------------------------------------------------------------------------------------------
// Frames without recursion and with hard shadows - the reference's default render and its multi-sample flag sets
// without -r / -ss - in one kernel: a lane generates its primary ray, traces it, sends one shadow ray per light from
// the hit (raytracer.cpp:509-530), shades (raytracer.cpp:262-347) and adds the colour to its pixel.  No hit record,
// shadow count or pixel slot goes through memory and the frame is one launch instead of three; the arithmetic is that
// of k_extend, k_shadow (hard shadows) and k_shade for a hit without children, statement for statement.
// ---------------------------------------------------------------------------------------------------------
template <int SUB, bool LDSP, bool STATS>
__global__ __launch_bounds__(RT_BLOCK, SUB ? RT_EXTEND_WAVES_SUB : RT_EXTEND_WAVES) void k_primary_shaded(DevScene sc_arg, DevFrame f, DevWork w) {
    extern __shared__ float4 s_mem[];
    DevScene sc = sc_arg;
    float4* s_nodes = s_mem;
    const int4* s_leaves = SUB ? sc.leaves : (const int4*)(s_mem + 2 * sc.n_nodes);
    int* s_stack = (int*)(s_mem + 2 * sc.n_nodes) + threadIdx.x;                      // [depth][thread], only with SUB
    __shared__ unsigned long long s_tmp[RT_BLOCK / 64];
    const int n_chunks = (f.pix_count + RT_BLOCK - 1) / RT_BLOCK;
    if ((int)blockIdx.x >= n_chunks) return;
    if (f.use_bvh) stage_bvh(sc, s_nodes, (int4*)(s_mem + 2 * sc.n_nodes), !SUB);
    if (LDSP) stage_prims(sc, s_mem + 2 * sc.n_nodes + 2 * sc.n_leaves);
    const int L = sc.n_lights;
    const V3 cam = mk(f.cam_loc[0], f.cam_loc[1], f.cam_loc[2]);
    unsigned n_primary = 0, n_shadow = 0, n_hits = 0;                  // per thread: one ray per round
    for (int g = (int)blockIdx.x; g < n_chunks; g += (int)gridDim.x) {
        const int i = g * RT_BLOCK + (int)threadIdx.x;
        bool active = i < f.pix_count;
        Ray r; uint32_t path = 0; int slot = 0;
        r.o = mk(0, 0, 0); r.d = mk(0, 0, 1); r.time = 0.0f;
        if (active) active = generate_primary(f, i, r, path, slot);
        Best best; best.t = 0.0f; best.cls = 0; best.idx = 0; best.fileset = 0;
        const bool hit = active && trace<false, SUB, false, false>(sc, f, s_nodes, s_leaves, s_stack, r, false, best);
        if (STATS) { n_primary += active ? 1u : 0u; n_hits += hit ? 1u : 0u; }
        if (!hit) continue;
        HitOut h;
        full_hit<SUB>(sc, f, r, best, h);
        const V3 p = h.p, N = h.n;
        const float4 m0 = sc.mats[3 * h.gid], m1 = sc.mats[3 * h.gid + 1], m2 = sc.mats[3 * h.gid + 2];
        const float ka = 0.25f, shiny = m0.w;
        V3 mat = mk(m0.x, m0.y, m0.z);
        const int tex = f2i(m2.y);
        if (f.textures && tex >= 0) {                                    // raytracer.cpp:271-297
            const int4 td = sc.tex_desc[tex];
            float u = fmodf(h.u, 1.0f); if (h.u < 0) u += 1.0f;
            float v = fmodf(h.v, 1.0f); if (h.v < 0) v += 1.0f;
            int xi = (int)(u * (float)(td.x - 1)); if (td.x - 1 < xi) xi = td.x - 1;
            int yi = (int)((1.0f - v) * (float)(td.y - 1)); if (td.y - 1 < yi) yi = td.y - 1;
            if (xi < 0) xi = 0;
            if (yi < 0) yi = 0;
            const float4 tx = sc.texels[td.z + yi * td.x + xi];
            mat = mk(tx.x, tx.y, tx.z);
        }
        const V3 V = normalise(cam - p);
        V3 col = mk(ka * mat.x, ka * mat.y, ka * mat.z);
        for (int l = 0; l < L; ++l) {
            const float4 lt = sc.lights[l];
            // the shadow ray (k_shadow, hard shadows): not traced where the surface clearly faces away from the light
            bool unblocked = false;
            {
                const V3 Lr = mk(lt.x - p.x, lt.y - p.y, lt.z - p.z);
                const float d = dot(N, Lr);
                const bool faces = !f.skip_unlit || !(d < 0.0f && d * d > 1e-6f * dot(N, N) * dot(Lr, Lr));
                if (faces) {
                    Ray sr; sr.time = r.time;
                    sr.d = normalise(Lr);
                    sr.o = p + N * 1e-2f;                                // raytracer.cpp:515: unflipped geometric normal
                    Best sb; sb.t = 0.0f; sb.cls = 0; sb.idx = 0; sb.fileset = 0;
                    if (STATS) ++n_shadow;
                    unblocked = !trace<true, SUB>(sc, f, s_nodes, s_leaves, s_stack, sr, false, sb);
                }
            }
            V3 Lv = mk(lt.x - p.x, lt.y - p.y, lt.z - p.z);
            const float L_len = len_ref(Lv);
            Lv = normalise(Lv);
            const float irr = (float)((double)lt.w / ((double)1.0f + (double)L_len * (double)L_len));   // raytracer.cpp:320
            const V3 H = normalise(Lv + V);
            const float N_L = smax(0.0f, dot(N, Lv));
            const float N_H = smax(0.0f, dot(N, H));
            const float pw = powf_ref(N_H, shiny);
            const V3 spec = mk(0.5f * m1.x * pw * irr, 0.5f * m1.y * pw * irr, 0.5f * m1.z * pw * irr);
            const float shadow = unblocked ? 1.0f : 0.0f;
            col.x += shadow * mat.x * N_L * irr + spec.x;
            col.y += shadow * mat.y * N_L * irr + spec.y;
            col.z += shadow * mat.z * N_L * irr + spec.z;
        }
        col.x = smin(1.0f, smax(col.x, 0.0f));                           // raytracer.cpp:342-344
        col.y = smin(1.0f, smax(col.y, 0.0f));
        col.z = smin(1.0f, smax(col.z, 0.0f));
        if (f.direct_out) store_pixel(f, f.direct_out, f.direct_tile_major, slot, col.x * 1.0f, col.y * 1.0f, col.z * 1.0f);
        else {
            const float4 acc = w.accum[slot];
            w.accum[slot] = make_float4(acc.x + col.x, acc.y + col.y, acc.z + col.z, 0.0f);
        }
    }
    if (!STATS) return;
    block_stat_add(&w.stats[RT_S_PRIMARY], (unsigned long long)n_primary, s_tmp);
    block_stat_add(&w.stats[RT_S_SHADOW], (unsigned long long)n_shadow, s_tmp);
    block_stat_add(&w.stats[RT_S_SHADOW_REF], (unsigned long long)n_hits * (unsigned long long)L, s_tmp);
    block_stat_add(&w.stats[RT_S_NODES], (unsigned long long)n_hits, s_tmp);
}

#if RT_PART_KIND == 0
// Secondary rays of crowded scenes, re-coalesced before they are traced.  The rays a level queues sit in the order their
// parents were traced; neighbouring lanes then start at unrelated points and walk unrelated parts of the second-level
// hierarchies (a wavefront of such rays costs two to three times one of neighbouring rays).  This kernel gives every
// queue slot of `level` a key - region | cell of the origin (Morton code of a 256^3 grid over the scene) | octant of the
// direction - and one radix sort of (key, slot) per level puts rays that start close together and head the same way
// side by side; k_extend then takes its k-th ray from slot perm[k].  (Other arrangements of the key's fields - octant
// first, octant between coarse and fine cell bits - measured within 2 % of this one.)  Regions stay intact (the key's top bits), slots past a region's count sort
// behind its rays.  Only the order of tracing changes: values never depend on it.
__global__ __launch_bounds__(RT_BLOCK) void k_sort_keys(DevScene sc, DevWork w, int level, unsigned* __restrict__ keys, unsigned* __restrict__ slots) {
    const int rcap = w.link_rcap[level];
    const long long n = (long long)w.regions * rcap;
    const float4* __restrict__ qa = w.q_a[level & 1];
    const float4* __restrict__ qb = w.q_b[level & 1];
    const float lo_x = sc.center[0] - sc.radius, lo_y = sc.center[1] - sc.radius, lo_z = sc.center[2] - sc.radius;
    const float scale = 256.0f / fmaxf(2.0f * sc.radius, 1e-30f);
    auto spread = [](unsigned v) -> unsigned {            // 8 bits -> every third bit
        v = (v | (v << 16)) & 0x0300F00Fu; v = (v | (v << 8)) & 0x0300F00Fu; v = (v | (v << 4)) & 0x030C30C3u; v = (v | (v << 2)) & 0x09249249u;
        return v;
    };
    for (long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (long long)gridDim.x * blockDim.x) {
        const int rg = (int)(s / rcap), local = (int)(s - (long long)rg * rcap);
        unsigned key = ((unsigned)rg << 27) | 0x07FFFFFFu;
        if (local < min(w.ctr[RT_CTR_LINK(level, rg)], rcap)) {
            const float4 a = qa[s], b = qb[s];
            auto cell = [&](float o, float lo) -> unsigned { const float q = (o - lo) * scale; return q > 0.0f ? (q < 255.0f ? (unsigned)q : 255u) : 0u; };   // NaN -> 0
            const unsigned m = spread(cell(a.x, lo_x)) | (spread(cell(a.y, lo_y)) << 1) | (spread(cell(a.z, lo_z)) << 2);   // 24 bits
            const unsigned oct = (a.w < 0.0f ? 1u : 0u) | (b.x < 0.0f ? 2u : 0u) | (b.y < 0.0f ? 4u : 0u);
            key = ((unsigned)rg << 27) | (m << 3) | oct;
            if (key == (((unsigned)rg << 27) | 0x07FFFFFFu)) --key;              // the one key reserved for empty slots
        }
        keys[s] = key; slots[s] = (unsigned)s;
    }
}

// raytracer.cpp:171-178: average over the samples and truncate to 0..255.
__global__ void k_finalize(DevFrame f, DevWork w, unsigned char* __restrict__ out, int tile_major, int n_slots, int samples) {
    const float inv = 1.0f / (float)samples;
    for (int slot = blockIdx.x * blockDim.x + threadIdx.x; slot < n_slots; slot += gridDim.x * blockDim.x) {
        float4 acc = w.accum[slot];
        store_pixel(f, out, tile_major, slot, acc.x * inv, acc.y * inv, acc.z * inv);
    }
}

// Gathered per-rank tile blocks -> row-major image.  Tile (tx,ty) belongs to rank (tx+ty) % world and is that
// rank's k-th tile in ascending global tile order.
__global__ void k_assemble(const unsigned char* __restrict__ tiles, int world, int max_tiles, int tiles_x, int tiles_y,
                           int width, int height, unsigned char* __restrict__ out) {
    const int n = width * height;
    for (int pix = blockIdx.x * blockDim.x + threadIdx.x; pix < n; pix += gridDim.x * blockDim.x) {
        int x = pix % width, y = pix / width;
        int tx = x / RT_TILE, ty = y / RT_TILE;
        int owner = (tx + ty) % world;
        // tiles of `owner` before (tx,ty): full rows above plus this row's tiles left of tx
        int k = 0;
        for (int row = 0; row < ty; ++row) {
            int first = ((owner - row) % world + world) % world;         // smallest tx in that row owned by `owner`
            if (first < tiles_x) k += (tiles_x - 1 - first) / world + 1;
        }
        {
            int first = ((owner - ty) % world + world) % world;
            if (tx > first) k += (tx - 1 - first) / world + 1;
        }
        size_t src = (((size_t)owner * max_tiles + k) * RT_TILE_PIX + (size_t)(y % RT_TILE) * RT_TILE + (x % RT_TILE)) * 3;
        size_t dst = (size_t)pix * 3;
        out[dst] = tiles[src]; out[dst + 1] = tiles[src + 1]; out[dst + 2] = tiles[src + 2];
    }
}

#endif  // RT_PART_KIND == 0

} // namespace


// ---------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------
// Per-part launchers: the part that instantiates a family of kernels exports the function that launches them; the
// public launchers of rt_types.h (part 0) work out the dynamic LDS size and pick the part by the scene's SUB value.
void rt_launch_extend_part0(hipStream_t s, const DevScene& sc, const DevFrame& f, const DevWork& w, int level, int grid, size_t lds, bool ldsp);
void rt_launch_extend_part1(hipStream_t s, const DevScene& sc, const DevFrame& f, const DevWork& w, int level, int grid, size_t lds, bool ldsp);
void rt_launch_extend_part2(hipStream_t s, const DevScene& sc, const DevFrame& f, const DevWork& w, int level, int grid, size_t lds, bool ldsp);
void rt_launch_fused_part0(hipStream_t s, const DevScene& sc, const DevFrame& f, const DevWork& w, int grid, size_t lds, bool ldsp);
void rt_launch_fused_part1(hipStream_t s, const DevScene& sc, const DevFrame& f, const DevWork& w, int grid, size_t lds, bool ldsp);
void rt_launch_fused_part2(hipStream_t s, const DevScene& sc, const DevFrame& f, const DevWork& w, int grid, size_t lds, bool ldsp);
void rt_launch_shadow_part0(hipStream_t s, const DevScene& sc, const DevFrame& f, const DevWork& w, int lv_begin, int lv_end, int grid, size_t lds, bool ldsp, int mode, int phase, int sq_k);
void rt_launch_shadow_part1(hipStream_t s, const DevScene& sc, const DevFrame& f, const DevWork& w, int lv_begin, int lv_end, int grid, size_t lds, bool ldsp, int mode, int phase, int sq_k);
void rt_launch_shadow_part2(hipStream_t s, const DevScene& sc, const DevFrame& f, const DevWork& w, int lv_begin, int lv_end, int grid, size_t lds, bool ldsp, int mode, int phase, int sq_k);

#if RT_PART_KIND == 1
void RT_CAT(rt_launch_extend_part, RT_PART_SUB)(hipStream_t s, const DevScene& sc, const DevFrame& f, const DevWork& w, int level, int grid, size_t lds, bool ldsp) {
    constexpr int SUB = RT_PART_SUB;
#define RT_EXTEND_LAUNCH(PRIMARY, LDSP) \
    do { if (f.count_stats) hipLaunchKernelGGL((k_extend<PRIMARY, SUB, LDSP, true>), dim3(grid), dim3(RT_BLOCK), lds, s, sc, f, w, level); \
         else hipLaunchKernelGGL((k_extend<PRIMARY, SUB, LDSP, false>), dim3(grid), dim3(RT_BLOCK), lds, s, sc, f, w, level); } while (0)
    if (level == 0) {
        if (SUB == 0 && ldsp) RT_EXTEND_LAUNCH(true, (SUB == 0));
        else RT_EXTEND_LAUNCH(true, false);
    } else {
        if (SUB == 0 && ldsp) RT_EXTEND_LAUNCH(false, (SUB == 0));
        else RT_EXTEND_LAUNCH(false, false);
    }
#undef RT_EXTEND_LAUNCH
}
#endif

#if RT_PART_KIND == 2
void RT_CAT(rt_launch_fused_part, RT_PART_SUB)(hipStream_t s, const DevScene& sc, const DevFrame& f, const DevWork& w, int grid, size_t lds, bool ldsp) {
    constexpr int SUB = RT_PART_SUB;
#define RT_FUSED_LAUNCH(LDSP) \
    do { if (f.count_stats) hipLaunchKernelGGL((k_primary_shaded<SUB, LDSP, true>), dim3(grid), dim3(RT_BLOCK), lds, s, sc, f, w); \
         else hipLaunchKernelGGL((k_primary_shaded<SUB, LDSP, false>), dim3(grid), dim3(RT_BLOCK), lds, s, sc, f, w); } while (0)
    if (SUB == 0 && ldsp) RT_FUSED_LAUNCH((SUB == 0));
    else RT_FUSED_LAUNCH(false);
#undef RT_FUSED_LAUNCH
}
#endif

#if RT_PART_KIND == 3
void RT_CAT(rt_launch_shadow_part, RT_PART_SUB)(hipStream_t s, const DevScene& sc, const DevFrame& f, const DevWork& w, int lv_begin, int lv_end, int grid, size_t lds, bool ldsp, int mode, int phase, int sq_k) {
    constexpr int SUB = RT_PART_SUB;
#define RT_SHADOW_LAUNCH_P(LDSP, MODE, PHASE) hipLaunchKernelGGL((k_shadow<SUB, LDSP, MODE, PHASE>), dim3(grid), dim3(RT_BLOCK), lds, s, sc, f, w, lv_begin, lv_end, sq_k)
    // (the phases of the queue of pairs exist for soft shadows of scenes without second-level hierarchies only)
#define RT_SHADOW_LAUNCH_M(LDSP, MODE) \
    do { if (SUB == 0 && MODE != 0 && phase == 1) RT_SHADOW_LAUNCH_P(LDSP, MODE, (SUB == 0 && MODE != 0 ? 1 : 0)); \
         else if (SUB == 0 && MODE != 0 && phase == 2) RT_SHADOW_LAUNCH_P(LDSP, MODE, (SUB == 0 && MODE != 0 ? 2 : 0)); \
         else RT_SHADOW_LAUNCH_P(LDSP, MODE, 0); } while (0)
#define RT_SHADOW_LAUNCH(LDSP) \
    do { if (mode == 2) RT_SHADOW_LAUNCH_M(LDSP, 2); else if (mode == 1) RT_SHADOW_LAUNCH_M(LDSP, 1); else RT_SHADOW_LAUNCH_M(LDSP, 0); } while (0)
    if (SUB == 0 && ldsp) RT_SHADOW_LAUNCH((SUB == 0));
    else RT_SHADOW_LAUNCH(false);
#undef RT_SHADOW_LAUNCH_P
#undef RT_SHADOW_LAUNCH_M
#undef RT_SHADOW_LAUNCH
}
#endif

#if RT_PART_KIND == 0
// Start of a frame: zero the per-pixel sums (or, for frames that write their pixels directly, the image), the first
// batch's counters and (optionally) the ray statistics in one launch instead of three fills.
__global__ void k_begin_frame(uint4* __restrict__ zero, size_t n_words, int tail_bytes, int* __restrict__ ctr, unsigned long long* __restrict__ stats) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride) zero[i] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0) {
        if ((int)threadIdx.x < tail_bytes) ((unsigned char*)(zero + n_words))[threadIdx.x] = 0;      // < 16 bytes past the last word
        for (int i = threadIdx.x; i < RT_REGIONS * RT_CTR_WORDS_IN_USE; i += blockDim.x)        // the words in use
            ctr[(i / RT_CTR_WORDS_IN_USE) * RT_CTR_REGION_INTS + i % RT_CTR_WORDS_IN_USE] = 0;
        if (stats) for (int i = threadIdx.x; i < RT_S_TOTAL; i += blockDim.x) stats[i] = 0ull;
    }
}

// LDS per workgroup: the first-level nodes and leaf records (BVH mode) and, when any group of primitives has a
// second-level hierarchy, one traversal stack per thread.  The kernels index from n_nodes / n_leaves, which the
// context zeroes when no BVH is uploaded.
// file_set: the launch may walk the hierarchy of the whole file-order set (-u, or the retry of secondary rays), which is
// deeper than any leaf's and needs the longer stack.
static size_t bvh_lds_bytes(const DevScene& sc, const DevFrame& f, bool file_set) {
    size_t b = (size_t)(2 * sc.n_nodes) * 16;
    if (sc.sub_nodes) b += (size_t)RT_BLOCK * (size_t)(file_set ? sc.sub_stack_entries : sc.sub_stack_entries_leaves) * sizeof(int);
    else b += (size_t)(2 * sc.n_leaves) * 16;
    if (sc.lds_prims) b += (size_t)rt_lds_prim_words(sc) * 16;
    return b ? b : 16;
}

// SUB of the kernels that render this scene: 0 = no second-level hierarchy, 1 = nodes of 4 words, 2 = of 8 (boxes per time)
static int scene_sub(const DevScene& sc) { return !sc.sub_nodes ? 0 : (sc.sub_stride == 8 ? 2 : 1); }

void rt_launch_extend(hipStream_t s, const DevScene& sc, const DevFrame& f, const DevWork& w, int level, int grid) {
    const size_t lds = bvh_lds_bytes(sc, f, !f.use_bvh || (level > 0 && f.fallback_full));
    const int sub = scene_sub(sc);
    const bool ldsp = !sub && sc.lds_prims > 1;
    if (sub == 2) rt_launch_extend_part2(s, sc, f, w, level, grid, lds, false);
    else if (sub == 1) rt_launch_extend_part1(s, sc, f, w, level, grid, lds, false);
    else rt_launch_extend_part0(s, sc, f, w, level, grid, lds, ldsp);
}

void rt_launch_primary_shaded(hipStream_t s, const DevScene& sc, const DevFrame& f, const DevWork& w, int grid) {
    const size_t lds = bvh_lds_bytes(sc, f, !f.use_bvh);
    const int sub = scene_sub(sc);
    const bool ldsp = !sub && sc.lds_prims > 1;
    if (sub == 2) rt_launch_fused_part2(s, sc, f, w, grid, lds, false);
    else if (sub == 1) rt_launch_fused_part1(s, sc, f, w, grid, lds, false);
    else rt_launch_fused_part0(s, sc, f, w, grid, lds, ldsp);
}

int rt_launch_shadow(hipStream_t s, const DevScene& sc, const DevFrame& f, const DevWork& w, int lv_begin, int lv_end, int grid, int grid_drain) {
    const size_t lds = bvh_lds_bytes(sc, f, !f.use_bvh);
    const int mode = !f.soft_shadows ? 0 : (f.fast_samples ? 2 : 1);  // hard shadows are deterministic: always exact
    const int sub = scene_sub(sc);
    const bool ldsp = !sub && sc.lds_prims != 0;
    if (sub == 2) rt_launch_shadow_part2(s, sc, f, w, lv_begin, lv_end, grid, lds, false, mode, 0, 0);
    else if (sub == 1) rt_launch_shadow_part1(s, sc, f, w, lv_begin, lv_end, grid, lds, false, mode, 0, 0);
    else if (w.sq && mode != 0 && grid_drain > 0) {
        // (the context provides the queue where it pays: soft shadows of tiny scenes with the cone rules on, sample counts
        // that tile a wavefront).  Launches of one batch may run side by side: each has a queue of its own.
        const int sq_k = lv_begin < RT_SHADOWQ_LAUNCHES - 1 ? lv_begin : RT_SHADOWQ_LAUNCHES - 1;
        rt_launch_shadow_part0(s, sc, f, w, lv_begin, lv_end, grid, lds, ldsp, mode, 1, sq_k);
        rt_launch_shadow_part0(s, sc, f, w, lv_begin, lv_end, grid_drain, lds, ldsp, mode, 2, sq_k);
        return 2;
    } else rt_launch_shadow_part0(s, sc, f, w, lv_begin, lv_end, grid, lds, ldsp, mode, 0, 0);
    return 1;
}

void rt_launch_sort_keys(hipStream_t s, const DevScene& sc, const DevWork& w, int level, unsigned* keys, unsigned* slots) {
    const long long n = (long long)w.regions * w.link_rcap[level];
    long long grid = (n + RT_BLOCK - 1) / RT_BLOCK; if (grid > 8192) grid = 8192; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_sort_keys, dim3((unsigned)grid), dim3(RT_BLOCK), 0, s, sc, w, level, keys, slots);
}

void rt_launch_shade(hipStream_t s, const DevScene& sc, const DevFrame& f, const DevWork& w, int level, int grid) {
    if (w.slot_mul[level] > 1) hipLaunchKernelGGL(k_shade<4>, dim3(grid), dim3(RT_BLOCK), 0, s, sc, f, w, level);
    else hipLaunchKernelGGL(k_shade<1>, dim3(grid), dim3(RT_BLOCK), 0, s, sc, f, w, level);
}

void rt_launch_begin_frame(hipStream_t s, const DevWork& w, void* zero, size_t zero_bytes, bool zero_stats) {
    const size_t n_words = zero_bytes / 16;                 // `zero` is 16-byte aligned
    size_t grid = (n_words + RT_BLOCK - 1) / RT_BLOCK; if (grid > 4096) grid = 4096; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_begin_frame, dim3((unsigned)grid), dim3(RT_BLOCK), 0, s, (uint4*)zero, n_words, (int)(zero_bytes % 16), w.ctr,
                       zero_stats ? w.stats : (unsigned long long*)nullptr);
}

void rt_launch_finalize(hipStream_t s, const DevFrame& f, const DevWork& w, unsigned char* out, int tile_major, int n_slots, int samples) {
    int grid = (n_slots + RT_BLOCK - 1) / RT_BLOCK; if (grid > 4096) grid = 4096; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_finalize, dim3(grid), dim3(RT_BLOCK), 0, s, f, w, out, tile_major, n_slots, samples);
}

void rt_launch_assemble(hipStream_t s, const unsigned char* tiles, int world, int max_tiles, int tiles_x, int tiles_y,
                        int width, int height, unsigned char* out) {
    int n = width * height;
    int grid = (n + RT_BLOCK - 1) / RT_BLOCK; if (grid > 8192) grid = 8192; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_assemble, dim3(grid), dim3(RT_BLOCK), 0, s, tiles, world, max_tiles, tiles_x, tiles_y, width, height, out);
}
#endif  // RT_PART_KIND == 0
