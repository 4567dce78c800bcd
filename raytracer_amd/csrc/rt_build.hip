// On-device construction of the second-level hierarchies.
//
// The reference's tree stops at depth 8 (bvh.cpp:109-117), so every leaf of a large scene - and the whole scene for
// -u (raytracer.cpp:202-256) - is a long list that the traversal kernels search through a binary hierarchy of their
// own.  Building those hierarchies is sort-shaped, HBM-bound work:
//   1. one lane per primitive: true bounds (everything the primitive can occupy for ray times in [0,1], enlarged for
//      rounding and for what the reference's float sphere test reports outside the geometry);
//   2. every group becomes a perfect binary tree over an ordering of its items: node j of level l owns the items at
//      positions [j*n >> l, (j+1)*n >> l), the 2^depth runs at the bottom hold 1..4 items.  Nodes are stored in heap
//      order from the group's node_base (2^depth slots are reserved, one stays unused), run j's entries in the four
//      slots from 4 * (node_base + j): the traversal computes both addresses from a node's index alone.  The ordering is made level
//      by level, top down: the centres of each node's items are bounded (atomics per node), and one radix sort over
//      all items of all groups (rocPRIM; key = node << 32 | centre along the node's widest axis) puts the lower half
//      of every node first - a median split of every node of the level at once;
//   3. bottom up: the lowest interior level unions item boxes, each level above the two boxes stored in its children.
// The tree shape is fixed by the item count alone, so its depth is known before anything is built
// (<= 22 for 16M primitives; the traversal stack holds RT_SUB_STACK entries).
// Nothing here influences a rendered value: the hierarchy only decides which primitives a ray may skip.
#include "rt_build.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

#define DEVI __device__ __forceinline__

DEVI int f2i(float f) { return __float_as_int(f); }
DEVI float i2f(int i) { return __int_as_float(i); }

// floats as unsigned integers that order the same way (for atomicMin / atomicMax)
DEVI unsigned enc_ordered(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
inline float dec_ordered(unsigned u) { u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u; float f; std::memcpy(&f, &u, 4); return f; }

struct DBox { float mn[3], mx[3]; };
DEVI void db_clear(DBox& b) { for (int i = 0; i < 3; ++i) { b.mn[i] = INFINITY; b.mx[i] = -INFINITY; } }
// NaN coordinates are ignored, exactly like std::min / std::max with the accumulator first
DEVI void db_add(DBox& b, float x, float y, float z) {
    b.mn[0] = (x < b.mn[0]) ? x : b.mn[0]; b.mx[0] = (b.mx[0] < x) ? x : b.mx[0];
    b.mn[1] = (y < b.mn[1]) ? y : b.mn[1]; b.mx[1] = (b.mx[1] < y) ? y : b.mx[1];
    b.mn[2] = (z < b.mn[2]) ? z : b.mn[2]; b.mx[2] = (b.mx[2] < z) ? z : b.mx[2];
}
// Enlargement per axis that covers the rounding of the reference's own position arithmetic (relative to the box's
// largest coordinate or width).
DEVI void db_slack(const DBox& b, float e[3]) {
    for (int i = 0; i < 3; ++i) {
        float m = fabsf(b.mn[i]), M = fabsf(b.mx[i]), w = b.mx[i] - b.mn[i];
        float big = (m < M) ? M : m; big = (big < w) ? w : big;
        e[i] = 1e-5f * big + 1e-6f;
    }
}
DEVI void db_grow_by(DBox& b, const float e[3]) { for (int i = 0; i < 3; ++i) { b.mn[i] -= e[i]; b.mx[i] += e[i]; } }

// True bounds of one primitive of a set (entry = class << 30 | index): `sw` = everything it occupies for ray times in
// [0,1]; `t0` / `t1` (optional) = what it occupies at time 0 and at time 1.  A primitive moves on a straight line
// (lerp of its two locations, sphere.cpp:123, cube.cpp:131), so the box interpolated between t0 and t1 bounds it at
// any time in between; all three carry the enlargement computed for the swept box.
DEVI void prim_boxes(const DevPrimSet& S, unsigned entry, double sphere_e, DBox& sw, DBox* t0, DBox* t1) {
    const int cls = (int)(entry >> 30), idx = (int)(entry & 0x3fffffffu);
    db_clear(sw);
    if (t0) { db_clear(*t0); db_clear(*t1); }
    float e[3];
    if (cls == 0) {
        const float4 w0 = S.planes[3 * (size_t)idx], w1 = S.planes[3 * (size_t)idx + 1];
        const int first = f2i(w0.w), nv = f2i(w1.w) & 0xffff, drop = f2i(w1.w) >> 16;
        for (int v = 0; v < nv; ++v) {                        // vertices are stored projected: (kept, kept, dropped)
            float4 p = S.verts[first + v];
            if (drop == 0) db_add(sw, p.z, p.x, p.y); else if (drop == 1) db_add(sw, p.x, p.z, p.y); else db_add(sw, p.x, p.y, p.z);
        }
        db_slack(sw, e); db_grow_by(sw, e);
        if (t0) { *t0 = sw; *t1 = sw; }
        return;
    }
    if (cls == 1) {
        const float4* q = S.cubes + 6 * (size_t)idx;
        const float4 w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3];
        const float cx = w2.x, sx = w2.y, cy = w2.z, sy = w2.w, cz = w3.x, sz = w3.y, scale = w0.w;
        for (int m = 0; m < 8; ++m) {
            float x = (m & 1) ? 1.0f : -1.0f, y = (m & 2) ? 1.0f : -1.0f, z = (m & 4) ? 1.0f : -1.0f;
            float y1 = y * cx - z * sx, z1 = y * sx + z * cx;
            float x2 = x * cy + z1 * sy, z2 = -x * sy + z1 * cy;
            float rx = (x2 * cz - y1 * sz) * scale, ry = (x2 * sz + y1 * cz) * scale, rz = z2 * scale;
            db_add(sw, rx + w0.x, ry + w0.y, rz + w0.z);
            db_add(sw, rx + w1.x, ry + w1.y, rz + w1.z);
            if (t0) { db_add(*t0, rx + w0.x, ry + w0.y, rz + w0.z); db_add(*t1, rx + w1.x, ry + w1.y, rz + w1.z); }
        }
        db_slack(sw, e); db_grow_by(sw, e);
        if (t0) { db_grow_by(*t0, e); db_grow_by(*t1, e); }
        return;
    }
    const float4 w0 = S.spheres[2 * (size_t)idx], w1 = S.spheres[2 * (size_t)idx + 1];
    const float r = fabsf(w0.w);
    db_add(sw, w0.x - r, w0.y - r, w0.z - r); db_add(sw, w0.x + r, w0.y + r, w0.z + r);
    db_add(sw, w1.x - r, w1.y - r, w1.z - r); db_add(sw, w1.x + r, w1.y + r, w1.z + r);
    if (t0) {
        db_add(*t0, w0.x - r, w0.y - r, w0.z - r); db_add(*t0, w0.x + r, w0.y + r, w0.z + r);
        db_add(*t1, w1.x - r, w1.y - r, w1.z - r); db_add(*t1, w1.x + r, w1.y + r, w1.z + r);
    }
    db_slack(sw, e);
    if (sphere_e > 0.0) {
        // sphere.cpp:131-139 forms |l|^2 - tca^2 in float: a reported hit lies within sqrt(r^2 + e) of the centre
        const double rd = (double)r;
        const float extra = (float)(sqrt(rd * rd + sphere_e) - rd) * 1.01f;
        if (extra < INFINITY) for (int i = 0; i < 3; ++i) e[i] += extra;
    }
    db_grow_by(sw, e);
    if (t0) { db_grow_by(*t0, e); db_grow_by(*t1, e); }
}

// True bounds of item k of a group (planes, then cubes, then spheres); returns the packed entry.
DEVI unsigned item_box(const DevPrimSet& S, const BuildGroup& g, int k, double sphere_e, DBox& b) {
    unsigned entry;
    if (k < g.pn) entry = (unsigned)(g.p0 + k);
    else if (k < g.pn + g.cn) entry = (1u << 30) | (unsigned)(g.c0 + k - g.pn);
    else entry = (2u << 30) | (unsigned)(g.s0 + k - g.pn - g.cn);
    prim_boxes(S, entry, sphere_e, b, nullptr, nullptr);
    return entry;
}

DEVI float wave_min(float v) { for (int o = 32; o > 0; o >>= 1) { float t = __shfl_xor(v, o); v = (t < v) ? t : v; } return v; }
DEVI float wave_max(float v) { for (int o = 32; o > 0; o >>= 1) { float t = __shfl_xor(v, o); v = (v < t) ? t : v; } return v; }

// blockIdx.y = group.  BOXES: write boxes (mn.w = entry bits) and the initial ordering; REDUCE: union into `red`.
template <bool BOXES, bool REDUCE>
__global__ __launch_bounds__(256) void k_items(DevPrimSet S, const BuildGroup* __restrict__ groups, double sphere_e,
                                               float4* __restrict__ boxes, unsigned* __restrict__ order, unsigned* __restrict__ red) {
    const BuildGroup g = groups[blockIdx.y];
    if ((long long)blockIdx.x * blockDim.x >= g.n) return;   // whole workgroup idle (groups differ in size)
    DBox acc; db_clear(acc);
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < g.n; k += (long long)gridDim.x * blockDim.x) {
        DBox b;
        const unsigned entry = item_box(S, g, (int)k, sphere_e, b);
        if (BOXES) {
            const size_t item = (size_t)g.item_base + (size_t)k;
            boxes[2 * item] = make_float4(b.mn[0], b.mn[1], b.mn[2], i2f((int)entry));
            boxes[2 * item + 1] = make_float4(b.mx[0], b.mx[1], b.mx[2], 0.0f);
            order[item] = (unsigned)item;
        }
        if (REDUCE) { db_add(acc, b.mn[0], b.mn[1], b.mn[2]); db_add(acc, b.mx[0], b.mx[1], b.mx[2]); }
    }
    if (REDUCE) {
        for (int i = 0; i < 3; ++i) {
            float lo = wave_min(acc.mn[i]), hi = wave_max(acc.mx[i]);
            if ((threadIdx.x & 63) == 0) { atomicMin(&red[i], enc_ordered(lo)); atomicMax(&red[3 + i], enc_ordered(hi)); }
        }
    }
}

// Node (of level `level`) that owns position p of a group of n items: ranges are [j*n >> level, (j+1)*n >> level).
DEVI unsigned node_of(long long p, int n, int level) {
    return (unsigned)((((unsigned long long)(p + 1) << level) + (unsigned long long)n - 1ull) / (unsigned long long)n - 1ull);
}

__global__ __launch_bounds__(256) void k_fill_bounds(unsigned* __restrict__ segb, long long n_nodes) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes * 6; i += (long long)gridDim.x * blockDim.x)
        segb[i] = (i % 6 < 3) ? 0xffffffffu : 0u;
}

// Bounds of the item centres of every node of `level` (groups that are still being split).
// Small nodes (at most 32 items on average): one lane per node loops over the node's items and stores the result.
// Large nodes: a wavefront walks a contiguous stretch of positions and keeps per-lane bounds for as long as it stays
// inside one node, so a node costs six atomics per wavefront that touches it; a step that straddles nodes (at most
// three of them) is reduced node by node.
__global__ __launch_bounds__(256) void k_node_bounds(const BuildGroup* __restrict__ groups, const unsigned* __restrict__ node_base, int level,
                                                     const float4* __restrict__ boxes, const unsigned* __restrict__ order, unsigned* __restrict__ segb) {
    const BuildGroup g = groups[blockIdx.y];
    if (g.depth <= level) return;
    const unsigned base = node_base[blockIdx.y];
    if ((g.n >> level) <= 32) {
        const long long width = 1ll << level;
        for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < width; j += (long long)gridDim.x * blockDim.x) {
            const long long a = (j * (long long)g.n) >> level, b = ((j + 1) * (long long)g.n) >> level;
            float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
            for (long long p = a; p < b; ++p) {
                const size_t item = order[(size_t)g.item_base + (size_t)p];
                const float4 mn = boxes[2 * item], mx = boxes[2 * item + 1];
                const float c[3] = { 0.5f * (mn.x + mx.x), 0.5f * (mn.y + mx.y), 0.5f * (mn.z + mx.z) };
                for (int i = 0; i < 3; ++i) if (c[i] == c[i]) { lo[i] = (c[i] < lo[i]) ? c[i] : lo[i]; hi[i] = (hi[i] < c[i]) ? c[i] : hi[i]; }
            }
            unsigned* sb = segb + 6 * ((size_t)base + (size_t)j);
            for (int i = 0; i < 3; ++i) { sb[i] = enc_ordered(lo[i]); sb[3 + i] = enc_ordered(hi[i]); }
        }
        return;
    }
    const long long waves = (long long)gridDim.x * (blockDim.x / 64);
    long long per_wave = (g.n + waves - 1) / waves;
    per_wave = (per_wave + 63) / 64 * 64;
    const long long wave = (long long)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const long long begin = wave * per_wave, end = (begin + per_wave < g.n) ? begin + per_wave : g.n;
    const int lane = threadIdx.x & 63;
    const unsigned NONE = 0xffffffffu;
    unsigned cur = NONE;
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    auto flush = [&]() {
        if (cur == NONE) return;
        for (int i = 0; i < 3; ++i) {
            float l = wave_min(lo[i]), h = wave_max(hi[i]);
            if (lane == 0) { atomicMin(&segb[6 * (size_t)cur + i], enc_ordered(l)); atomicMax(&segb[6 * (size_t)cur + 3 + i], enc_ordered(h)); }
            lo[i] = INFINITY; hi[i] = -INFINITY;
        }
        cur = NONE;
    };
    for (long long q = begin; q < end; q += 64) {
        const long long p = q + lane;
        const bool active = p < end;
        unsigned node = NONE;
        float c[3] = { 0.0f, 0.0f, 0.0f };
        if (active) {
            const size_t item = order[(size_t)g.item_base + (size_t)p];
            const float4 mn = boxes[2 * item], mx = boxes[2 * item + 1];
            c[0] = 0.5f * (mn.x + mx.x); c[1] = 0.5f * (mn.y + mx.y); c[2] = 0.5f * (mn.z + mx.z);
            node = base + node_of(p, g.n, level);
        }
        // the nodes of a step, one after the other (positions are contiguous, so normally there is exactly one)
        unsigned long long todo = __ballot(active);
        while (todo) {
            const unsigned first = __shfl(node, __ffsll((long long)todo) - 1);
            const bool mine = active && node == first;
            if (first != cur) { flush(); cur = first; }
            if (mine)
                for (int i = 0; i < 3; ++i) if (c[i] == c[i]) { lo[i] = (c[i] < lo[i]) ? c[i] : lo[i]; hi[i] = (hi[i] < c[i]) ? c[i] : hi[i]; }
            todo &= ~__ballot(mine);
        }
    }
    flush();
}

DEVI float dec_ordered_dev(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

// Sort key of every position: node << 32 | centre along the node's widest axis.  Items of a group that is already
// split down to its runs keep their run (key = run, equal low bits; the sort is stable).
__global__ __launch_bounds__(256) void k_node_keys(const BuildGroup* __restrict__ groups, const unsigned* __restrict__ node_base, int level,
                                                   const float4* __restrict__ boxes, const unsigned* __restrict__ order, const unsigned* __restrict__ segb,
                                                   unsigned long long* __restrict__ keys) {
    const BuildGroup g = groups[blockIdx.y];
    const unsigned base = node_base[blockIdx.y];
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < g.n; p += (long long)gridDim.x * blockDim.x) {
        const size_t pos = (size_t)g.item_base + (size_t)p;
        if (g.depth <= level) { keys[pos] = (unsigned long long)(base + node_of(p, g.n, g.depth)) << 32; continue; }
        const unsigned node = base + node_of(p, g.n, level);
        const unsigned* sb = segb + 6 * (size_t)node;
        const float e0 = dec_ordered_dev(sb[3]) - dec_ordered_dev(sb[0]), e1 = dec_ordered_dev(sb[4]) - dec_ordered_dev(sb[1]),
                    e2 = dec_ordered_dev(sb[5]) - dec_ordered_dev(sb[2]);
        int axis = 0; float ext = e0;
        if (e1 > ext) { axis = 1; ext = e1; }
        if (e2 > ext) axis = 2;
        const size_t item = order[pos];
        const float4 mn = boxes[2 * item], mx = boxes[2 * item + 1];
        const float cv = axis == 0 ? 0.5f * (mn.x + mx.x) : (axis == 1 ? 0.5f * (mn.y + mx.y) : 0.5f * (mn.z + mx.z));
        keys[pos] = ((unsigned long long)node << 32) | (unsigned long long)enc_ordered(cv);
    }
}

// Run j of a group owns the entry slots 4 * (node_base + j) .. + 3: its 2..4 sorted items, then -1.
__global__ __launch_bounds__(256) void k_entries(const BuildGroup* __restrict__ groups, const float4* __restrict__ boxes,
                                                 const unsigned* __restrict__ sorted, int* __restrict__ entries) {
    const BuildGroup g = groups[blockIdx.y];
    const long long runs = 1ll << g.depth;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < runs; j += (long long)gridDim.x * blockDim.x) {
        const long long a = (j * (long long)g.n) >> g.depth, b = ((j + 1) * (long long)g.n) >> g.depth;
        int4 e = make_int4(-1, -1, -1, -1);
        int* ev = &e.x;
        for (long long p = a; p < b && p < a + RT_SUB_LEAF; ++p) ev[p - a] = f2i(boxes[2 * (size_t)sorted[(size_t)g.item_base + (size_t)p]].w);
        *(int4*)(entries + 4 * ((size_t)g.node_base + (size_t)j)) = e;
    }
}

DEVI void box_union(float4& mn, float4& mx, float4 a, float4 b) {
    mn.x = fminf(mn.x, a.x); mn.y = fminf(mn.y, a.y); mn.z = fminf(mn.z, a.z);
    mx.x = fmaxf(mx.x, b.x); mx.y = fmaxf(mx.y, b.y); mx.z = fmaxf(mx.z, b.z);
}

// Node records.  Static scenes: 4 words per node, (minL, refL) (maxL, cntL) (minR, refR) (maxR, cntR) - both children's
// boxes over the whole motion.  Scenes with moving primitives: 8 words per node, the same four words for time 0 and
// four words that hold, per child, how far the box's corners move between time 0 and time 1; the traversal adds
// time x that to the first four (while the hierarchy is being built the second four hold the boxes at time 1
// themselves, k_motion_deltas converts them at the end).  Two siblings share a block of 16 words - both time-0 records,
// then both motion records (rt_sub_motion_offset) - so that a step of a ray at time 0 reads one 128-byte line.
//
// Lowest interior level: node = parent of two runs of sorted items.
template <bool MOTION>
DEVI float4* node_words(float4* nodes, int node_base, long long h) {
    return nodes + (MOTION ? 8 * (size_t)node_base + rt_sub_motion_offset((int)h) : 4 * ((size_t)node_base + (size_t)h));
}

template <bool MOTION>
__global__ __launch_bounds__(256) void k_leaf_parents(DevPrimSet S, const BuildGroup* __restrict__ groups, double sphere_e, const float4* __restrict__ boxes,
                                                      const unsigned* __restrict__ sorted, float4* __restrict__ nodes) {
    constexpr int ST = MOTION ? 8 : 4;
    const BuildGroup g = groups[blockIdx.y];
    const long long half = 1ll << (g.depth - 1);
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < half; t += (long long)gridDim.x * blockDim.x) {
        float4 out[ST];
        for (int side = 0; side < 2; ++side) {
            const long long j = 2 * t + side;
            const long long a = (j * (long long)g.n) >> g.depth, b = ((j + 1) * (long long)g.n) >> g.depth;
            float4 mn = make_float4(INFINITY, INFINITY, INFINITY, 0.0f), mx = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f);
            float4 mn1 = mn, mx1 = mx;
            for (long long p = a; p < b; ++p) {
                const size_t item = sorted[(size_t)g.item_base + (size_t)p];
                if (MOTION) {
                    DBox sw, t0, t1;
                    prim_boxes(S, (unsigned)f2i(boxes[2 * item].w), sphere_e, sw, &t0, &t1);
                    box_union(mn, mx, make_float4(t0.mn[0], t0.mn[1], t0.mn[2], 0.0f), make_float4(t0.mx[0], t0.mx[1], t0.mx[2], 0.0f));
                    box_union(mn1, mx1, make_float4(t1.mn[0], t1.mn[1], t1.mn[2], 0.0f), make_float4(t1.mx[0], t1.mx[1], t1.mx[2], 0.0f));
                } else {
                    box_union(mn, mx, boxes[2 * item], boxes[2 * item + 1]);
                }
            }
            mn.w = i2f(~(int)(4 * ((long long)g.node_base + j)));
            mx.w = i2f((int)(b - a));
            out[2 * side] = mn; out[2 * side + 1] = mx;
            if (MOTION) { out[4 + 2 * side] = mn1; out[4 + 2 * side + 1] = mx1; }
        }
        float4* nd = node_words<MOTION>(nodes, g.node_base, half - 1 + t);
        for (int k = 0; k < 4; ++k) nd[k] = out[k];
        if (MOTION) for (int k = 0; k < 4; ++k) nd[8 + k] = out[4 + k];
    }
}

// Level `depth - height` of every group that is at least `height` levels deep (height >= 2): union of the children.
template <bool MOTION>
__global__ __launch_bounds__(256) void k_upper_level(const BuildGroup* __restrict__ groups, int height, float4* __restrict__ nodes) {
    constexpr int ST = MOTION ? 8 : 4;
    const BuildGroup g = groups[blockIdx.y];
    if (g.depth < height) return;
    const long long width = 1ll << (g.depth - height);
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < width; t += (long long)gridDim.x * blockDim.x) {
        const long long i = width - 1 + t;
        float4 out[ST];
        for (int side = 0; side < 2; ++side) {
            const long long ch = 2 * i + 1 + side;
            const float4* cd = node_words<MOTION>(nodes, g.node_base, ch);
            float4 mn = cd[0], mx = cd[1];
            box_union(mn, mx, cd[2], cd[3]);
            mn.w = i2f(g.node_base + (int)ch); mx.w = i2f(0);
            out[2 * side] = mn; out[2 * side + 1] = mx;
            if (MOTION) {
                float4 mn1 = cd[8], mx1 = cd[9];
                box_union(mn1, mx1, cd[10], cd[11]);
                mn1.w = 0.0f; mx1.w = 0.0f;
                out[4 + 2 * side] = mn1; out[4 + 2 * side + 1] = mx1;
            }
        }
        float4* nd = node_words<MOTION>(nodes, g.node_base, i);
        for (int k = 0; k < 4; ++k) nd[k] = out[k];
        if (MOTION) for (int k = 0; k < 4; ++k) nd[8 + k] = out[4 + k];
    }
}

// Motion words of every 8-word node: box at time 1 -> (box at time 1) - (box at time 0).  Corners that are not finite
// at either time get a NaN difference, which makes the traversal visit the child.
__global__ __launch_bounds__(256) void k_motion_deltas(const BuildGroup* __restrict__ groups, float4* __restrict__ nodes) {
    const BuildGroup g = groups[blockIdx.y];
    const long long n = (1ll << g.depth) - 1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float4* nd = node_words<true>(nodes, g.node_base, i);
        for (int k = 0; k < 4; ++k) {
            const float4 a = nd[k], b = nd[8 + k];
            nd[8 + k] = make_float4(b.x - a.x, b.y - a.y, b.z - a.z, 0.0f);
        }
    }
}

// What the walk reads (sub_trace in rt_kernels.hip), made from the finished binary trees.  blockIdx.y = group; one thread
// per record; a group's records start at its wide_base.
//  * Records 0 .. n_rec - 1, n_rec = (4^ceil(depth / 2) - 1) / 3: record c stands for the binary node of level 2j at
//    place i inside that level, c = (4^j - 1) / 3 + i, and holds, plane by plane, the boxes of that node's four
//    grandchildren - which the binary records of its two children hold, two each - or, at the bottom of a tree of odd
//    depth, of its two children (its own binary record), with two boxes nothing can enter in the unused places.
//  * Records n_rec + r: run r of the group, holding the boxes of the run's own primitives (true bounds, enlarged like
//    every box of the hierarchy), their entries and their places in the reference's testing order.
//   words 0-2: min x / y / z of children 0..3   words 3-5: max x / y / z
//   word 6: the four entries of a run (class << 30 | index in the set, -1 = unused)   word 7: their testing-order keys
//   MOTION: words 8-13 the planes' change until time 1, words 14-15 unused
template <bool MOTION>
__global__ __launch_bounds__(256) void k_wide_records(DevPrimSet S, const BuildGroup* __restrict__ groups, double sphere_e, const float4* __restrict__ nodes,
                                                      const int* __restrict__ entries, const int* __restrict__ order, float4* __restrict__ wide) {
    constexpr int W = MOTION ? 16 : 8;
    const BuildGroup g = groups[blockIdx.y];
    const int levels = (g.depth + 1) >> 1;
    const long long n_rec = ((1ll << (2 * levels)) - 1) / 3, n_all = n_rec + (1ll << g.depth);
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < n_all; c += (long long)gridDim.x * blockDim.x) {
        float4 mn[4], mx[4], dmn[4], dmx[4];
        float4 w6 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), w7 = w6;
        for (int k = 0; k < 4; ++k) {
            mn[k] = make_float4(FLT_MAX, FLT_MAX, FLT_MAX, 0.0f); mx[k] = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, 0.0f);
            dmn[k] = dmx[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        if (c >= n_rec) {
            const size_t slot0 = 4 * ((size_t)g.node_base + (size_t)(c - n_rec));
            int ev[4], ov[4];
            for (int k = 0; k < 4; ++k) {
                ev[k] = entries[slot0 + k]; ov[k] = order ? order[slot0 + k] : ev[k];
                if (ev[k] == -1) continue;
                DBox sw, t0, t1;
                prim_boxes(S, (unsigned)ev[k], sphere_e, sw, MOTION ? &t0 : nullptr, MOTION ? &t1 : nullptr);
                if (MOTION) {
                    mn[k] = make_float4(t0.mn[0], t0.mn[1], t0.mn[2], 0.0f); mx[k] = make_float4(t0.mx[0], t0.mx[1], t0.mx[2], 0.0f);
                    dmn[k] = make_float4(t1.mn[0] - t0.mn[0], t1.mn[1] - t0.mn[1], t1.mn[2] - t0.mn[2], 0.0f);
                    dmx[k] = make_float4(t1.mx[0] - t0.mx[0], t1.mx[1] - t0.mx[1], t1.mx[2] - t0.mx[2], 0.0f);
                } else {
                    mn[k] = make_float4(sw.mn[0], sw.mn[1], sw.mn[2], 0.0f); mx[k] = make_float4(sw.mx[0], sw.mx[1], sw.mx[2], 0.0f);
                }
            }
            w6 = make_float4(i2f(ev[0]), i2f(ev[1]), i2f(ev[2]), i2f(ev[3]));
            w7 = make_float4(i2f(ov[0]), i2f(ov[1]), i2f(ov[2]), i2f(ov[3]));
        } else {
            int j = 0;
            while (((1ll << (2 * (j + 1))) - 1) / 3 <= c) ++j;
            const long long i = c - ((1ll << (2 * j)) - 1) / 3;
            const long long h = (1ll << (2 * j)) - 1 + i;                 // the binary node
            if (2 * j + 2 <= g.depth) {
                for (int side = 0; side < 2; ++side) {
                    const float4* cd = node_words<MOTION>(const_cast<float4*>(nodes), g.node_base, 2 * h + 1 + side);
                    mn[2 * side] = cd[0]; mx[2 * side] = cd[1]; mn[2 * side + 1] = cd[2]; mx[2 * side + 1] = cd[3];
                    if (MOTION) { dmn[2 * side] = cd[8]; dmx[2 * side] = cd[9]; dmn[2 * side + 1] = cd[10]; dmx[2 * side + 1] = cd[11]; }
                }
            } else {
                const float4* cd = node_words<MOTION>(const_cast<float4*>(nodes), g.node_base, h);
                mn[0] = cd[0]; mx[0] = cd[1]; mn[1] = cd[2]; mx[1] = cd[3];
                if (MOTION) { dmn[0] = cd[8]; dmx[0] = cd[9]; dmn[1] = cd[10]; dmx[1] = cd[11]; }
            }
        }
        float4* out = wide + (size_t)W * ((size_t)g.wide_base + (size_t)c);
        out[0] = make_float4(mn[0].x, mn[1].x, mn[2].x, mn[3].x); out[1] = make_float4(mn[0].y, mn[1].y, mn[2].y, mn[3].y);
        out[2] = make_float4(mn[0].z, mn[1].z, mn[2].z, mn[3].z); out[3] = make_float4(mx[0].x, mx[1].x, mx[2].x, mx[3].x);
        out[4] = make_float4(mx[0].y, mx[1].y, mx[2].y, mx[3].y); out[5] = make_float4(mx[0].z, mx[1].z, mx[2].z, mx[3].z);
        out[6] = w6; out[7] = w7;
        if (MOTION) {
            out[8] = make_float4(dmn[0].x, dmn[1].x, dmn[2].x, dmn[3].x); out[9] = make_float4(dmn[0].y, dmn[1].y, dmn[2].y, dmn[3].y);
            out[10] = make_float4(dmn[0].z, dmn[1].z, dmn[2].z, dmn[3].z); out[11] = make_float4(dmx[0].x, dmx[1].x, dmx[2].x, dmx[3].x);
            out[12] = make_float4(dmx[0].y, dmx[1].y, dmx[2].y, dmx[3].y); out[13] = make_float4(dmx[0].z, dmx[1].z, dmx[2].z, dmx[3].z);
            out[14] = out[15] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    }
}

// blockIdx.y = leaf: copy the leaf's primitive records from the file-order arrays.  A leaf's planes take their
// offset vertex from the file-order first vertex (kept in the spare lanes of the file record's third word).
__global__ __launch_bounds__(256) void k_gather_leaves(DevPrimSet F, const int* __restrict__ ids, const LeafGather* __restrict__ leaves,
                                                       float4* __restrict__ lp, float4* __restrict__ lc, float4* __restrict__ ls) {
    const LeafGather L = leaves[blockIdx.y];
    const int n = L.n_planes + L.n_cubes + L.n_spheres;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const int id = ids[L.first_id + k];
        if (k < L.n_planes) {
            const float4 w0 = F.planes[3 * (size_t)id], w1 = F.planes[3 * (size_t)id + 1], w2 = F.planes[3 * (size_t)id + 2];
            float4* o = lp + 3 * ((size_t)L.lp0 + (size_t)k);
            o[0] = w0; o[1] = make_float4(w2.y, w2.z, w2.w, w1.w); o[2] = make_float4(w2.x, 0.0f, 0.0f, 0.0f);
        } else if (k < L.n_planes + L.n_cubes) {
            const float4* q = F.cubes + 6 * (size_t)id;
            float4* o = lc + 6 * ((size_t)L.lc0 + (size_t)(k - L.n_planes));
            for (int w = 0; w < 6; ++w) o[w] = q[w];
        } else {
            float4* o = ls + 2 * ((size_t)L.ls0 + (size_t)(k - L.n_planes - L.n_cubes));
            o[0] = F.spheres[2 * (size_t)id]; o[1] = F.spheres[2 * (size_t)id + 1];
        }
    }
}


// ---- records in hierarchy order ------------------------------------------------------------------------
// A run's primitives are tested together, so their records should sit together.  Slots are numbered in hierarchy order
// (run after run), and the k-th plane / cube / sphere met in that order inside a group moves to the k-th place of the
// group's range of its class: the records of a run become neighbours in memory.  What the old index stood for - the
// reference's testing order, which decides exact ties - is kept per slot in `order`.
struct ClassCount { unsigned p, c, v, pad; };                // planes, cubes and used slots before a slot
struct ClassPlus {
    __host__ __device__ ClassCount operator()(const ClassCount& a, const ClassCount& b) const {
        ClassCount r; r.p = a.p + b.p; r.c = a.c + b.c; r.v = a.v + b.v; r.pad = 0u; return r;
    }
};

__global__ __launch_bounds__(256) void k_slot_flags(const int* __restrict__ entries, long long slot0, long long n_slots, ClassCount* __restrict__ flags) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += (long long)gridDim.x * blockDim.x) {
        const int e = entries[slot0 + i];
        ClassCount f; f.pad = 0u;
        const unsigned cls = (unsigned)e >> 30;
        f.p = cls == 0u ? 1u : 0u; f.c = cls == 1u ? 1u : 0u; f.v = e != -1 ? 1u : 0u;          // -1 (class bits 3) = unused slot
        flags[i] = f;
    }
}

// blockIdx.y = group.  entries: old index in, new index out; order: the old entry (or -1) of every slot.
__global__ __launch_bounds__(256) void k_reorder(DevPrimSet S, const BuildGroup* __restrict__ groups, long long slot0, const ClassCount* __restrict__ scan,
                                                 int* __restrict__ entries, int* __restrict__ order,
                                                 float4* __restrict__ op, float4* __restrict__ oc, float4* __restrict__ os) {
    const BuildGroup g = groups[blockIdx.y];
    const long long first = 4ll * g.node_base, n = 4ll << g.depth;
    const ClassCount base = scan[first - slot0];
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
        const long long slot = first + k;
        const int e = entries[slot];
        order[slot] = e;
        if (e == -1) continue;
        const ClassCount cc = scan[slot - slot0];
        const unsigned cls = (unsigned)e >> 30; const size_t old = (size_t)((unsigned)e & 0x3fffffffu);
        const unsigned rp = cc.p - base.p, rc = cc.c - base.c, rs = (cc.v - base.v) - rp - rc;
        if (cls == 0u) {
            const size_t to = (size_t)g.p0 + rp;
            for (int w = 0; w < 3; ++w) op[3 * to + w] = S.planes[3 * old + w];
            entries[slot] = (int)to;
        } else if (cls == 1u) {
            const size_t to = (size_t)g.c0 + rc;
            for (int w = 0; w < 6; ++w) oc[6 * to + w] = S.cubes[6 * old + w];
            entries[slot] = (int)((1u << 30) | (unsigned)to);
        } else {
            const size_t to = (size_t)g.s0 + rs;
            os[2 * to] = S.spheres[2 * old]; os[2 * to + 1] = S.spheres[2 * old + 1];
            entries[slot] = (int)((2u << 30) | (unsigned)to);
        }
    }
}

struct Scratch {                       // device allocations of one build, freed on every path out
    std::vector<void*> ptrs;
    ~Scratch() { for (void* p : ptrs) (void)hipFree(p); }
    template <class T> hipError_t get(T** out, size_t count) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess) { ptrs.push_back(p); *out = (T*)p; }
        return e;
    }
};

#define BUILD_TRY(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return e__; } while (0)

inline int grid_for(long long n) { long long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

const unsigned RED_INIT[6] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u };

} // namespace

hipError_t rt_build_scene_bounds(hipStream_t s, const DevPrimSet& set, float out6[6]) {
    const long long n = (long long)set.n_planes + set.n_cubes + set.n_spheres;
    for (int i = 0; i < 3; ++i) { out6[i] = INFINITY; out6[3 + i] = -INFINITY; }
    if (n == 0) return hipSuccess;
    Scratch sc; BuildGroup* d_group; unsigned* d_red;
    BUILD_TRY(sc.get(&d_group, 1)); BUILD_TRY(sc.get(&d_red, 6));
    BuildGroup g{}; g.pn = set.n_planes; g.cn = set.n_cubes; g.sn = set.n_spheres; g.n = (int)n; g.depth = 1;
    BUILD_TRY(hipMemcpyAsync(d_group, &g, sizeof g, hipMemcpyHostToDevice, s));
    BUILD_TRY(hipMemcpyAsync(d_red, RED_INIT, sizeof RED_INIT, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL((k_items<false, true>), dim3(grid_for(n), 1), dim3(256), 0, s, set, d_group, 0.0, (float4*)nullptr, (unsigned*)nullptr, d_red);
    BUILD_TRY(hipGetLastError());
    unsigned red[6];
    BUILD_TRY(hipMemcpyAsync(red, d_red, sizeof red, hipMemcpyDeviceToHost, s));
    BUILD_TRY(hipStreamSynchronize(s));
    for (int i = 0; i < 6; ++i) out6[i] = dec_ordered(red[i]);
    return hipSuccess;
}

hipError_t rt_build_hierarchies(hipStream_t s, const DevPrimSet& set, const BuildGroup* groups, int n_groups, long long n_items,
                                double sphere_e, bool motion, float4* nodes, int* entries, float final6[6]) {
    if (final6) for (int i = 0; i < 3; ++i) { final6[i] = INFINITY; final6[3 + i] = -INFINITY; }
    if (n_groups <= 0 || n_items <= 0) return hipSuccess;
    if (n_groups > 65535 || n_items > 0x7fffffffll) return hipErrorInvalidValue;
    int max_n = 0, max_depth = 1;
    for (int g = 0; g < n_groups; ++g) { max_n = groups[g].n > max_n ? groups[g].n : max_n; max_depth = groups[g].depth > max_depth ? groups[g].depth : max_depth; }
    // first node of every group at every level (a group that is already down to its runs keeps 2^depth of them)
    std::vector<unsigned> node_base((size_t)(max_depth + 1) * (size_t)n_groups);
    std::vector<long long> level_nodes((size_t)max_depth + 1, 0);
    for (int l = 0; l <= max_depth; ++l) {
        long long acc = 0;
        for (int g = 0; g < n_groups; ++g) { node_base[(size_t)l * n_groups + g] = (unsigned)acc; acc += 1ll << (l < groups[g].depth ? l : groups[g].depth); }
        level_nodes[(size_t)l] = acc;
        if (acc > 0xffffffffll) return hipErrorInvalidValue;
    }
    Scratch sc;
    BuildGroup* d_groups; unsigned *d_red, *d_node_base, *d_segb; float4* d_boxes; unsigned long long *d_keys, *d_keys2; unsigned *d_ord, *d_ord2;
    BUILD_TRY(sc.get(&d_groups, (size_t)n_groups)); BUILD_TRY(sc.get(&d_red, 6));
    BUILD_TRY(sc.get(&d_node_base, node_base.size()));
    BUILD_TRY(sc.get(&d_segb, 6 * (size_t)level_nodes[(size_t)max_depth - 1]));
    BUILD_TRY(sc.get(&d_boxes, 2 * (size_t)n_items));
    BUILD_TRY(sc.get(&d_keys, (size_t)n_items)); BUILD_TRY(sc.get(&d_keys2, (size_t)n_items));
    BUILD_TRY(sc.get(&d_ord, (size_t)n_items)); BUILD_TRY(sc.get(&d_ord2, (size_t)n_items));
    BUILD_TRY(hipMemcpyAsync(d_groups, groups, sizeof(BuildGroup) * (size_t)n_groups, hipMemcpyHostToDevice, s));
    BUILD_TRY(hipMemcpyAsync(d_node_base, node_base.data(), sizeof(unsigned) * node_base.size(), hipMemcpyHostToDevice, s));
    BUILD_TRY(hipMemcpyAsync(d_red, RED_INIT, sizeof RED_INIT, hipMemcpyHostToDevice, s));

    const int gx = grid_for(max_n);
    if (final6) hipLaunchKernelGGL((k_items<true, true>), dim3(gx, n_groups), dim3(256), 0, s, set, d_groups, sphere_e, d_boxes, d_ord, d_red);
    else hipLaunchKernelGGL((k_items<true, false>), dim3(gx, n_groups), dim3(256), 0, s, set, d_groups, sphere_e, d_boxes, d_ord, d_red);
    BUILD_TRY(hipGetLastError());

    size_t temp_bytes = 0;
    BUILD_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, d_keys, d_keys2, d_ord, d_ord2, (size_t)n_items, 0u, 64u, s));
    char* d_temp; BUILD_TRY(sc.get(&d_temp, temp_bytes));
    for (int level = 0; level < max_depth; ++level) {
        const unsigned* nb = d_node_base + (size_t)level * n_groups;
        hipLaunchKernelGGL(k_fill_bounds, dim3(grid_for(level_nodes[(size_t)level] * 6)), dim3(256), 0, s, d_segb, level_nodes[(size_t)level]);
        hipLaunchKernelGGL(k_node_bounds, dim3(gx, n_groups), dim3(256), 0, s, d_groups, nb, level, d_boxes, d_ord, d_segb);
        hipLaunchKernelGGL(k_node_keys, dim3(gx, n_groups), dim3(256), 0, s, d_groups, nb, level, d_boxes, d_ord, d_segb, d_keys);
        BUILD_TRY(hipGetLastError());
        int node_bits = 1; while ((1ll << node_bits) < level_nodes[(size_t)level]) ++node_bits;
        size_t tb = temp_bytes;
        BUILD_TRY(rocprim::radix_sort_pairs(d_temp, tb, d_keys, d_keys2, d_ord, d_ord2, (size_t)n_items, 0u, (unsigned)(32 + node_bits), s));
        unsigned* t = d_ord; d_ord = d_ord2; d_ord2 = t;
    }

    hipLaunchKernelGGL(k_entries, dim3(grid_for(1ll << max_depth), n_groups), dim3(256), 0, s, d_groups, d_boxes, d_ord, entries);
    if (motion) {
        hipLaunchKernelGGL(k_leaf_parents<true>, dim3(grid_for(((long long)max_n + 3) / 4), n_groups), dim3(256), 0, s, set, d_groups, sphere_e, d_boxes, d_ord, nodes);
        for (int h = 2; h <= max_depth; ++h)
            hipLaunchKernelGGL(k_upper_level<true>, dim3(grid_for(1ll << (max_depth - h)), n_groups), dim3(256), 0, s, d_groups, h, nodes);
        hipLaunchKernelGGL(k_motion_deltas, dim3(grid_for(1ll << max_depth), n_groups), dim3(256), 0, s, d_groups, nodes);
    } else {
        hipLaunchKernelGGL(k_leaf_parents<false>, dim3(grid_for(((long long)max_n + 3) / 4), n_groups), dim3(256), 0, s, set, d_groups, sphere_e, d_boxes, d_ord, nodes);
        for (int h = 2; h <= max_depth; ++h)
            hipLaunchKernelGGL(k_upper_level<false>, dim3(grid_for(1ll << (max_depth - h)), n_groups), dim3(256), 0, s, d_groups, h, nodes);
    }
    BUILD_TRY(hipGetLastError());
    unsigned red[6];
    if (final6) BUILD_TRY(hipMemcpyAsync(red, d_red, sizeof red, hipMemcpyDeviceToHost, s));
    BUILD_TRY(hipStreamSynchronize(s));
    if (final6) for (int i = 0; i < 6; ++i) final6[i] = dec_ordered(red[i]);
    return hipSuccess;
}

hipError_t rt_build_wide_records(hipStream_t s, const DevPrimSet& set, const BuildGroup* groups, int n_groups, bool motion, double sphere_e,
                                 const float4* nodes, const int* entries, const int* order, float4* wide) {
    if (n_groups <= 0) return hipSuccess;
    if (n_groups > 65535) return hipErrorInvalidValue;
    int max_depth = 1;
    for (int g = 0; g < n_groups; ++g) max_depth = groups[g].depth > max_depth ? groups[g].depth : max_depth;
    Scratch sc; BuildGroup* d_groups;
    BUILD_TRY(sc.get(&d_groups, (size_t)n_groups));
    BUILD_TRY(hipMemcpyAsync(d_groups, groups, sizeof(BuildGroup) * (size_t)n_groups, hipMemcpyHostToDevice, s));
    const long long most = rt_wide_records(max_depth);
    if (motion) hipLaunchKernelGGL(k_wide_records<true>, dim3(grid_for(most), n_groups), dim3(256), 0, s, set, d_groups, sphere_e, nodes, entries, order, wide);
    else hipLaunchKernelGGL(k_wide_records<false>, dim3(grid_for(most), n_groups), dim3(256), 0, s, set, d_groups, sphere_e, nodes, entries, order, wide);
    BUILD_TRY(hipGetLastError());
    BUILD_TRY(hipStreamSynchronize(s));
    return hipSuccess;
}

hipError_t rt_build_gather_leaves(hipStream_t s, const DevPrimSet& fileset, const int* prim_ids_dev, const LeafGather* leaves, int n_leaves,
                                  float4* leaf_planes, float4* leaf_cubes, float4* leaf_spheres) {
    if (n_leaves <= 0) return hipSuccess;
    Scratch sc; LeafGather* d_leaves;
    BUILD_TRY(sc.get(&d_leaves, (size_t)n_leaves));
    BUILD_TRY(hipMemcpyAsync(d_leaves, leaves, sizeof(LeafGather) * (size_t)n_leaves, hipMemcpyHostToDevice, s));
    long long max_n = 0;
    for (int i = 0; i < n_leaves; ++i) { long long n = (long long)leaves[i].n_planes + leaves[i].n_cubes + leaves[i].n_spheres; max_n = n > max_n ? n : max_n; }
    hipLaunchKernelGGL(k_gather_leaves, dim3(grid_for(max_n), n_leaves), dim3(256), 0, s, fileset, prim_ids_dev, d_leaves, leaf_planes, leaf_cubes, leaf_spheres);
    BUILD_TRY(hipGetLastError());
    return hipStreamSynchronize(s);
}

hipError_t rt_build_reorder(hipStream_t s, const DevPrimSet& set, const BuildGroup* groups, int n_groups, int* entries, int* order,
                            float4* out_planes, float4* out_cubes, float4* out_spheres) {
    if (n_groups <= 0) return hipSuccess;
    if (n_groups > 65535) return hipErrorInvalidValue;
    long long slot0 = 0x7fffffffffffffffll, slot1 = 0; int max_depth = 1;
    for (int g = 0; g < n_groups; ++g) {
        const long long a = 4ll * groups[g].node_base, b = a + (4ll << groups[g].depth);
        slot0 = a < slot0 ? a : slot0; slot1 = b > slot1 ? b : slot1; max_depth = groups[g].depth > max_depth ? groups[g].depth : max_depth;
    }
    const long long n_slots = slot1 - slot0;
    Scratch sc; BuildGroup* d_groups; ClassCount *d_flags, *d_scan;
    BUILD_TRY(sc.get(&d_groups, (size_t)n_groups)); BUILD_TRY(sc.get(&d_flags, (size_t)n_slots)); BUILD_TRY(sc.get(&d_scan, (size_t)n_slots));
    BUILD_TRY(hipMemcpyAsync(d_groups, groups, sizeof(BuildGroup) * (size_t)n_groups, hipMemcpyHostToDevice, s));
    // slots between groups (none today) would hold stale values: count what the entries say, group by group they are all written
    hipLaunchKernelGGL(k_slot_flags, dim3(grid_for(n_slots)), dim3(256), 0, s, entries, slot0, n_slots, d_flags);
    BUILD_TRY(hipGetLastError());
    size_t temp_bytes = 0;
    ClassCount zero{}; zero.p = zero.c = zero.v = zero.pad = 0u;
    BUILD_TRY(rocprim::exclusive_scan(nullptr, temp_bytes, d_flags, d_scan, zero, (size_t)n_slots, ClassPlus(), s));
    char* d_temp; BUILD_TRY(sc.get(&d_temp, temp_bytes));
    BUILD_TRY(rocprim::exclusive_scan(d_temp, temp_bytes, d_flags, d_scan, zero, (size_t)n_slots, ClassPlus(), s));
    hipLaunchKernelGGL(k_reorder, dim3(grid_for(4ll << max_depth), n_groups), dim3(256), 0, s, set, d_groups, slot0, d_scan, entries, order,
                       out_planes, out_cubes, out_spheres);
    BUILD_TRY(hipGetLastError());
    return hipStreamSynchronize(s);
}

hipError_t rt_sort_pairs_temp_bytes(size_t n, size_t* bytes) {
    *bytes = 0;
    return rocprim::radix_sort_pairs(nullptr, *bytes, (const unsigned*)nullptr, (unsigned*)nullptr, (const unsigned*)nullptr, (unsigned*)nullptr, n, 0u, 32u, (hipStream_t)nullptr);
}

hipError_t rt_sort_pairs(hipStream_t s, void* temp, size_t temp_bytes, const unsigned* keys_in, unsigned* keys_out, const unsigned* vals_in, unsigned* vals_out,
                         size_t n, int bits) {
    if (n == 0) return hipSuccess;
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, (unsigned)bits, s);
}
