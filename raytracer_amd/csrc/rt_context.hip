// Context behind the C ABI of include/rt_api.h: device copies of the scene, the re-laid-out BVH, the work
// buffers, and the per-frame launch sequence.  All device work goes through the kernels in rt_kernels.hip.
#include "rt_api.h"
#include "rt_types.h"
#include "camera_basis.h"
#include "rt_build.h"
#include "rt_parallel.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <limits>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

std::string g_create_error;

template <class T> struct DevBuf {
    T* p = nullptr; size_t n = 0;
    void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
    hipError_t resize(size_t count) {
        if (count <= n && p) return hipSuccess;
        release();
        size_t c = std::max<size_t>(count, 1);
        hipError_t e = hipMalloc((void**)&p, c * sizeof(T));
        if (e == hipSuccess) n = c; else p = nullptr;
        return e;
    }
    hipError_t upload(const std::vector<T>& v, hipStream_t s) {
        hipError_t e = resize(v.size());
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
    }
};

inline float4 f4(float x, float y, float z, float w) { return make_float4(x, y, z, w); }
inline float as_f(int i) { float f; std::memcpy(&f, &i, 4); return f; }
inline float smaxh(float a, float b) { return (a < b) ? b : a; }
inline float sminh(float a, float b) { return (b < a) ? b : a; }

// plane.cpp:207-216
int drop_axis(const float n[3]) {
    float nx = std::fabs(n[0]), ny = std::fabs(n[1]), nz = std::fabs(n[2]);
    if (nx > ny && nx > nz) return 0;
    if (ny > nz) return 1;
    return 2;
}

// Plane::sortVerticesWinding, plane.cpp:254-301: order by atan2 of the offset from the centroid in the plane's
// dominant 2D projection.  std::sort on <= 16 elements is an insertion sort, which a stable sort reproduces.
std::vector<int> winding_order(const float* v, int n, const float normal[3]) {
    float c[3] = { 0.0f, 0.0f, 0.0f };
    for (int i = 0; i < n; ++i) { c[0] += v[3 * i]; c[1] += v[3 * i + 1]; c[2] += v[3 * i + 2]; }
    float fn = (float)(size_t)n;
    c[0] /= fn; c[1] /= fn; c[2] /= fn;
    int drop = drop_axis(normal);
    int a0 = (drop == 0) ? 1 : 0, a1 = (drop == 2) ? 1 : 2;
    std::vector<float> ang((size_t)n);
    for (int i = 0; i < n; ++i) ang[i] = std::atan2(v[3 * i + a1] - c[a1], v[3 * i + a0] - c[a0]);
    std::vector<int> idx((size_t)n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return ang[a] < ang[b]; });
    return idx;
}

float len_ref(const float v[3]) {
    return (float)std::sqrt((double)v[0] * (double)v[0] + (double)v[1] * (double)v[1] + (double)v[2] * (double)v[2]);
}

// ---- true bounds of the primitives (NOT the reference's shape boxes, which under-cover moving scaled cubes):
// everything a primitive can occupy for ray times in [0,1], slightly enlarged for the rounding of this very
// computation.  They only steer the second-level hierarchy, never a reported result.
struct Box3 {
    float mn[3], mx[3];
    void clear() { for (int i = 0; i < 3; ++i) { mn[i] = INFINITY; mx[i] = -INFINITY; } }
    void add(const float p[3]) { for (int i = 0; i < 3; ++i) { mn[i] = std::min(mn[i], p[i]); mx[i] = std::max(mx[i], p[i]); } }
    void add(const Box3& o) { for (int i = 0; i < 3; ++i) { mn[i] = std::min(mn[i], o.mn[i]); mx[i] = std::max(mx[i], o.mx[i]); } }
    void grow() {
        for (int i = 0; i < 3; ++i) {
            float e = 1e-5f * std::max(std::max(std::fabs(mn[i]), std::fabs(mx[i])), mx[i] - mn[i]) + 1e-6f;
            mn[i] -= e; mx[i] += e;
        }
    }
};

Box3 true_box(const rt_sphere& s) {
    Box3 b; b.clear();
    for (int k = 0; k < 2; ++k) {
        const float* c = k ? s.end : s.start; float r = std::fabs(s.radius);
        float lo[3] = { c[0] - r, c[1] - r, c[2] - r }, hi[3] = { c[0] + r, c[1] + r, c[2] + r };
        b.add(lo); b.add(hi);
    }
    b.grow(); return b;
}

Box3 true_box(const rt_cube& c) {
    Box3 b; b.clear();
    const float cx = std::cos(c.rotation[0]), sx = std::sin(c.rotation[0]), cy = std::cos(c.rotation[1]), sy = std::sin(c.rotation[1]);
    const float cz = std::cos(c.rotation[2]), sz = std::sin(c.rotation[2]);
    for (int m = 0; m < 8; ++m) {
        float x = (m & 1) ? 1.0f : -1.0f, y = (m & 2) ? 1.0f : -1.0f, z = (m & 4) ? 1.0f : -1.0f;
        float y1 = y * cx - z * sx, z1 = y * sx + z * cx;
        float x2 = x * cy + z1 * sy, z2 = -x * sy + z1 * cy;
        float r[3] = { (x2 * cz - y1 * sz) * c.scale, (x2 * sz + y1 * cz) * c.scale, z2 * c.scale };
        for (int k = 0; k < 2; ++k) {
            const float* o = k ? c.end : c.start;
            float p[3] = { r[0] + o[0], r[1] + o[1], r[2] + o[2] };
            b.add(p);
        }
    }
    b.grow(); return b;
}

Box3 true_box(const rt_plane& p, const float* verts) {
    Box3 b; b.clear();
    for (int k = 0; k < p.n_verts; ++k) b.add(verts + (size_t)(p.first_vert + k) * 3);
    b.grow(); return b;
}

// ---- second-level hierarchy builder on the host (rt_set_option("device_build", 0)): the layout of rt_build.hip - a
// perfect binary tree in heap order from node_base, node j of level l owning the items at positions
// [j*n >> l, (j+1)*n >> l) of an ordering made by median splits on the widest centroid axis, four entry slots per run
// from 4 * (node_base + run).
struct SubItem { unsigned entry; Box3 box; float cen[3]; };

struct SubBuilder {
    std::vector<float4>& nodes;      // 4 words per node, absolute indices
    std::vector<int>& entries;
    std::vector<SubItem>* it = nullptr;
    long long n = 0; int depth = 0, node_base = 0;

    // reserves 2^depth node slots and 4 * 2^depth entry slots at the end of the arrays; returns node_base
    int build_group(std::vector<SubItem>& items, int depth_) {
        it = &items; n = (long long)items.size(); depth = depth_;
        node_base = (int)(nodes.size() / 4);
        nodes.resize(nodes.size() + 4 * ((size_t)1 << depth), f4(0, 0, 0, 0));
        entries.resize(4 * (nodes.size() / 4), -1);
        int ref, cnt;
        rec(0, 0, ref, cnt);
        return node_base;
    }
    Box3 rec(int level, long long j, int& ref, int& cnt) {
        const size_t a = (size_t)((j * n) >> level), b = (size_t)(((j + 1) * n) >> level);
        std::vector<SubItem>& v = *it;
        Box3 out; out.clear();
        if (level == depth) {
            std::sort(v.begin() + (long)a, v.begin() + (long)b, [](const SubItem& x, const SubItem& y) { return x.entry < y.entry; });
            const size_t e0 = 4 * ((size_t)node_base + (size_t)j);
            for (size_t i = a; i < b; ++i) { entries[e0 + (i - a)] = (int)v[i].entry; out.add(v[i].box); }
            ref = ~(int)e0; cnt = (int)(b - a);
            return out;
        }
        float cmn[3] = { INFINITY, INFINITY, INFINITY }, cmx[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (size_t i = a; i < b; ++i) for (int k = 0; k < 3; ++k) { cmn[k] = std::min(cmn[k], v[i].cen[k]); cmx[k] = std::max(cmx[k], v[i].cen[k]); }
        int axis = 0;
        if (cmx[1] - cmn[1] > cmx[axis] - cmn[axis]) axis = 1;
        if (cmx[2] - cmn[2] > cmx[axis] - cmn[axis]) axis = 2;
        const size_t mid = (size_t)(((2 * j + 1) * n) >> (level + 1));
        if (mid > a && mid < b)
            std::nth_element(v.begin() + (long)a, v.begin() + (long)mid, v.begin() + (long)b,
                             [axis](const SubItem& x, const SubItem& y) { return x.cen[axis] < y.cen[axis] || (x.cen[axis] == y.cen[axis] && x.entry < y.entry); });
        const size_t me = (size_t)node_base + (((size_t)1 << level) - 1) + (size_t)j;
        int refL, cntL, refR, cntR;
        Box3 L = rec(level + 1, 2 * j, refL, cntL), R = rec(level + 1, 2 * j + 1, refR, cntR);
        nodes[me * 4 + 0] = f4(L.mn[0], L.mn[1], L.mn[2], as_f(refL));
        nodes[me * 4 + 1] = f4(L.mx[0], L.mx[1], L.mx[2], as_f(cntL));
        nodes[me * 4 + 2] = f4(R.mn[0], R.mn[1], R.mn[2], as_f(refR));
        nodes[me * 4 + 3] = f4(R.mx[0], R.mx[1], R.mx[2], as_f(cntR));
        ref = (int)me; cnt = 0; out = L; out.add(R);
        return out;
    }
};

SubItem make_item(int cls, int idx, const Box3& b) {
    SubItem it; it.entry = ((unsigned)cls << 30) | (unsigned)idx; it.box = b;
    for (int k = 0; k < 3; ++k) it.cen[k] = 0.5f * (b.mn[k] + b.mx[k]);
    return it;
}

} // namespace

struct rt_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    uint64_t workspace_limit = 0;
    std::string err;

    // host copy of what the BVH upload needs
    bool have_scene = false, have_bvh = false;
    rt_camera camera{};
    int np = 0, nc = 0, ns = 0;              // planes, cubes, spheres of the uploaded scene
    std::vector<float4> h_verts_sorted;      // device layout, indexed by first_vert + k
    std::vector<float4> h_edges;             // crossing-test constants of the edge that ends in each vertex
    std::vector<float4> h_file_planes, h_file_cubes, h_file_spheres;
    int n_lights = 0;
    bool any_reflective = false, any_transparent = false;
    // true bounds per primitive (file order) and the second-level hierarchy of the file-order set
    std::vector<Box3> tb_planes, tb_cubes, tb_spheres;
    std::vector<float4> h_sub_nodes_file; std::vector<int> h_sub_entries_file;
    int file_sub_root = -1;
    int sub_min_group = RT_SUB_MIN_GROUP;    // groups above this size get a second-level hierarchy (rt_set_option)
    // workgroups per launch (more work than that is taken in rounds): primary extend, secondary extend, shadow, shade
    bool skip_unlit = true;                  // see k_shadow: shadow rays whose result is multiplied by zero are not traced
    bool count_rays = true;                  // frames with statistics count rays (off: stage times only, at the speed of plain frames)
    // 0 = chosen per launch (see enqueue_frame)
    int grid_primary = 0, grid_secondary = 0, grid_shadow = 0, grid_shade = 0, grid_shadow_drain = 0;
    bool cull_cones = true;                  // see k_shadow: soft-shadow cones that miss every primitive are not traced
    bool fuse_simple = true;                 // frames without recursion and with hard shadows: trace, shadow and shade in one kernel
    bool shadow_queue_deep = true;           // ... also for the launch that covers the levels below the first reflection
    int shadow_queue_rcap = 0;               // (tests) entries per region of such a queue instead of the planned number
    bool shadow_queue = true;                // ... and the pairs that do need tracing go through a queue in memory to a launch of their own
    bool cull_umbra = true;                  // ... and pairs whose whole cone of sample rays is certain to be blocked
    bool cull_own = true;                    // ... where a sphere or cube is left out for the rays that leave its own surface outwards
    bool bundle_reflections = true;          // ... as one traced ray with gr_samples hit records (off: gr_samples traced rays)
    bool replicate_reflections = true;       // soft shadows: first reflections traced gr_samples times, each with its own shadow samples
    bool fast_samples = true;                // soft-shadow sample rays: hardware reciprocals instead of IEEE division
    int lds_prims = 1;                       // tiny scenes: primitive records in LDS; 1 = shadow kernel only, 2 = extend too
    int device_build = 1;                    // hierarchies and leaf-order records are made on the GPU (rt_set_option)
    int motion_boxes = 1;                    // device-built hierarchies of scenes with moving primitives keep boxes per time
    int hierarchy_order = 1;                 // records of groups with a second-level hierarchy are stored in hierarchy order
    bool reorder_active = true;              // ... as the option stood when the current scene was uploaded
    int sort_rays = 1;                       // scenes with second-level hierarchies: secondary rays are sorted by origin cell and direction before they are traced
    long long sort_rays_min_prims = 2000000; // ... when the scene has at least this many primitives (see enqueue_frame)
    DevBuf<unsigned> sort_keys[2], sort_slots[2];
    DevBuf<unsigned char> sort_temp;
    int sub_stride = 4;                      // words (float4) per second-level node: 4, or 8 with boxes per time
    bool built_on_device = false;            // what the current scene's upload used
    long long n_file_nodes = 0, n_file_entries = 0;
    int file_sub_depth = 0;                  // depth of the file-order hierarchy
    double sphere_e = 0.0;                   // rounding allowance of the float sphere test, see rt_upload_scene
    DevBuf<float4> d_sub_wide;                      // what the walk reads: 4-ary records made from the binary trees below (rt_build_wide_records)
    DevBuf<float4> d_sub_nodes_file, d_sub_nodes;   // file-order set alone / file-order set followed by the leaves'
    DevBuf<int> d_sub_entries_file, d_sub_entries, d_prim_ids;
    DevBuf<int> d_sub_order_file, d_sub_order;      // per entry slot: the primitive's place in the reference's testing order
    DevBuf<float4> d_walk_planes, d_walk_cubes, d_walk_spheres;   // the file-order set in hierarchy order

    // device scene
    DevBuf<float4> d_nodes, d_mats, d_plane_uv, d_plane_bound, d_texels, d_lights, d_verts, d_edges;
    DevBuf<int4> d_leaves, d_tex_desc;
    DevBuf<float4> d_file_planes, d_file_cubes, d_file_spheres, d_leaf_planes, d_leaf_cubes, d_leaf_spheres;
    DevScene scene{};

    // what export / adopt of the device-resident scene need beyond DevScene (rt_export_scene_layout)
    long long n_textures = 0, n_texels = 0, total_sub_nodes = 0, total_sub_entries = 0, total_wide = 0;
    int file_wide_base = -1;                             // first record of the file-order set's hierarchy in d_sub_wide, or -1
    bool adopted = false;                                // the scene's arrays were received from another context, not uploaded
    std::vector<DevBuf<unsigned char>> adopted_arrays;   // ... and live here,
    std::vector<std::pair<const void*, size_t>> shared_arrays;   // or in the context they were shared from (rt_share_scene_from), which owns them
    std::vector<std::pair<const void*, size_t>> export_arrays;   // distinct arrays of the last export, in order

    // sharding
    int rank = 0, world = 1;
    int tiles_x = 0, tiles_y = 0, max_tiles = 0;
    std::vector<int> local_tiles;
    DevBuf<int> d_local_tiles;

    // work buffers
    DevBuf<float4> q_a[2], q_b[2], n_a, n_b, n_c, n_col, accum;
    DevBuf<float2> n_uv;
    DevBuf<int> n_pix, link, ctr;
    DevBuf<float> shadow;
    DevBuf<int4> sq;                         // queues of (hit, light) pairs to trace, see DevWork::sq
    DevBuf<unsigned long long> stats;
    DevBuf<unsigned char> out_tmp;
    unsigned long long last_batches = 0;
    const unsigned long long* last_flag = nullptr;   // overflow flag of the most recent frame (device memory)
    std::vector<hipEvent_t> event_pool;
    // Captured launch sequences of whole frames (rt_render_device without statistics), replayed with one
    // hipGraphLaunch.  A frame is captured the second time it is asked for with the same configuration, output
    // pointer and stream, i.e. once its buffers exist; anything that changes the scene or the shard drops them.
    struct FrameGraph { rt_config cfg; void* out; hipStream_t stream; unsigned long long generation; bool warm; hipGraphExec_t exec; };
    std::vector<FrameGraph> graphs;
    unsigned long long generation = 0;
    int use_graphs = 1;
    // Second stream on which the shadow rays of the primary hits run while the render stream traces the deeper
    // recursion levels (fork and join through the two events; works the same under graph capture).
    hipStream_t aux_stream = nullptr, aux_stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_fork2 = nullptr, ev_join2 = nullptr;
    int shadow_streams = 2;                   // forked shadow launches of a frame with recursion: level 0 (1) and level 1 (2)
    int overlap = 1;
    int walk_stats = 0;                      // development: kernels built with RT_WALK_STATS count their steps (rt_debug_walk_stats)
    void drop_graphs() {
        for (auto& g : graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
        graphs.clear(); ++generation;
    }

    int fail(int code, const char* fmt, ...) {
        char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
        err = buf; return code;
    }
    int hip(hipError_t e, const char* what) {
        if (e == hipSuccess) return RT_OK;
        return fail(RT_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    }
    void rebuild_tiles() {
        int W = (int)camera.res_x, H = (int)camera.res_y;
        tiles_x = (W + RT_TILE - 1) / RT_TILE; tiles_y = (H + RT_TILE - 1) / RT_TILE;
        local_tiles.clear();
        std::vector<int> per_rank((size_t)world, 0);
        for (int ty = 0; ty < tiles_y; ++ty)
            for (int tx = 0; tx < tiles_x; ++tx) {
                int owner = (tx + ty) % world;
                per_rank[owner]++;
                if (owner == rank) local_tiles.push_back(ty * tiles_x + tx);
            }
        max_tiles = 0;
        for (int c : per_rank) max_tiles = std::max(max_tiles, c);
    }
};

#define RT_TRY(expr) do { int rc__ = (expr); if (rc__ != RT_OK) return rc__; } while (0)
#define RT_HIP(ctx, call) RT_TRY((ctx)->hip((call), #call))

static hipError_t lookup_scratch(int device);
static void release_scratch(rt_ctx* c);

extern "C" {

rt_ctx* rt_create(int device_id, uint64_t workspace_bytes) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { g_create_error = std::string("no HIP device: ") + hipGetErrorString(e); return nullptr; }
    if (device_id < 0 || device_id >= n) { g_create_error = "device id out of range"; return nullptr; }
    e = hipSetDevice(device_id);
    if (e != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return nullptr; }
    // once per device, with that device current: where this GPU keeps the library's scratch array
    e = lookup_scratch(device_id);
    if (e != hipSuccess) { g_create_error = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e); return nullptr; }
    rt_ctx* c = new rt_ctx();
    c->device = device_id;
    c->workspace_limit = workspace_bytes ? workspace_bytes : (4ull << 30);
    e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(e); delete c; return nullptr; }
    c->stream = c->own_stream;
    e = hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->aux_stream2, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork2, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join2, hipEventDisableTiming);
    if (e != hipSuccess) { g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(e); rt_destroy(c); return nullptr; }
    return c;
}

void rt_destroy(rt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    release_scratch(c);
    if (c->aux_stream) { (void)hipStreamSynchronize(c->aux_stream); (void)hipStreamDestroy(c->aux_stream); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->aux_stream2) { (void)hipStreamSynchronize(c->aux_stream2); (void)hipStreamDestroy(c->aux_stream2); }
    if (c->ev_fork2) (void)hipEventDestroy(c->ev_fork2);
    if (c->ev_join2) (void)hipEventDestroy(c->ev_join2);
    DevBuf<float4>* f4s[] = { &c->d_nodes, &c->d_mats, &c->d_plane_uv, &c->d_plane_bound, &c->d_texels, &c->d_lights, &c->d_verts, &c->d_edges, &c->d_file_planes,
                              &c->d_file_cubes, &c->d_file_spheres, &c->d_leaf_planes, &c->d_leaf_cubes, &c->d_leaf_spheres,
                              &c->q_a[0], &c->q_a[1], &c->q_b[0], &c->q_b[1], &c->n_a, &c->n_b, &c->n_c, &c->n_col, &c->accum };
    for (auto* b : f4s) b->release();
    c->d_sub_wide.release();
    c->d_sub_nodes.release(); c->d_sub_entries.release(); c->d_sub_nodes_file.release(); c->d_sub_entries_file.release(); c->d_prim_ids.release();
    for (int k = 0; k < 2; ++k) { c->sort_keys[k].release(); c->sort_slots[k].release(); }
    c->sort_temp.release();
    c->d_sub_order_file.release(); c->d_sub_order.release(); c->d_walk_planes.release(); c->d_walk_cubes.release(); c->d_walk_spheres.release();
    c->d_leaves.release(); c->d_tex_desc.release(); c->d_local_tiles.release(); c->n_uv.release(); c->n_pix.release();
    c->link.release(); c->ctr.release(); c->shadow.release(); c->sq.release(); c->stats.release(); c->out_tmp.release();
    for (auto& b : c->adopted_arrays) b.release();
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    c->drop_graphs();
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int rt_set_stream(rt_ctx* c, void* hip_stream) {
    if (!c) return RT_ERR_ARG;
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;   // recorded frames are keyed by stream and stay valid
    return RT_OK;
}

void* rt_get_stream(rt_ctx* c) { return c ? (void*)c->stream : nullptr; }

int rt_set_option(rt_ctx* c, const char* name, int64_t value) {
    if (!c || !name) return RT_ERR_ARG;
    if (std::strcmp(name, "second_level_min_group") == 0) {
        if (value < RT_SUB_LEAF) return c->fail(RT_ERR_ARG, "second_level_min_group must be >= %d", RT_SUB_LEAF);
        c->sub_min_group = value > 0x7fffffff ? 0x7fffffff : (int)value;
        return RT_OK;
    }
    if (std::strcmp(name, "lds_prims") == 0) { c->lds_prims = (int)value; return RT_OK; }
    if (std::strcmp(name, "device_build") == 0) { c->device_build = value != 0; return RT_OK; }
    if (std::strcmp(name, "motion_boxes") == 0) { c->motion_boxes = value != 0; return RT_OK; }
    if (std::strcmp(name, "hierarchy_order") == 0) { c->hierarchy_order = value != 0; return RT_OK; }
    if (std::strcmp(name, "sort_rays") == 0) { c->sort_rays = value != 0; c->drop_graphs(); return RT_OK; }
    if (std::strcmp(name, "sort_rays_min_prims") == 0) { c->sort_rays_min_prims = value < 0 ? 0 : value; c->drop_graphs(); return RT_OK; }
    if (std::strcmp(name, "graphs") == 0) { c->use_graphs = value != 0; c->drop_graphs(); return RT_OK; }
    if (std::strcmp(name, "max_grid") == 0) {
        if (value < 0 || value > (1 << 20)) return c->fail(RT_ERR_ARG, "max_grid must be 0..%d", 1 << 20);
        c->grid_primary = c->grid_secondary = c->grid_shadow = c->grid_shade = (int)value; c->drop_graphs(); return RT_OK;
    }
    if (std::strncmp(name, "grid_", 5) == 0) {
        int* which = std::strcmp(name, "grid_primary") == 0 ? &c->grid_primary : std::strcmp(name, "grid_secondary") == 0 ? &c->grid_secondary :
                     std::strcmp(name, "grid_shadow") == 0 ? &c->grid_shadow : std::strcmp(name, "grid_shade") == 0 ? &c->grid_shade :
                     std::strcmp(name, "grid_shadow_drain") == 0 ? &c->grid_shadow_drain : nullptr;
        if (!which) return c->fail(RT_ERR_ARG, "unknown option %s", name);
        if (value < 0 || value > (1 << 20)) return c->fail(RT_ERR_ARG, "%s must be 0..%d", name, 1 << 20);
        *which = (int)value; c->drop_graphs(); return RT_OK;
    }
    if (std::strcmp(name, "skip_unlit") == 0) { c->skip_unlit = value != 0; c->drop_graphs(); return RT_OK; }
    if (std::strcmp(name, "count_rays") == 0) { c->count_rays = value != 0; return RT_OK; }
    if (std::strcmp(name, "fast_samples") == 0) { c->fast_samples = value != 0; c->drop_graphs(); return RT_OK; }
    if (std::strcmp(name, "cull_cones") == 0) { c->cull_cones = value != 0; c->drop_graphs(); return RT_OK; }
    if (std::strcmp(name, "fuse_simple") == 0) { c->fuse_simple = value != 0; c->drop_graphs(); return RT_OK; }
    if (std::strcmp(name, "shadow_queue_rcap") == 0) { c->shadow_queue_rcap = value > 0 ? (int)std::min<long long>(value, 1 << 24) : 0; c->drop_graphs(); return RT_OK; }
    if (std::strcmp(name, "shadow_queue_deep") == 0) { c->shadow_queue_deep = value != 0; c->drop_graphs(); return RT_OK; }
    if (std::strcmp(name, "shadow_queue") == 0) { c->shadow_queue = value != 0; c->drop_graphs(); return RT_OK; }
    if (std::strcmp(name, "cull_umbra") == 0) { c->cull_umbra = value != 0; c->drop_graphs(); return RT_OK; }
    if (std::strcmp(name, "cull_own") == 0) { c->cull_own = value != 0; c->drop_graphs(); return RT_OK; }
    if (std::strcmp(name, "bundle_reflections") == 0) { c->bundle_reflections = value != 0; c->drop_graphs(); return RT_OK; }
    if (std::strcmp(name, "replicate_reflections") == 0) { c->replicate_reflections = value != 0; c->drop_graphs(); return RT_OK; }
    if (std::strcmp(name, "overlap") == 0) { c->overlap = value != 0; c->drop_graphs(); return RT_OK; }
    if (std::strcmp(name, "shadow_streams") == 0) { c->shadow_streams = value >= 2 ? 2 : 1; c->drop_graphs(); return RT_OK; }
    if (std::strcmp(name, "walk_stats") == 0) { c->walk_stats = value != 0; c->drop_graphs(); return RT_OK; }
    return c->fail(RT_ERR_ARG, "unknown option %s", name);
}

const char* rt_last_error(const rt_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int rt_synchronize(rt_ctx* c) {
    if (!c) return RT_ERR_ARG;
    RT_HIP(c, hipSetDevice(c->device));
    // Frames enqueued without statistics are never looked at by the host; their overflow flag (set by a kernel that
    // found a work buffer full - the capacities are worst cases, so this should be unreachable) is read here.
    unsigned long long flag = 0;
    if (c->last_flag) RT_HIP(c, hipMemcpyAsync(&flag, c->last_flag, sizeof flag, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    release_scratch(c);                             // frames of rt_render_device_noalloc are complete: the GPU's scratch array is free again
    if (flag) return c->fail(RT_ERR_CAPACITY, "a work buffer overflowed during a frame; its image is incomplete");
    return RT_OK;
}

// The records the walk reads (rt_build_wide_records), made from the finished binary trees in `binary`, the final entry
// slots and their testing-order keys: the file-order set's group (when it has a hierarchy) first, then the leaves' groups.
// Assigns every group its wide_base; *file_wide_base receives the file-order group's.
static int make_wide_records(rt_ctx* c, std::vector<BuildGroup>& leaf_groups, const DevPrimSet& leafset, const float4* binary, const int* entries, const int* order) {
    const size_t words = c->sub_stride == 8 ? 16 : 8;
    std::vector<BuildGroup> file_group;
    long long total = 0;
    if (c->file_sub_root >= 0) {
        BuildGroup g{}; g.pn = c->np; g.cn = c->nc; g.sn = c->ns; g.n = c->np + c->nc + c->ns; g.depth = c->file_sub_depth; g.node_base = c->file_sub_root;
        g.wide_base = 0; total = rt_wide_records(g.depth);
        file_group.push_back(g);
    }
    c->file_wide_base = file_group.empty() ? -1 : 0;
    for (BuildGroup& g : leaf_groups) { g.wide_base = (int)total; total += rt_wide_records(g.depth); }
    // the walk addresses a record by a 32-bit byte offset
    if ((unsigned long long)total * words * 16ull >= (1ull << 32))
        return c->fail(RT_ERR_CAPACITY, "second-level hierarchies of %lld records exceed the 4 GiB the walk can address", total);
    c->total_wide = total;
    RT_HIP(c, c->d_sub_wide.resize((size_t)std::max<long long>(total, 1) * words));
    const bool motion = c->sub_stride == 8;
    RT_HIP(c, rt_build_wide_records(c->stream, c->scene.walkset, file_group.data(), (int)file_group.size(), motion, c->sphere_e, binary, entries, order, c->d_sub_wide.p));
    RT_HIP(c, rt_build_wide_records(c->stream, leafset, leaf_groups.data(), (int)leaf_groups.size(), motion, c->sphere_e, binary, entries, order, c->d_sub_wide.p));
    return RT_OK;
}

static int upload_scene_impl(rt_ctx* c, const rt_scene* sc) {
    if (!c || !sc) return RT_ERR_ARG;
    if (sc->n_lights < 0 || sc->n_spheres < 0 || sc->n_cubes < 0 || sc->n_planes < 0 || sc->n_textures < 0 || sc->n_plane_verts < 0)
        return c->fail(RT_ERR_ARG, "negative count in scene");
    if ((sc->n_lights && !sc->lights) || (sc->n_spheres && !sc->spheres) || (sc->n_cubes && !sc->cubes) ||
        (sc->n_planes && !sc->planes) || (sc->n_plane_verts && !sc->plane_verts) || (sc->n_textures && !sc->textures))
        return c->fail(RT_ERR_ARG, "null array with non-zero count in scene");
    if (!(sc->camera.res_x >= 1.0f && sc->camera.res_y >= 1.0f && sc->camera.res_x <= 65536.0f && sc->camera.res_y <= 65536.0f))
        return c->fail(RT_ERR_ARG, "film resolution out of range");
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipStreamSynchronize(c->stream));     // previous frames may still read the old arrays

    c->drop_graphs();
    c->have_scene = false; c->have_bvh = false;
    c->camera = sc->camera;
    c->n_lights = sc->n_lights;
    const int np = sc->n_planes, nc = sc->n_cubes, ns = sc->n_spheres;
    if (np >= (1 << 30) || nc >= (1 << 30) || ns >= (1 << 30) || (long long)np + nc + ns > 0x7fffffffll)
        return c->fail(RT_ERR_ARG, "too many primitives");
    c->np = np; c->nc = nc; c->ns = ns;
    // Groups above sub_min_group primitives get a second-level hierarchy; with "device_build" (default) those are
    // built on the GPU from the uploaded records, otherwise on the host.
    const bool on_device = c->device_build && (long long)np + nc + ns > c->sub_min_group;
    c->built_on_device = on_device;

    std::vector<float4> mats((size_t)(np + nc + ns) * 3);
    std::atomic<int> any_refl{ 0 }, any_transp{ 0 }, any_moving{ 0 }, bad_tex{ -1 }, bad_plane{ -1 };
    auto put_mat = [&](int gid, const rt_material& m) -> bool {
        if (m.tex_id >= sc->n_textures) { bad_tex = gid; return false; }
        // raytracer.cpp:374-375: mean specular times 0.5 (the 0.5 is a double literal; the product is exact), clamped
        float refl = (float)((double)((m.specular[0] + m.specular[1] + m.specular[2]) / 3.0f) * 0.5);
        refl = sminh(smaxh(refl, 0.0f), 1.0f);
        mats[(size_t)gid * 3 + 0] = f4(m.diffuse[0], m.diffuse[1], m.diffuse[2], m.shininess);
        mats[(size_t)gid * 3 + 1] = f4(m.specular[0], m.specular[1], m.specular[2], m.transparency);
        mats[(size_t)gid * 3 + 2] = f4(m.ior, as_f(m.tex_id < 0 ? -1 : m.tex_id), refl, 0.0f);
        if (refl > 0) any_refl = 1;
        if (m.transparency > 0) any_transp = 1;
        return true;
    };

    // every plane owns its vertex range (each range is re-ordered and re-laid out in place below)
    {
        std::vector<char> claimed((size_t)sc->n_plane_verts, 0);
        for (int i = 0; i < np; ++i) {
            const rt_plane& p = sc->planes[i];
            if (p.n_verts < 3 || p.n_verts > 0xffff || p.first_vert < 0 || (long long)p.first_vert + p.n_verts > (long long)sc->n_plane_verts)
                return c->fail(RT_ERR_ARG, "plane %d: vertex range out of bounds", i);
            for (int k = 0; k < p.n_verts; ++k) {
                if (claimed[(size_t)p.first_vert + k]) return c->fail(RT_ERR_ARG, "plane %d: vertex range overlaps another plane's", i);
                claimed[(size_t)p.first_vert + k] = 1;
            }
        }
    }
    // planes: winding-sorted vertices, projection axis, texture axes
    c->h_verts_sorted.assign((size_t)sc->n_plane_verts, f4(0, 0, 0, 0));
    c->h_edges.assign((size_t)sc->n_plane_verts, f4(0, 0, 0, 0));
    c->h_file_planes.assign((size_t)np * 3, f4(0, 0, 0, 0));
    std::vector<float4> plane_uv((size_t)np * 3, f4(0, 0, 0, 0));
    std::vector<float4> plane_bound((size_t)np, f4(0, 0, 0, INFINITY));
    rt_parallel_for((size_t)np, 2048, [&](size_t lo, size_t hi, int) {
        for (size_t i = lo; i < hi; ++i) {
            const rt_plane& p = sc->planes[i];
            if (p.n_verts < 3 || p.n_verts > 0xffff || p.first_vert < 0 || (long long)p.first_vert + p.n_verts > (long long)sc->n_plane_verts) { bad_plane = (int)i; continue; }
            const float* v = sc->plane_verts + (size_t)p.first_vert * 3;
            std::vector<int> ord = winding_order(v, p.n_verts, p.normal);
            for (int k = 0; k < p.n_verts; ++k)
                c->h_verts_sorted[(size_t)p.first_vert + k] = f4(v[3 * ord[k]], v[3 * ord[k] + 1], v[3 * ord[k] + 2], 0.0f);
            const float4 s0 = c->h_verts_sorted[(size_t)p.first_vert];
            const float4 s1 = c->h_verts_sorted[(size_t)p.first_vert + 1];
            const float4 sl = c->h_verts_sorted[(size_t)p.first_vert + p.n_verts - 1];
            int packed = p.n_verts | (drop_axis(p.normal) << 16);
            // File-order set = the objects in the reference's own vectors, which Plane::intersect sorts in place on
            // first use (plane.cpp:158), so the plane offset is taken from the first SORTED vertex there.  Leaves hold
            // copies made before any ray was traced (bvh.cpp:113, :70): their offset comes from the first vertex in
            // FILE order, which rides in the spare lanes of the third word until the leaf-order copy is made.
            c->h_file_planes[i * 3 + 0] = f4(p.normal[0], p.normal[1], p.normal[2], as_f(p.first_vert));
            c->h_file_planes[i * 3 + 1] = f4(s0.x, s0.y, s0.z, as_f(packed));
            c->h_file_planes[i * 3 + 2] = f4(as_f((int)i), v[0], v[1], v[2]);
            // plane.cpp:164-170
            float us[3] = { s1.x - s0.x, s1.y - s0.y, s1.z - s0.z }, vs[3] = { sl.x - s0.x, sl.y - s0.y, sl.z - s0.z };
            float ul = len_ref(us), vl = len_ref(vs);
            rt_normalise_ref(us); rt_normalise_ref(vs);
            plane_uv[i * 3 + 0] = f4(us[0], us[1], us[2], ul);
            plane_uv[i * 3 + 1] = f4(vs[0], vs[1], vs[2], vl);
            plane_uv[i * 3 + 2] = f4(s0.x, s0.y, s0.z, 0.0f);
            {   // sphere around the polygon (k_shadow's cone test): centroid and the farthest vertex, in double
                double cx = 0, cy = 0, cz = 0;
                for (int k = 0; k < p.n_verts; ++k) { cx += v[3 * k]; cy += v[3 * k + 1]; cz += v[3 * k + 2]; }
                cx /= p.n_verts; cy /= p.n_verts; cz /= p.n_verts;
                double r2 = 0;
                for (int k = 0; k < p.n_verts; ++k) {
                    const double dx = v[3 * k] - cx, dy = v[3 * k + 1] - cy, dz = v[3 * k + 2] - cz;
                    r2 = std::max(r2, dx * dx + dy * dy + dz * dz);
                }
                const double rr = std::sqrt(r2) * 1.001 + 1e-6;
                plane_bound[i] = f4((float)cx, (float)cy, (float)cz, (rr == rr && rr < 1e30) ? (float)rr : INFINITY);
            }
            put_mat((int)i, p.mat);
            // the device tests the polygon in the projection that drops the dominant normal axis (plane.cpp:207-216):
            // store every vertex as (kept, kept, dropped)
            const int drop = drop_axis(p.normal);
            for (int k = 0; k < p.n_verts; ++k) {
                float4& q = c->h_verts_sorted[(size_t)p.first_vert + k];
                if (drop == 0) q = f4(q.y, q.z, q.x, 0.0f); else if (drop == 1) q = f4(q.x, q.z, q.y, 0.0f);
            }
            // plane.cpp:228-245: the even-odd test divides (Pj.x - Pi.x) by (Pj.y - Pi.y) for every edge and every ray.
            // The quotient depends on the polygon alone and IEEE division rounds the same here as on the device, so
            // it is formed once; an edge the reference skips (|Pj.y - Pi.y| < 1e-8) carries NaN, which fails the
            // final comparison exactly like the skip.
            for (int k = 0; k < p.n_verts; ++k) {
                const float4 vi = c->h_verts_sorted[(size_t)p.first_vert + k];
                const float4 vj = c->h_verts_sorted[(size_t)p.first_vert + (k + p.n_verts - 1) % p.n_verts];
                const float dy = vj.y - vi.y;
                const float slope = std::fabs(dy) < 1e-8f ? std::numeric_limits<float>::quiet_NaN() : (vj.x - vi.x) / dy;
                c->h_edges[(size_t)p.first_vert + k] = f4(vi.x, vi.y, vj.y, slope);
            }
        }
    });
    if (bad_plane >= 0) return c->fail(RT_ERR_ARG, "plane %d: vertex range out of bounds", bad_plane.load());
    // cubes: rotation sines/cosines from the host libm, exactly the values cube.cpp:301-313 and :328-346 compute per ray
    c->h_file_cubes.assign((size_t)nc * 6, f4(0, 0, 0, 0));
    rt_parallel_for((size_t)nc, 4096, [&](size_t lo, size_t hi, int) {
        for (size_t i = lo; i < hi; ++i) {
            const rt_cube& q = sc->cubes[i];
            float ix = -q.rotation[0], iy = -q.rotation[1], iz = -q.rotation[2];
            float4* w = &c->h_file_cubes[i * 6];
            w[0] = f4(q.start[0], q.start[1], q.start[2], q.scale);
            w[1] = f4(q.end[0], q.end[1], q.end[2], 1.0f / q.scale);
            w[2] = f4(std::cos(q.rotation[0]), std::sin(q.rotation[0]), std::cos(q.rotation[1]), std::sin(q.rotation[1]));
            w[3] = f4(std::cos(q.rotation[2]), std::sin(q.rotation[2]), as_f(np + (int)i), 0.0f);
            w[4] = f4(std::cos(iz), std::sin(iz), std::cos(iy), std::sin(iy));
            w[5] = f4(std::cos(ix), std::sin(ix), 0.0f, 0.0f);
            if (!(q.start[0] == q.end[0] && q.start[1] == q.end[1] && q.start[2] == q.end[2])) any_moving = 1;
            put_mat(np + (int)i, q.mat);
        }
    });
    c->h_file_spheres.assign((size_t)ns * 2, f4(0, 0, 0, 0));
    rt_parallel_for((size_t)ns, 16384, [&](size_t lo, size_t hi, int) {
        for (size_t i = lo; i < hi; ++i) {
            const rt_sphere& s = sc->spheres[i];
            c->h_file_spheres[i * 2 + 0] = f4(s.start[0], s.start[1], s.start[2], s.radius);
            c->h_file_spheres[i * 2 + 1] = f4(s.end[0], s.end[1], s.end[2], as_f(np + nc + (int)i));
            if (!(s.start[0] == s.end[0] && s.start[1] == s.end[1] && s.start[2] == s.end[2])) any_moving = 1;
            put_mat(np + nc + (int)i, s.mat);
        }
    });
    if (bad_tex >= 0) return c->fail(RT_ERR_ARG, "primitive %d (planes, cubes, spheres in file order): texture id out of range", bad_tex.load());
    c->any_reflective = any_refl != 0; c->any_transparent = any_transp != 0;
    // Hierarchies built on the device keep, for scenes in which something moves, boxes per time instead of boxes over
    // the whole motion (8 words per node instead of 4).
    c->sub_stride = (on_device && any_moving != 0 && c->motion_boxes) ? 8 : 4;

    std::vector<float4> lights((size_t)sc->n_lights);
    for (int i = 0; i < sc->n_lights; ++i)
        lights[i] = f4(sc->lights[i].location[0], sc->lights[i].location[1], sc->lights[i].location[2], sc->lights[i].rad_intensity);
    // textures: value / 255.0f as raytracer.cpp:294-296 computes per lookup
    std::vector<int4> tex_desc((size_t)sc->n_textures);
    std::vector<float4> texels;
    for (int t = 0; t < sc->n_textures; ++t) {
        const rt_texture& tx = sc->textures[t];
        if (tx.width < 1 || tx.height < 1 || !tx.rgb) return c->fail(RT_ERR_ARG, "texture %d: bad size or null data", t);
        tex_desc[t] = make_int4(tx.width, tx.height, (int)texels.size(), 0);
        size_t n = (size_t)tx.width * tx.height;
        size_t base = texels.size();
        texels.resize(base + n);
        for (size_t k = 0; k < n; ++k)
            texels[base + k] = f4(tx.rgb[3 * k] / 255.0f, tx.rgb[3 * k + 1] / 255.0f, tx.rgb[3 * k + 2] / 255.0f, 0.0f);
    }

    hipStream_t s = c->stream;
    RT_HIP(c, c->d_mats.upload(mats, s));
    RT_HIP(c, c->d_plane_uv.upload(plane_uv, s));
    RT_HIP(c, c->d_plane_bound.upload(plane_bound, s));
    RT_HIP(c, c->d_verts.upload(c->h_verts_sorted, s));
    RT_HIP(c, c->d_edges.upload(c->h_edges, s));
    RT_HIP(c, c->d_file_planes.upload(c->h_file_planes, s));
    RT_HIP(c, c->d_file_cubes.upload(c->h_file_cubes, s));
    RT_HIP(c, c->d_file_spheres.upload(c->h_file_spheres, s));
    RT_HIP(c, c->d_lights.upload(lights, s));
    RT_HIP(c, c->d_tex_desc.upload(tex_desc, s));
    RT_HIP(c, c->d_texels.upload(texels, s));
    RT_HIP(c, hipStreamSynchronize(s));             // the host vectors above die with this scope
    c->n_textures = sc->n_textures; c->n_texels = (long long)texels.size();
    c->adopted = false; for (auto& b : c->adopted_arrays) b.release();
    c->adopted_arrays.clear(); c->export_arrays.clear(); c->shared_arrays.clear();

    DevScene& d = c->scene;
    std::memset(&d, 0, sizeof d);
    d.fileset.planes = c->d_file_planes.p; d.fileset.verts = c->d_verts.p; d.fileset.edges = c->d_edges.p; d.fileset.cubes = c->d_file_cubes.p;
    d.fileset.spheres = c->d_file_spheres.p; d.fileset.n_planes = np; d.fileset.n_cubes = nc; d.fileset.n_spheres = ns;
    d.leafset = d.fileset;

    // True bounds, the sphere around all of them, and the second-level hierarchy over the whole file-order set.
    // The sphere test forms |l|^2 - tca^2 in float (sphere.cpp:131-139): a difference of two numbers of the size of
    // the squared distance to the centre, so far-away spheres are reported hit by rays that pass a little outside
    // them.  With e = 6e-7 * Dmax^2 bounding that rounding error (Dmax = largest distance between a ray origin - the
    // camera or a point of the scene - and a centre), a reported hit lies within sqrt(r^2 + e) of the centre; the
    // boxes that steer the second-level hierarchy are enlarged accordingly.
    auto sphere_allowance = [&](Box3 reach) {
        reach.add(sc->camera.location);
        double diag2 = 0.0;
        for (int k = 0; k < 3; ++k) { double dd = (double)reach.mx[k] - (double)reach.mn[k] + 0.05; diag2 += dd * dd; }
        return 6e-7 * diag2;
    };
    Box3 all; all.clear();
    c->sphere_e = 0.0; c->file_sub_root = -1; c->n_file_nodes = 0; c->n_file_entries = 0; c->file_sub_depth = 0;
    c->h_sub_nodes_file.clear(); c->h_sub_entries_file.clear();
    c->tb_planes.clear(); c->tb_cubes.clear(); c->tb_spheres.clear();
    if (on_device) {
        float b6[6];
        RT_HIP(c, rt_build_scene_bounds(s, d.fileset, b6));
        Box3 raw; for (int k = 0; k < 3; ++k) { raw.mn[k] = b6[k]; raw.mx[k] = b6[3 + k]; }
        if (ns > 0) c->sphere_e = sphere_allowance(raw);
        BuildGroup g{}; g.pn = np; g.cn = nc; g.sn = ns; g.n = np + nc + ns; g.depth = rt_build_depth(g.n);
        if (g.depth > RT_SUB_MAX_DEPTH) return c->fail(RT_ERR_CAPACITY, "second-level hierarchy deeper than %d", RT_SUB_MAX_DEPTH);
        c->n_file_nodes = 1ll << g.depth; c->n_file_entries = 4ll << g.depth; c->file_sub_depth = g.depth;   // one node slot stays unused
        RT_HIP(c, c->d_sub_nodes_file.resize((size_t)c->n_file_nodes * (size_t)c->sub_stride));
        RT_HIP(c, c->d_sub_entries_file.resize((size_t)c->n_file_entries));
        float fin[6];
        RT_HIP(c, rt_build_hierarchies(s, d.fileset, &g, 1, g.n, c->sphere_e, c->sub_stride == 8, c->d_sub_nodes_file.p, c->d_sub_entries_file.p, fin));
        for (int k = 0; k < 3; ++k) { all.mn[k] = fin[k]; all.mx[k] = fin[3 + k]; }
        c->file_sub_root = 0;
        // the host copies of the records are only needed by the host build
        std::vector<float4>().swap(c->h_file_planes); std::vector<float4>().swap(c->h_file_cubes);
        std::vector<float4>().swap(c->h_file_spheres); std::vector<float4>().swap(c->h_verts_sorted); std::vector<float4>().swap(c->h_edges);
    } else {
        c->tb_planes.resize((size_t)np); c->tb_cubes.resize((size_t)nc); c->tb_spheres.resize((size_t)ns);
        for (int i = 0; i < np; ++i) { c->tb_planes[i] = true_box(sc->planes[i], sc->plane_verts); all.add(c->tb_planes[i]); }
        for (int i = 0; i < nc; ++i) { c->tb_cubes[i] = true_box(sc->cubes[i]); all.add(c->tb_cubes[i]); }
        for (int i = 0; i < ns; ++i) { c->tb_spheres[i] = true_box(sc->spheres[i]); all.add(c->tb_spheres[i]); }
        if (ns > 0) {
            const double e = sphere_allowance(all);
            c->sphere_e = e;
            all.clear();
            for (int i = 0; i < np; ++i) all.add(c->tb_planes[i]);
            for (int i = 0; i < nc; ++i) all.add(c->tb_cubes[i]);
            for (int i = 0; i < ns; ++i) {
                const double r = std::fabs((double)sc->spheres[i].radius);
                const float extra = (float)(std::sqrt(r * r + e) - r) * 1.01f;
                Box3& b = c->tb_spheres[i];
                if (extra < INFINITY) for (int k = 0; k < 3; ++k) { b.mn[k] -= extra; b.mx[k] += extra; }
                all.add(b);
            }
        }
        if (np + nc + ns > c->sub_min_group) {
            std::vector<SubItem> items; items.reserve((size_t)(np + nc + ns));
            for (int i = 0; i < np; ++i) items.push_back(make_item(0, i, c->tb_planes[i]));
            for (int i = 0; i < nc; ++i) items.push_back(make_item(1, i, c->tb_cubes[i]));
            for (int i = 0; i < ns; ++i) items.push_back(make_item(2, i, c->tb_spheres[i]));
            const int depth = rt_build_depth((long long)items.size());
            if (depth > RT_SUB_MAX_DEPTH) return c->fail(RT_ERR_CAPACITY, "second-level hierarchy deeper than %d", RT_SUB_MAX_DEPTH);
            SubBuilder sb{ c->h_sub_nodes_file, c->h_sub_entries_file };
            c->file_sub_root = sb.build_group(items, depth); c->file_sub_depth = depth;
        }
        c->n_file_nodes = (long long)(c->h_sub_nodes_file.size() / 4); c->n_file_entries = (long long)c->h_sub_entries_file.size();
        RT_HIP(c, c->d_sub_nodes_file.upload(c->h_sub_nodes_file, s));
        RT_HIP(c, c->d_sub_entries_file.upload(c->h_sub_entries_file, s));
        RT_HIP(c, hipStreamSynchronize(s));
    }
    // The file-order set as its hierarchy reads it: records in hierarchy order (the file-order arrays themselves stay,
    // the leaf-order copies are gathered from them by file index).
    d.walkset = d.fileset; d.sub_order = nullptr;
    c->reorder_active = c->hierarchy_order != 0;
    if (c->file_sub_root >= 0 && c->reorder_active) {
        BuildGroup g{}; g.pn = np; g.cn = nc; g.sn = ns; g.n = np + nc + ns; g.depth = c->file_sub_depth; g.node_base = c->file_sub_root;
        RT_HIP(c, c->d_walk_planes.resize((size_t)np * 3)); RT_HIP(c, c->d_walk_cubes.resize((size_t)nc * 6)); RT_HIP(c, c->d_walk_spheres.resize((size_t)ns * 2));
        RT_HIP(c, c->d_sub_order_file.resize((size_t)c->n_file_entries));
        RT_HIP(c, rt_build_reorder(s, d.fileset, &g, 1, c->d_sub_entries_file.p, c->d_sub_order_file.p, c->d_walk_planes.p, c->d_walk_cubes.p, c->d_walk_spheres.p));
        d.walkset.planes = c->d_walk_planes.p; d.walkset.cubes = c->d_walk_cubes.p; d.walkset.spheres = c->d_walk_spheres.p;
        d.sub_order = c->d_sub_order_file.p;
    }
    float centre[3] = { 0.0f, 0.0f, 0.0f }, radius = 1.0f;
    if (np + nc + ns > 0) {
        float half[3];
        for (int k = 0; k < 3; ++k) { centre[k] = 0.5f * (all.mn[k] + all.mx[k]); half[k] = 0.5f * (all.mx[k] - all.mn[k]); }
        radius = std::sqrt(half[0] * half[0] + half[1] * half[1] + half[2] * half[2]);
        if (!(radius < 1e30f)) { radius = 1e30f; for (int k = 0; k < 3; ++k) if (!(std::fabs(centre[k]) < 1e30f)) centre[k] = 0.0f; }
    }

    c->total_wide = 0; c->file_wide_base = -1;
    if (c->file_sub_root >= 0) {
        std::vector<BuildGroup> none;
        RT_TRY(make_wide_records(c, none, d.fileset, c->d_sub_nodes_file.p, c->d_sub_entries_file.p, d.sub_order));
    }
    d.mats = c->d_mats.p; d.plane_uv = c->d_plane_uv.p; d.plane_bound = c->d_plane_bound.p; d.tex_desc = c->d_tex_desc.p; d.texels = c->d_texels.p;
    d.lights = c->d_lights.p; d.n_lights = sc->n_lights;
    d.sub_nodes = c->n_file_nodes ? c->d_sub_wide.p : nullptr;
    d.sub_entries = c->d_sub_entries_file.p; d.file_sub_root = c->file_wide_base;
    d.file_sub_depth = c->file_sub_depth; d.sub_stride = c->sub_stride;
    d.sub_stack_entries = d.sub_stack_entries_leaves = rt_sub_stack_entries(c->file_sub_depth);
    d.n_verts = sc->n_plane_verts;
    d.lds_prims = (!d.sub_nodes && rt_lds_prim_words(d) <= RT_LDS_PRIM_MAX_WORDS) ? c->lds_prims : 0;
    for (int k = 0; k < 3; ++k) d.center[k] = centre[k];
    d.radius = radius;
    c->total_sub_nodes = c->n_file_nodes; c->total_sub_entries = c->n_file_entries;
    c->have_scene = true;
    c->rebuild_tiles();
    RT_HIP(c, c->d_local_tiles.upload(c->local_tiles, s));
    RT_HIP(c, hipStreamSynchronize(s));
    return RT_OK;
}

static int upload_bvh_impl(rt_ctx* c, const rt_bvh_node* nodes, int32_t n_nodes, const int32_t* prim_ids, int32_t n_prim_ids) {
    if (!c) return RT_ERR_ARG;
    if (!c->have_scene) return c->fail(RT_ERR_STATE, "rt_upload_bvh before rt_upload_scene");
    if (c->adopted) return c->fail(RT_ERR_STATE, "rt_upload_bvh on a context whose scene was adopted from another one: upload the tree there and adopt again");
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    c->have_bvh = false;
    c->drop_graphs();
    DevScene& d = c->scene;
    d.n_nodes = 0; d.n_leaves = 0;
    const bool reordered = d.walkset.cubes != d.fileset.cubes || d.walkset.spheres != d.fileset.spheres || d.walkset.planes != d.fileset.planes;
    d.sub_order = reordered ? c->d_sub_order_file.p : nullptr;
    c->total_wide = 0;
    if (c->n_file_nodes && c->file_sub_root >= 0) {      // (a tree uploaded earlier may have left the leaves' records here)
        std::vector<BuildGroup> none;
        RT_TRY(make_wide_records(c, none, d.fileset, c->d_sub_nodes_file.p, c->d_sub_entries_file.p, d.sub_order));
    }
    d.sub_nodes = c->n_file_nodes ? c->d_sub_wide.p : nullptr; d.sub_entries = c->d_sub_entries_file.p; d.file_sub_root = c->file_wide_base;
    d.sub_stack_entries = d.sub_stack_entries_leaves = rt_sub_stack_entries(c->file_sub_depth);
    c->total_sub_nodes = c->n_file_nodes; c->total_sub_entries = c->n_file_entries; c->export_arrays.clear();
    int deepest = c->file_sub_depth, deepest_leaf = 0;
    if (!nodes || n_nodes <= 0) return RT_OK;
    if (n_prim_ids < 0 || (n_prim_ids && !prim_ids)) return c->fail(RT_ERR_ARG, "bad prim_ids");
    const int np = c->np, nc = c->nc, ns = c->ns;
    const bool on_device = c->built_on_device;

    // Re-lay the pointer tree out for a stackless walk.  The reference's traversal keeps the hit of the last leaf
    // (in left-to-right depth-first order) that has one, so the device visits right children first and stops at the
    // first leaf with a hit; nodes are stored in exactly that visiting order, each with the index to continue at
    // when its subtree is skipped.  A node with a single child has the same primitive set, hence bit-identical
    // box, as that child (AABB::fromPoints over the same objects): such chains are folded into one test.
    struct Out { float mn[3], mx[3]; int skip, leaf; };
    std::vector<Out> out; std::vector<int4> leaves;
    std::vector<LeafGather> gather;                           // per leaf: where its ids are and where its records go
    std::vector<char> seen((size_t)n_nodes, 0);
    std::string problem;
    long long lp_n = 0, lc_n = 0, ls_n = 0;
    // iterative pre-order, right child first
    struct Item { int node; int phase; int out_idx; };
    std::vector<Item> st; st.push_back({ 0, 0, -1 });
    while (!st.empty()) {
        Item it = st.back(); st.pop_back();
        if (it.phase == 1) { out[(size_t)it.out_idx].skip = (int)out.size(); continue; }
        int ni = it.node;
        if (ni < 0 || ni >= n_nodes) { problem = "child index out of range"; break; }
        if (seen[(size_t)ni]) { problem = "node reachable twice"; break; }
        seen[(size_t)ni] = 1;
        // fold single-child chains with identical boxes
        for (;;) {
            const rt_bvh_node& n = nodes[ni];
            int only = (n.left >= 0 && n.right < 0) ? n.left : ((n.right >= 0 && n.left < 0) ? n.right : -1);
            if (only < 0 || only >= n_nodes) break;
            if (std::memcmp(n.box_min, nodes[only].box_min, 12) != 0 || std::memcmp(n.box_max, nodes[only].box_max, 12) != 0) break;
            if (seen[(size_t)only]) { problem = "node reachable twice"; break; }
            seen[(size_t)only] = 1; ni = only;
        }
        if (!problem.empty()) break;
        const rt_bvh_node& n = nodes[ni];
        Out o{}; std::memcpy(o.mn, n.box_min, 12); std::memcpy(o.mx, n.box_max, 12); o.skip = 0; o.leaf = -1;
        int my = (int)out.size();
        if (n.left < 0 && n.right < 0) {
            long long cnt = (long long)n.n_planes + n.n_cubes + n.n_spheres;
            if (n.n_planes < 0 || n.n_cubes < 0 || n.n_spheres < 0 || n.first_prim < 0 || n.first_prim + cnt > n_prim_ids) { problem = "leaf range out of bounds"; break; }
            if (lp_n + n.n_planes > 0x3fffffffll || lc_n + n.n_cubes > 0x3fffffffll || ls_n + n.n_spheres > 0x3fffffffll) { problem = "too many primitives in leaves"; break; }
            o.leaf = (int)(leaves.size() / 2);
            leaves.push_back(make_int4((int)lp_n, n.n_planes, (int)lc_n, n.n_cubes));
            leaves.push_back(make_int4((int)ls_n, n.n_spheres, -1, 0));
            gather.push_back(LeafGather{ n.first_prim, n.n_planes, n.n_cubes, n.n_spheres, (int)lp_n, (int)lc_n, (int)ls_n, 0 });
            lp_n += n.n_planes; lc_n += n.n_cubes; ls_n += n.n_spheres;
            out.push_back(o);
            out[(size_t)my].skip = my + 1;
        } else {
            out.push_back(o);
            st.push_back({ ni, 1, my });                     // runs after both subtrees: fixes skip
            if (n.left >= 0) st.push_back({ n.left, 0, -1 });  // popped second
            if (n.right >= 0) st.push_back({ n.right, 0, -1 }); // popped first
        }
    }
    if (!problem.empty()) return c->fail(RT_ERR_ARG, "rt_upload_bvh: %s", problem.c_str());
    if (out.size() > RT_MAX_LDS_NODES || leaves.size() / 2 > RT_MAX_LDS_LEAVES)
        return c->fail(RT_ERR_CAPACITY, "BVH has %zu nodes / %zu leaves after folding; at most %d / %d fit the on-chip copy",
                       out.size(), leaves.size() / 2, RT_MAX_LDS_NODES, RT_MAX_LDS_LEAVES);
    // every id a leaf names must exist: the copy kernels index the file-order arrays with them
    {
        std::atomic<int> bad{ -1 };
        rt_parallel_for(gather.size(), 1, [&](size_t lo, size_t hi, int) {
            for (size_t g = lo; g < hi; ++g) {
                const LeafGather& L = gather[g];
                const int32_t* ids = prim_ids + L.first_id;
                for (int k = 0; k < L.n_planes; ++k) if (ids[k] < 0 || ids[k] >= np) bad = 0;
                for (int k = 0; k < L.n_cubes; ++k) if (ids[L.n_planes + k] < 0 || ids[L.n_planes + k] >= nc) bad = 1;
                for (int k = 0; k < L.n_spheres; ++k) if (ids[L.n_planes + L.n_cubes + k] < 0 || ids[L.n_planes + L.n_cubes + k] >= ns) bad = 2;
            }
        });
        static const char* const kind[3] = { "plane", "cube", "sphere" };
        if (bad >= 0) return c->fail(RT_ERR_ARG, "rt_upload_bvh: %s id out of range", kind[bad.load()]);
    }

    std::vector<float4> dn(out.size() * 2);
    for (size_t i = 0; i < out.size(); ++i) {
        dn[2 * i] = f4(out[i].mn[0], out[i].mn[1], out[i].mn[2], as_f(out[i].skip));
        dn[2 * i + 1] = f4(out[i].mx[0], out[i].mx[1], out[i].mx[2], as_f(out[i].leaf));
    }
    hipStream_t s = c->stream;
    DevPrimSet leafset = d.fileset;
    leafset.n_planes = (int)lp_n; leafset.n_cubes = (int)lc_n; leafset.n_spheres = (int)ls_n;
    long long total_nodes = c->n_file_nodes, total_entries = c->n_file_entries;
    std::vector<BuildGroup> groups;                               // leaves that get a second-level hierarchy
    if (on_device) {
        // leaf-order copies of the records, then one build over all leaves that are large enough
        RT_HIP(c, c->d_prim_ids.resize((size_t)n_prim_ids));
        if (n_prim_ids) RT_HIP(c, hipMemcpyAsync(c->d_prim_ids.p, prim_ids, sizeof(int32_t) * (size_t)n_prim_ids, hipMemcpyHostToDevice, s));
        RT_HIP(c, c->d_leaf_planes.resize((size_t)lp_n * 3)); RT_HIP(c, c->d_leaf_cubes.resize((size_t)lc_n * 6));
        RT_HIP(c, c->d_leaf_spheres.resize((size_t)ls_n * 2));
        RT_HIP(c, rt_build_gather_leaves(s, d.fileset, c->d_prim_ids.p, gather.data(), (int)gather.size(),
                                         c->d_leaf_planes.p, c->d_leaf_cubes.p, c->d_leaf_spheres.p));
        leafset.planes = c->d_leaf_planes.p; leafset.cubes = c->d_leaf_cubes.p; leafset.spheres = c->d_leaf_spheres.p;
        long long n_items = 0;
        for (size_t g = 0; g < gather.size(); ++g) {
            const LeafGather& L = gather[g];
            const long long cnt = (long long)L.n_planes + L.n_cubes + L.n_spheres;
            if (cnt <= c->sub_min_group) continue;
            BuildGroup b{}; b.p0 = L.lp0; b.pn = L.n_planes; b.c0 = L.lc0; b.cn = L.n_cubes; b.s0 = L.ls0; b.sn = L.n_spheres;
            b.n = (int)cnt; b.depth = rt_build_depth(cnt); b.item_base = (int)n_items; b.node_base = (int)total_nodes;
            if (b.depth > RT_SUB_MAX_DEPTH) return c->fail(RT_ERR_CAPACITY, "second-level hierarchy deeper than %d", RT_SUB_MAX_DEPTH);
            leaves[2 * g + 1].z = b.node_base; leaves[2 * g + 1].w = b.depth; deepest = std::max(deepest, b.depth); deepest_leaf = std::max(deepest_leaf, b.depth);
            n_items += cnt; total_nodes += 1ll << b.depth;                 // heap order, one slot stays unused
            groups.push_back(b);
        }
        total_entries = 4 * total_nodes;                                  // four entry slots per run, at 4 * (node_base + run)
        if (total_nodes > 0x7fffffffll / 4 || total_entries > 0x7fffffffll) return c->fail(RT_ERR_CAPACITY, "second-level hierarchies too large");
        const size_t stride = (size_t)c->sub_stride;
        RT_HIP(c, c->d_sub_nodes.resize((size_t)total_nodes * stride)); RT_HIP(c, c->d_sub_entries.resize((size_t)total_entries));
        if (c->n_file_nodes) RT_HIP(c, hipMemcpyAsync(c->d_sub_nodes.p, c->d_sub_nodes_file.p, sizeof(float4) * stride * (size_t)c->n_file_nodes, hipMemcpyDeviceToDevice, s));
        if (c->n_file_entries) RT_HIP(c, hipMemcpyAsync(c->d_sub_entries.p, c->d_sub_entries_file.p, sizeof(int) * (size_t)c->n_file_entries, hipMemcpyDeviceToDevice, s));
        RT_HIP(c, rt_build_hierarchies(s, leafset, groups.data(), (int)groups.size(), n_items, c->sphere_e, c->sub_stride == 8,
                                       c->d_sub_nodes.p, c->d_sub_entries.p, nullptr));
    } else {
        std::vector<float4> lp, lc, ls;
        std::vector<float4> sub_nodes = c->h_sub_nodes_file;      // the file-order hierarchy keeps its node indices
        std::vector<int> sub_entries = c->h_sub_entries_file;     // (replaced below by the device's copy when that was re-ordered)
        std::vector<SubItem> items;
        lp.reserve((size_t)lp_n * 3); lc.reserve((size_t)lc_n * 6); ls.reserve((size_t)ls_n * 2);
        for (size_t g = 0; g < gather.size(); ++g) {
            const LeafGather& L = gather[g];
            const int32_t* ids = prim_ids + L.first_id;
            for (int k = 0; k < L.n_planes; ++k) {
                const int id = ids[k];
                const float4 w1 = c->h_file_planes[(size_t)id * 3 + 1], w2 = c->h_file_planes[(size_t)id * 3 + 2];
                lp.push_back(c->h_file_planes[(size_t)id * 3 + 0]);
                lp.push_back(f4(w2.y, w2.z, w2.w, w1.w));        // offset vertex = first vertex in file order
                lp.push_back(f4(w2.x, 0.0f, 0.0f, 0.0f));
            }
            for (int k = 0; k < L.n_cubes; ++k) for (int q = 0; q < 6; ++q) lc.push_back(c->h_file_cubes[(size_t)ids[L.n_planes + k] * 6 + q]);
            for (int k = 0; k < L.n_spheres; ++k) for (int q = 0; q < 2; ++q) ls.push_back(c->h_file_spheres[(size_t)ids[L.n_planes + L.n_cubes + k] * 2 + q]);
            const int cnt = L.n_planes + L.n_cubes + L.n_spheres;
            if (cnt > c->sub_min_group) {           // second-level hierarchy over this leaf's primitives (leaf-order indices)
                items.clear(); items.reserve((size_t)cnt);
                for (int k = 0; k < L.n_planes; ++k) items.push_back(make_item(0, L.lp0 + k, c->tb_planes[(size_t)ids[k]]));
                for (int k = 0; k < L.n_cubes; ++k) items.push_back(make_item(1, L.lc0 + k, c->tb_cubes[(size_t)ids[L.n_planes + k]]));
                for (int k = 0; k < L.n_spheres; ++k) items.push_back(make_item(2, L.ls0 + k, c->tb_spheres[(size_t)ids[L.n_planes + L.n_cubes + k]]));
                const int depth = rt_build_depth(cnt);
                if (depth > RT_SUB_MAX_DEPTH) return c->fail(RT_ERR_CAPACITY, "second-level hierarchy deeper than %d", RT_SUB_MAX_DEPTH);
                SubBuilder sb{ sub_nodes, sub_entries };
                leaves[2 * g + 1].z = sb.build_group(items, depth); leaves[2 * g + 1].w = depth; deepest = std::max(deepest, depth); deepest_leaf = std::max(deepest_leaf, depth);
                BuildGroup b{}; b.p0 = L.lp0; b.pn = L.n_planes; b.c0 = L.lc0; b.cn = L.n_cubes; b.s0 = L.ls0; b.sn = L.n_spheres;
                b.n = cnt; b.depth = depth; b.node_base = leaves[2 * g + 1].z;
                groups.push_back(b);
            }
        }
        total_nodes = (long long)(sub_nodes.size() / 4); total_entries = (long long)sub_entries.size();
        RT_HIP(c, c->d_leaf_planes.upload(lp, s));
        RT_HIP(c, c->d_leaf_cubes.upload(lc, s));
        RT_HIP(c, c->d_leaf_spheres.upload(ls, s));
        RT_HIP(c, c->d_sub_nodes.upload(sub_nodes, s));
        RT_HIP(c, c->d_sub_entries.upload(sub_entries, s));
        // the file-order hierarchy's slots as the scene upload left them on the device (re-ordered or not)
        if (c->n_file_entries) RT_HIP(c, hipMemcpyAsync(c->d_sub_entries.p, c->d_sub_entries_file.p, sizeof(int) * (size_t)c->n_file_entries, hipMemcpyDeviceToDevice, s));
        RT_HIP(c, hipStreamSynchronize(s));
        leafset.planes = c->d_leaf_planes.p; leafset.cubes = c->d_leaf_cubes.p; leafset.spheres = c->d_leaf_spheres.p;
    }
    // Leaves with a hierarchy: their records go into hierarchy order (inside the leaf's own ranges of the leaf-order arrays).
    if (c->reorder_active && (reordered || !groups.empty())) {
        RT_HIP(c, c->d_sub_order.resize((size_t)std::max<long long>(total_entries, 1)));
        if (reordered && c->n_file_entries)
            RT_HIP(c, hipMemcpyAsync(c->d_sub_order.p, c->d_sub_order_file.p, sizeof(int) * (size_t)c->n_file_entries, hipMemcpyDeviceToDevice, s));
        else if (c->n_file_entries)       // not re-ordered: a slot's entry is its own place in the testing order
            RT_HIP(c, hipMemcpyAsync(c->d_sub_order.p, c->d_sub_entries_file.p, sizeof(int) * (size_t)c->n_file_entries, hipMemcpyDeviceToDevice, s));
        if (!groups.empty()) {
            DevBuf<float4> tp, tc, ts;
            hipError_t e = tp.resize((size_t)lp_n * 3);
            if (e == hipSuccess) e = tc.resize((size_t)lc_n * 6);
            if (e == hipSuccess) e = ts.resize((size_t)ls_n * 2);
            if (e == hipSuccess && lp_n) e = hipMemcpyAsync(tp.p, leafset.planes, sizeof(float4) * 3 * (size_t)lp_n, hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess && lc_n) e = hipMemcpyAsync(tc.p, leafset.cubes, sizeof(float4) * 6 * (size_t)lc_n, hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess && ls_n) e = hipMemcpyAsync(ts.p, leafset.spheres, sizeof(float4) * 2 * (size_t)ls_n, hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) e = rt_build_reorder(s, leafset, groups.data(), (int)groups.size(), c->d_sub_entries.p, c->d_sub_order.p, tp.p, tc.p, ts.p);
            if (e != hipSuccess) { tp.release(); tc.release(); ts.release(); return c->hip(e, "records in hierarchy order"); }
            std::swap(c->d_leaf_planes, tp); std::swap(c->d_leaf_cubes, tc); std::swap(c->d_leaf_spheres, ts);
            tp.release(); tc.release(); ts.release();
            leafset.planes = c->d_leaf_planes.p; leafset.cubes = c->d_leaf_cubes.p; leafset.spheres = c->d_leaf_spheres.p;
        }
        d.sub_order = c->d_sub_order.p;
    }
    RT_HIP(c, c->d_nodes.upload(dn, s));
    RT_HIP(c, c->d_leaves.upload(leaves, s));
    RT_HIP(c, hipStreamSynchronize(s));
    if (total_nodes) {
        RT_TRY(make_wide_records(c, groups, leafset, c->d_sub_nodes.p, c->d_sub_entries.p, d.sub_order));
        // a leaf's record names its hierarchy by the first of its records
        size_t gi = 0;
        for (size_t g = 0; g < gather.size() && gi < groups.size(); ++g)
            if (leaves[2 * g + 1].z == groups[gi].node_base && leaves[2 * g + 1].w == groups[gi].depth) { leaves[2 * g + 1].z = groups[gi].wide_base; ++gi; }
        if (gi != groups.size()) return c->fail(RT_ERR_STATE, "rt_upload_bvh: leaves and their hierarchies out of step");
        RT_HIP(c, c->d_leaves.upload(leaves, s));
        RT_HIP(c, hipStreamSynchronize(s));
        c->d_sub_nodes.release();                     // the binary trees of file-order set + leaves: only the records are read from here on
    }
    d.sub_nodes = total_nodes ? c->d_sub_wide.p : nullptr; d.sub_entries = c->d_sub_entries.p; d.file_sub_root = c->file_wide_base;
    c->total_sub_nodes = total_nodes; c->total_sub_entries = total_entries;
    d.nodes = c->d_nodes.p; d.n_nodes = (int)out.size();
    d.leaves = c->d_leaves.p; d.n_leaves = (int)(leaves.size() / 2);
    d.leafset = leafset;
    d.sub_stack_entries = rt_sub_stack_entries(deepest); d.sub_stack_entries_leaves = rt_sub_stack_entries(deepest_leaf);
    if (d.sub_stack_entries > RT_SUB_STACK) return c->fail(RT_ERR_CAPACITY, "second-level hierarchy of depth %d needs more than %d stack entries", deepest, RT_SUB_STACK);
    d.lds_prims = (!d.sub_nodes && rt_lds_prim_words(d) <= RT_LDS_PRIM_MAX_WORDS) ? c->lds_prims : 0;
    c->have_bvh = true;
    return RT_OK;
}

// The uploads build host-side vectors sized by the caller's counts: nothing may be thrown across the C boundary.
int rt_upload_scene(rt_ctx* c, const rt_scene* sc) {
    try { return upload_scene_impl(c, sc); }
    catch (const std::bad_alloc&) { c->have_scene = false; return c->fail(RT_ERR_ARG, "rt_upload_scene: out of host memory for a scene of this size"); }
    catch (const std::exception& e) { c->have_scene = false; return c->fail(RT_ERR_ARG, "rt_upload_scene: %s", e.what()); }
}

int rt_upload_bvh(rt_ctx* c, const rt_bvh_node* nodes, int32_t n_nodes, const int32_t* prim_ids, int32_t n_prim_ids) {
    try { return upload_bvh_impl(c, nodes, n_nodes, prim_ids, n_prim_ids); }
    catch (const std::bad_alloc&) { c->have_bvh = false; return c->fail(RT_ERR_ARG, "rt_upload_bvh: out of host memory for a tree of this size"); }
    catch (const std::exception& e) { c->have_bvh = false; return c->fail(RT_ERR_ARG, "rt_upload_bvh: %s", e.what()); }
}

// ---- the device-resident scene handed from one context to others ----------------------------------------------------
// After the uploads a context holds, in its GPU's memory, every array the kernels read (first-level nodes, leaf records,
// the primitive sets in leaf / file / hierarchy order, polygon edges, materials, texels, lights, the second-level
// hierarchies): DevScene is the table of contents.  Another context - on another GPU of the node, in this process or in
// another rank's - can take over that scene without parsing, building or uploading anything: it adopts the LAYOUT (a
// small POD blob: DevScene with array numbers in place of addresses, the counts that size the arrays, the camera) and
// allocates the arrays; the bytes then travel device to device (ncclBroadcast over xGMI in rt_multi_upload and
// raytracer_amd/dist.py, plain device copies on one GPU).
namespace {

struct PtrSlot { const void** p; size_t bytes; };

// every device pointer of a DevScene with the bytes behind it, always in the same order
void scene_ptr_slots(DevScene& d, long long n_mats, long long n_textures, long long n_texels, long long sub_nodes, long long sub_entries,
                     std::vector<PtrSlot>& out) {
    auto add = [&](const void* const& field, size_t bytes) { out.push_back(PtrSlot{ (const void**)&field, bytes }); };
    add(*(const void**)&d.nodes, (size_t)d.n_nodes * 32); add(*(const void**)&d.leaves, (size_t)d.n_leaves * 32);
    for (DevPrimSet* S : { &d.leafset, &d.fileset, &d.walkset }) {
        add(*(const void**)&S->planes, (size_t)S->n_planes * 48); add(*(const void**)&S->verts, (size_t)d.n_verts * 16);
        add(*(const void**)&S->edges, (size_t)d.n_verts * 16); add(*(const void**)&S->cubes, (size_t)S->n_cubes * 96);
        add(*(const void**)&S->spheres, (size_t)S->n_spheres * 32);
    }
    add(*(const void**)&d.mats, (size_t)n_mats * 48); add(*(const void**)&d.plane_uv, (size_t)d.fileset.n_planes * 48);
    add(*(const void**)&d.plane_bound, (size_t)d.fileset.n_planes * 16);
    add(*(const void**)&d.tex_desc, (size_t)n_textures * 16); add(*(const void**)&d.texels, (size_t)n_texels * 16);
    add(*(const void**)&d.lights, (size_t)d.n_lights * 16);
    add(*(const void**)&d.sub_nodes, (size_t)sub_nodes * (size_t)(d.sub_stride == 8 ? 16 : 8) * 16);     // the walk's records
    // (the entry slots and their testing-order keys are build products; the kernels read the copies inside the records)
    add(*(const void**)&d.sub_entries, 0); add(*(const void**)&d.sub_order, 0); (void)sub_entries;
}

enum { RT_LAYOUT_MAGIC = 0x52544C31, RT_LAYOUT_MAX_SLOTS = 32 };
struct SceneLayout {
    uint32_t magic, bytes;
    DevScene scene;                                  // pointer fields are meaningless outside the exporting process
    rt_camera camera;
    long long np, nc, ns, n_textures, n_texels, sub_nodes, sub_entries;
    int n_lights, any_reflective, any_transparent, have_bvh, sub_stride, file_sub_depth, n_slots, n_arrays;
    int slot_array[RT_LAYOUT_MAX_SLOTS];             // per pointer slot: which array it points into, or -1 (null / nothing behind it)
    unsigned long long array_bytes[RT_LAYOUT_MAX_SLOTS];
};

} // namespace

uint64_t rt_scene_layout_bytes(void) { return sizeof(SceneLayout); }

int rt_export_scene_layout(rt_ctx* c, void* layout, uint64_t cap) {
    if (!c || !layout) return RT_ERR_ARG;
    if (!c->have_scene) return c->fail(RT_ERR_STATE, "rt_export_scene_layout before rt_upload_scene");
    if (cap < sizeof(SceneLayout)) return c->fail(RT_ERR_ARG, "layout buffer of %llu bytes, %zu needed", (unsigned long long)cap, sizeof(SceneLayout));
    SceneLayout L; std::memset(&L, 0, sizeof L);
    L.magic = RT_LAYOUT_MAGIC; L.bytes = (uint32_t)sizeof L;
    L.scene = c->scene; L.scene.walk_stats = nullptr; L.camera = c->camera;
    L.np = c->np; L.nc = c->nc; L.ns = c->ns; L.n_textures = c->n_textures; L.n_texels = c->n_texels;
    L.sub_nodes = c->total_wide; L.sub_entries = c->total_sub_entries;
    L.n_lights = c->n_lights; L.any_reflective = c->any_reflective; L.any_transparent = c->any_transparent; L.have_bvh = c->have_bvh;
    L.sub_stride = c->sub_stride; L.file_sub_depth = c->file_sub_depth;
    std::vector<PtrSlot> slots;
    scene_ptr_slots(L.scene, (long long)c->np + c->nc + c->ns, L.n_textures, L.n_texels, L.sub_nodes, L.sub_entries, slots);
    if (slots.size() > RT_LAYOUT_MAX_SLOTS) return c->fail(RT_ERR_STATE, "layout table too small");
    L.n_slots = (int)slots.size();
    c->export_arrays.clear();
    for (size_t k = 0; k < slots.size(); ++k) {
        const void* ptr = *slots[k].p;
        L.slot_array[k] = -1;
        if (!ptr || !slots[k].bytes) continue;
        size_t a = 0;
        for (; a < c->export_arrays.size(); ++a) if (c->export_arrays[a].first == ptr) break;
        if (a == c->export_arrays.size()) c->export_arrays.push_back({ ptr, slots[k].bytes });
        else c->export_arrays[a].second = std::max(c->export_arrays[a].second, slots[k].bytes);
        L.slot_array[k] = (int)a;
    }
    L.n_arrays = (int)c->export_arrays.size();
    for (int a = 0; a < L.n_arrays; ++a) L.array_bytes[a] = c->export_arrays[(size_t)a].second;
    for (auto& sl : slots) *sl.p = nullptr;          // no address leaves the process
    std::memcpy(layout, &L, sizeof L);
    return RT_OK;
}

// share: the arrays of the layout, in its order, as they exist in another context of the same device: used in place, nothing is allocated or copied
static int adopt_layout_impl(rt_ctx* c, const void* layout, uint64_t bytes, const std::vector<std::pair<const void*, size_t>>* share = nullptr) {
    if (!c || !layout) return RT_ERR_ARG;
    SceneLayout L;
    if (bytes < sizeof L) return c->fail(RT_ERR_ARG, "scene layout of %llu bytes, %zu expected", (unsigned long long)bytes, sizeof L);
    std::memcpy(&L, layout, sizeof L);
    if (L.magic != RT_LAYOUT_MAGIC || L.bytes != sizeof L) return c->fail(RT_ERR_ARG, "not a scene layout of this library version");
    if (L.n_slots < 0 || L.n_slots > RT_LAYOUT_MAX_SLOTS || L.n_arrays < 0 || L.n_arrays > L.n_slots) return c->fail(RT_ERR_ARG, "scene layout: bad table");
    std::vector<PtrSlot> slots;
    scene_ptr_slots(L.scene, L.np + L.nc + L.ns, L.n_textures, L.n_texels, L.sub_nodes, L.sub_entries, slots);
    if ((int)slots.size() != L.n_slots) return c->fail(RT_ERR_ARG, "scene layout: %d pointer slots, %zu expected", L.n_slots, slots.size());
    for (int k = 0; k < L.n_slots; ++k) {
        const int a = L.slot_array[k];
        if (a < -1 || a >= L.n_arrays) return c->fail(RT_ERR_ARG, "scene layout: slot %d names array %d of %d", k, a, L.n_arrays);
        if (a >= 0 && slots[(size_t)k].bytes > L.array_bytes[a]) return c->fail(RT_ERR_ARG, "scene layout: slot %d needs %zu bytes, its array has %llu", k, slots[(size_t)k].bytes, L.array_bytes[a]);
    }
    if (!(L.camera.res_x >= 1.0f && L.camera.res_y >= 1.0f && L.camera.res_x <= 65536.0f && L.camera.res_y <= 65536.0f))
        return c->fail(RT_ERR_ARG, "scene layout: film resolution out of range");
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipStreamSynchronize(c->stream));     // previous frames may still read the old arrays
    c->drop_graphs();
    c->have_scene = false; c->have_bvh = false; c->adopted = false;
    for (auto& b : c->adopted_arrays) b.release();
    c->adopted_arrays.clear(); c->export_arrays.clear(); c->shared_arrays.clear();
    if (share) {
        if ((int)share->size() != L.n_arrays) return c->fail(RT_ERR_ARG, "scene layout: %d arrays, %zu offered", L.n_arrays, share->size());
        c->shared_arrays = *share;
        for (int k = 0; k < L.n_slots; ++k) *slots[(size_t)k].p = L.slot_array[k] >= 0 ? (unsigned char*)const_cast<void*>(c->shared_arrays[(size_t)L.slot_array[k]].first) : nullptr;
    } else {
        c->adopted_arrays.resize((size_t)L.n_arrays);
        for (int a = 0; a < L.n_arrays; ++a) RT_HIP(c, c->adopted_arrays[(size_t)a].resize((size_t)L.array_bytes[a]));
        for (int k = 0; k < L.n_slots; ++k) *slots[(size_t)k].p = L.slot_array[k] >= 0 ? c->adopted_arrays[(size_t)L.slot_array[k]].p : nullptr;
    }
    c->scene = L.scene; c->scene.walk_stats = nullptr;
    c->camera = L.camera; c->np = (int)L.np; c->nc = (int)L.nc; c->ns = (int)L.ns; c->n_lights = L.n_lights;
    c->n_textures = L.n_textures; c->n_texels = L.n_texels; c->total_wide = L.sub_nodes; c->total_sub_nodes = 0; c->total_sub_entries = L.sub_entries;
    c->any_reflective = L.any_reflective != 0; c->any_transparent = L.any_transparent != 0;
    c->sub_stride = L.sub_stride; c->file_sub_depth = L.file_sub_depth;
    // the uploads' host-side copies do not exist here: a tree can only arrive with the next adoption
    c->h_file_planes.clear(); c->h_file_cubes.clear(); c->h_file_spheres.clear(); c->h_verts_sorted.clear(); c->h_edges.clear();
    c->tb_planes.clear(); c->tb_cubes.clear(); c->tb_spheres.clear(); c->h_sub_nodes_file.clear(); c->h_sub_entries_file.clear();
    c->adopted = true; c->have_scene = true; c->have_bvh = L.have_bvh != 0;
    c->rebuild_tiles();
    RT_HIP(c, c->d_local_tiles.upload(c->local_tiles, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_adopt_scene_layout(rt_ctx* c, const void* layout, uint64_t bytes) {
    try { return adopt_layout_impl(c, layout, bytes); }
    catch (const std::exception& e) { if (c) { c->have_scene = false; return c->fail(RT_ERR_ARG, "rt_adopt_scene_layout: %s", e.what()); } return RT_ERR_ARG; }
}

int rt_scene_layout_resolution(const void* layout, uint64_t bytes, int32_t* width, int32_t* height) {
    SceneLayout L;
    if (!layout || !width || !height || bytes < sizeof L) return RT_ERR_ARG;
    std::memcpy(&L, layout, sizeof L);
    if (L.magic != RT_LAYOUT_MAGIC || L.bytes != sizeof L) return RT_ERR_ARG;
    *width = (int32_t)L.camera.res_x; *height = (int32_t)L.camera.res_y;
    return RT_OK;
}

int32_t rt_scene_array_count(const rt_ctx* c) {
    if (!c || !c->have_scene) return 0;
    return (int32_t)(!c->adopted ? c->export_arrays.size() : c->shared_arrays.empty() ? c->adopted_arrays.size() : c->shared_arrays.size());
}

int rt_scene_array(rt_ctx* c, int32_t k, void** dev_ptr, uint64_t* bytes) {
    if (!c || !dev_ptr || !bytes) return RT_ERR_ARG;
    if (k < 0 || k >= rt_scene_array_count(c)) return c->fail(RT_ERR_ARG, "scene array %d of %d (export or adopt a layout first)", k, rt_scene_array_count(c));
    if (c->adopted && !c->shared_arrays.empty()) { *dev_ptr = const_cast<void*>(c->shared_arrays[(size_t)k].first); *bytes = c->shared_arrays[(size_t)k].second; }
    else if (c->adopted) { *dev_ptr = c->adopted_arrays[(size_t)k].p; *bytes = c->adopted_arrays[(size_t)k].n; }
    else { *dev_ptr = const_cast<void*>(c->export_arrays[(size_t)k].first); *bytes = c->export_arrays[(size_t)k].second; }
    return RT_OK;
}

int rt_adopt_scene_from(rt_ctx* dst, rt_ctx* src) {
    if (!dst || !src || dst == src) return RT_ERR_ARG;
    std::vector<unsigned char> blob(sizeof(SceneLayout));
    if (rt_export_scene_layout(src, blob.data(), blob.size()) != RT_OK) return dst->fail(RT_ERR_STATE, "source context: %s", src->err.c_str());
    RT_TRY(rt_adopt_scene_layout(dst, blob.data(), blob.size()));
    RT_HIP(dst, hipSetDevice(dst->device));
    for (size_t a = 0; a < dst->adopted_arrays.size(); ++a)
        RT_HIP(dst, hipMemcpyAsync(dst->adopted_arrays[a].p, src->export_arrays[a].first, src->export_arrays[a].second, hipMemcpyDefault, dst->stream));
    RT_HIP(dst, hipStreamSynchronize(dst->stream));
    return RT_OK;
}

int rt_share_scene_from(rt_ctx* dst, rt_ctx* src) {
    if (!dst || !src || dst == src) return RT_ERR_ARG;
    if (dst->device != src->device) return dst->fail(RT_ERR_ARG, "rt_share_scene_from: contexts on devices %d and %d (use rt_adopt_scene_from)", dst->device, src->device);
    std::vector<unsigned char> blob(sizeof(SceneLayout));
    if (rt_export_scene_layout(src, blob.data(), blob.size()) != RT_OK) return dst->fail(RT_ERR_STATE, "source context: %s", src->err.c_str());
    const std::vector<std::pair<const void*, size_t>> arrays = src->export_arrays;       // (as the export just listed them)
    try { return adopt_layout_impl(dst, blob.data(), blob.size(), &arrays); }
    catch (const std::exception& e) { dst->have_scene = false; return dst->fail(RT_ERR_ARG, "rt_share_scene_from: %s", e.what()); }
}

int rt_set_shard(rt_ctx* c, int32_t rank, int32_t world) {
    if (!c) return RT_ERR_ARG;
    if (world < 1 || rank < 0 || rank >= world) return c->fail(RT_ERR_ARG, "rank %d of world %d", rank, world);
    c->rank = rank; c->world = world;
    c->drop_graphs();
    if (c->have_scene) {
        RT_HIP(c, hipSetDevice(c->device));
        RT_HIP(c, hipStreamSynchronize(c->stream));
        c->rebuild_tiles();
        RT_HIP(c, c->d_local_tiles.upload(c->local_tiles, c->stream));
        RT_HIP(c, hipStreamSynchronize(c->stream));
    }
    return RT_OK;
}

int32_t rt_local_tiles(const rt_ctx* c) { return c ? (int32_t)c->local_tiles.size() : 0; }
int32_t rt_max_tiles_per_rank(const rt_ctx* c) { return c ? c->max_tiles : 0; }

// Per-stage device timing of one frame: an event pair around every launch, taken from a pool that lives in the
// context (no event is created while a frame is being enqueued once the pool is warm).  Only used when the caller
// asked for statistics and the frame is short enough for that to be cheap.
enum { ST_EXTEND_PRIMARY = 0, ST_EXTEND_SECONDARY = 1, ST_SHADOW = 2, ST_SHADE = 3, ST_OTHER = 4, ST_COUNT = 5 };
struct StageTimer {
    struct Span { int a, b; int stage; };
    std::vector<Span> spans;
    std::vector<hipEvent_t>* pool = nullptr;
    size_t used = 0;
    hipStream_t s = nullptr;
    bool on = false;
    unsigned long long launches[ST_COUNT] = { 0, 0, 0, 0, 0 };
    int take() {
        if (used == pool->size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return -1; pool->push_back(e); }
        return (int)used++;
    }
    void reserve(size_t n) { while (on && pool->size() < n) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) { on = false; break; } pool->push_back(e); } }
    void begin(int stage) {
        launches[stage]++;
        if (!on) return;
        Span sp{}; sp.stage = stage; sp.a = take(); sp.b = take();
        if (sp.a < 0 || sp.b < 0) { on = false; return; }
        (void)hipEventRecord((*pool)[(size_t)sp.a], s); spans.push_back(sp);
    }
    void end() { if (on && !spans.empty()) (void)hipEventRecord((*pool)[(size_t)spans.back().b], s); }
    void collect(double ms[ST_COUNT]) {
        for (int i = 0; i < ST_COUNT; ++i) ms[i] = 0.0;
        for (auto& sp : spans) {
            float t = 0.0f;
            if (on && hipEventElapsedTime(&t, (*pool)[(size_t)sp.a], (*pool)[(size_t)sp.b]) == hipSuccess) ms[sp.stage] += t;
        }
        spans.clear(); used = 0;
    }
};

// ---- work plan: batch size, per-level regions and buffer capacities of one frame ---------------------------------
// Worst-case hits per primary sample at every recursion level decide the capacities, so a buffer cannot overflow
// (the kernels still check).  Fan-out of a hit: with glossy reflection, gr_samples reflection rays and one refraction
// ray; with soft shadows (and no glossy reflection) the first reflection of a path is traced gr_samples times (see
// k_extend: each of the reference's identical rays owns its shadow samples), later ones once; otherwise one each.
struct WorkPlan {
    int D = 0, R = 1;
    long long B = 0;                       // pixel slots per batch
    bool fanout = false, grouped = true;
    long long node_off[RT_MAX_LEVELS + 2], link_off[RT_MAX_LEVELS + 2], rcap[RT_MAX_LEVELS + 2];
    long long slot_off[RT_MAX_LEVELS + 2], slot_mul[RT_MAX_LEVELS + 2];    // per-copy data (colours, shadow counts), see rt_types.h
    size_t cap_nodes = 0, cap_slots = 0, cap_links = 0, cap_rays = 0, cap_pix = 0, cap_shadow = 0;
    bool shadow_queue = false;             // soft shadows of a tiny scene: pairs to trace go through queues in memory (DevWork::sq)
    size_t cap_sq = 0; long long sq_rcap = 0;
    long long widest_chunks[RT_MAX_LEVELS + 2];
};

// bytes of the per-batch buffers of a plan, each rounded up to 256
static size_t plan_bytes(const WorkPlan& p, bool textures) {
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    return 4 * al(p.cap_rays * 16) + 3 * al(p.cap_nodes * 16) + al(p.cap_slots * 16) + (textures ? al(p.cap_nodes * 8) : 0) + al(p.cap_pix * 4) +
           al(p.cap_shadow * 4) + al(p.cap_links * 4) + (p.cap_sq ? al(p.cap_sq * 32) : 0);
}

static int make_plan(rt_ctx* c, const rt_config* cfg, int D, int L, bool textures, long long n_slots, unsigned long long budget, WorkPlan& p) {
    const long long gr = cfg->gr_samples;
    const long long refl = c->any_reflective ? 1 : 0, transp = c->any_transparent ? 1 : 0;
    double npp[RT_MAX_LEVELS + 2];         // worst-case nodes per primary sample at level d
    double spn[RT_MAX_LEVELS + 2];         // slots per node of level d
    {
        double a = 1.0, b = 0.0;           // hits whose path has / has not yet had its gr_samples-fold reflection
        const bool replicate = cfg->soft_shadows && !cfg->glossy_reflect && c->replicate_reflections;
        const bool bundle = replicate && gr <= 4096 && c->bundle_reflections;      // as DevFrame::bundle in enqueue_frame
        for (int d = 0; d <= D; ++d) {
            npp[d] = a + b; spn[d] = 1.0;
            if (cfg->glossy_reflect) { a = a * (double)(refl * gr + transp); }
            else if (bundle) { a = a * (double)(refl + transp); if (d >= 1) spn[d] = (double)gr; }     // one node per traced ray
            else if (replicate) { const double a2 = a * (double)transp, b2 = a * (double)(refl * gr) + b * (double)(refl + transp); a = a2; b = b2; }
            else { a = a * (double)(refl + transp); }
        }
    }
    p.D = D;
    p.fanout = D > 0 && (refl || transp);
    double nodes_per = 0, slots_per = 0, widest = 1;
    for (int d = 0; d <= D; ++d) { nodes_per += npp[d]; slots_per += npp[d] * spn[d]; widest = std::max(widest, npp[d]); if (!p.fanout) break; }
    const double links_per = nodes_per - 1.0;
    p.grouped = !cfg->soft_shadows || (cfg->ss_samples <= 64 && (cfg->ss_samples & (cfg->ss_samples - 1)) == 0);
    p.shadow_queue = cfg->soft_shadows && p.grouped && L > 0 && c->shadow_queue && c->cull_cones && !c->scene.sub_nodes && c->np + c->nc + c->ns <= 8;
    const double bytes_per = nodes_per * (16.0 * 3 + (textures ? 8 : 0)) + slots_per * (16.0 + (p.shadow_queue ? 16.0 : 4.0) * std::max(L, 1)) + 4.0 + links_per * 4.0 +
                             (p.fanout ? 2.0 * widest * 32.0 : 0.0);
    // (1/64 of the budget covers the rounding of the regions' capacities, see below)
    long long B = (long long)((double)budget * (63.0 / 64.0) / bytes_per);
    B = std::min(B, n_slots);
    for (int attempt = 0; attempt < 40; ++attempt) {
        if (B >= RT_TILE_PIX) B -= B % RT_TILE_PIX; else B -= B % 64;
        if (B < 64) return c->fail(RT_ERR_CAPACITY, "a work area of %llu bytes cannot hold one wavefront of samples at depth %d (up to %.0f hits per sample)",
                                   budget, D, nodes_per);
        // Per-level parts of the node and link arenas, each cut into R regions with their own counters.  A region of
        // level 0 receives the hits of every R-th chunk of 256 primary rays; a region of level d + 1 the rays spawned by
        // the same region of level d.  R = RT_REGIONS once a batch has 64 chunks per region (the rounding of the
        // capacities then costs under 1/64), fewer for small batches, which issue few atomics anyway.
        const long long chunks0 = (B + RT_BLOCK - 1) / RT_BLOCK;
        int R = 1;
        while (R < RT_REGIONS && chunks0 >= 64ll * (2 * R)) R *= 2;
        long long widest_rcap = 0;
        const long long rcap0 = ((chunks0 + R - 1) / R) * RT_BLOCK;
        std::memset(p.node_off, 0, sizeof p.node_off); std::memset(p.link_off, 0, sizeof p.link_off); std::memset(p.rcap, 0, sizeof p.rcap);
        std::memset(p.slot_off, 0, sizeof p.slot_off);
        for (int d = 0; d < RT_MAX_LEVELS + 2; ++d) p.slot_mul[d] = 1;
        for (int d = 0; d <= D; ++d) {
            p.rcap[d] = (long long)((double)rcap0 * (p.fanout || d == 0 ? npp[d] : 0.0)) + (d ? 64 : 0);
            p.node_off[d + 1] = p.node_off[d] + R * p.rcap[d];
            p.slot_mul[d] = (long long)spn[d];
            p.slot_off[d + 1] = p.slot_off[d] + R * p.rcap[d] * p.slot_mul[d];
            if (d >= 1) { p.link_off[d + 1] = p.link_off[d] + R * p.rcap[d]; widest_rcap = std::max(widest_rcap, p.rcap[d]); }
        }
        p.link_off[D + 2] = p.link_off[D + 1];
        p.R = R; p.B = B;
        p.cap_nodes = (size_t)p.node_off[D + 1];
        p.cap_slots = (size_t)p.slot_off[D + 1];
        p.cap_links = (size_t)p.link_off[D + 1] + 64;
        p.cap_rays = p.fanout ? (size_t)(R * widest_rcap) + 64 : 64;
        p.cap_pix = (size_t)(R * p.rcap[0]);                             // n_pix is indexed by the level-0 node
        p.cap_shadow = p.cap_slots * (size_t)std::max(L, 1);
        // each shadow launch's queue holds an eighth of all pairs (more than are traced in any scene the rules apply to;
        // what does not fit is traced by the workgroup that met it)
        p.sq_rcap = p.shadow_queue ? std::max<long long>(256, (((long long)p.cap_shadow / 8 / R) + 255) & ~255ll) : 0;
        if (p.shadow_queue && c->shadow_queue_rcap) p.sq_rcap = c->shadow_queue_rcap;
        p.cap_sq = (size_t)(p.sq_rcap * R * RT_SHADOWQ_LAUNCHES);
        // 32-bit indexing inside the kernels: node and link indices, node * lights, and the shadow rays of one region of
        // one level (k_shadow numbers them inside such a segment)
        double widest_segment = 0;
        for (int d = 0; d <= D; ++d) widest_segment = std::max(widest_segment, (double)p.rcap[d] * (double)p.slot_mul[d] * std::max(1, L) * (cfg->soft_shadows ? cfg->ss_samples : 1));
        if (p.cap_nodes > 0x7fffffffull || p.cap_slots > 0x7fffffffull || p.cap_links > 0x7fffffffull || (double)p.cap_shadow > 2.0e9 || widest_segment > 2.0e9) {
            if (B <= 64) return c->fail(RT_ERR_CAPACITY, "batch too large for 32-bit indices");
            B /= 2; continue;
        }
        const size_t need = plan_bytes(p, textures);
        if (need <= budget) return RT_OK;
        B = (long long)((double)B * ((double)budget / (double)need) * 0.97);      // the estimate was short: smaller batches
    }
    return c->fail(RT_ERR_CAPACITY, "no batch size fits a work area of %llu bytes", budget);
}

// ---- one entry point that allocates nothing -----------------------------------------------------------------------
// rt_render_device_noalloc takes every work buffer of a frame (ray queues, hit tree, links, shadow counts, pixel sums,
// counters) from this array instead of hipMalloc'd memory.  One copy per GPU, held until the process ends.  The name
// is only a host-side handle: device code must be given the address hipGetSymbolAddress returns for the current
// device, which rt_create looks up once per device.
#define RT_NOALLOC_BYTES (64ull << 20)
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_BYTES];
enum { RT_MAX_DEVICES = 64 };
static unsigned char* g_scratch_ptr[RT_MAX_DEVICES];       // per device: address of that GPU's copy of `scratch`
static std::atomic<int> g_scratch_state[RT_MAX_DEVICES];   // 0 = not looked up, 1 = ready, -1 = lookup failed
// Every context of a GPU shares that GPU's array.  The context whose frames may still be using it owns it from its first
// rt_render_device_noalloc until its next rt_synchronize (or rt_destroy); another context that asks meanwhile is refused
// instead of overwriting a frame in flight.  Frames of ONE context are ordered by its stream and may follow each other.
static std::atomic<rt_ctx*> g_scratch_owner[RT_MAX_DEVICES];

static void release_scratch(rt_ctx* c) {
    if (c->device < 0 || c->device >= RT_MAX_DEVICES) return;
    rt_ctx* me = c;
    if (g_scratch_owner[c->device].compare_exchange_strong(me, nullptr)) c->last_flag = nullptr;   // (the flag word lies in the array)
}

// device must be current
static hipError_t lookup_scratch(int device) {
    if (device < 0 || device >= RT_MAX_DEVICES) return hipErrorInvalidDevice;
    if (g_scratch_state[device].load() == 1) return hipSuccess;
    static std::mutex m;
    std::lock_guard<std::mutex> lock(m);
    if (g_scratch_state[device].load() == 1) return hipSuccess;
    void* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) { g_scratch_state[device] = -1; return e != hipSuccess ? e : hipErrorInvalidSymbol; }
    g_scratch_ptr[device] = (unsigned char*)ptr;
    g_scratch_state[device] = 1;
    return hipSuccess;
}

// Enqueue one frame.  out_dev: device buffer in the layout rt_render_device documents.
// noalloc: the work buffers come from this GPU's `scratch` (through its looked-up address) and nothing is allocated.
static int enqueue_frame(rt_ctx* c, const rt_config* cfg, unsigned char* out_dev, bool want_stats_zero, StageTimer* tm, bool count_stats = true,
                         bool noalloc = false) {
    if (!c->have_scene) return c->fail(RT_ERR_STATE, "render before rt_upload_scene");
    if (cfg->bvh && !c->have_bvh) return c->fail(RT_ERR_STATE, "config asks for the BVH but none was uploaded");
    const int AA = cfg->anti_aliasing ? cfg->aa_samples : 1;
    const int DOF = cfg->dof ? cfg->dof_samples : 1;
    const int MB = cfg->motion_blur ? cfg->mb_samples : 1;
    if (AA < 1 || DOF < 1 || MB < 1) return c->fail(RT_ERR_ARG, "sample counts must be >= 1");
    if ((long long)AA * DOF * MB > (1ll << 30)) return c->fail(RT_ERR_ARG, "too many samples per pixel");
    if (cfg->gr_samples < 1 || cfg->gr_samples >= (1 << 29)) return c->fail(RT_ERR_ARG, "gr_samples must be in 1 .. 2^29 - 1");
    if (cfg->soft_shadows && cfg->ss_samples < 1) return c->fail(RT_ERR_ARG, "ss_samples must be >= 1");
    const int D = cfg->reflect_refract_depth < 0 ? 0 : cfg->reflect_refract_depth;
    if (D > RT_MAX_LEVELS - 4) return c->fail(RT_ERR_ARG, "reflect depth above %d", RT_MAX_LEVELS - 4);
    const int S = AA * DOF * MB;
    const int L = c->n_lights;

    DevFrame f; std::memset(&f, 0, sizeof f);
    rt_camera_basis cb = rt_make_camera_basis(c->camera);
    for (int i = 0; i < 3; ++i) { f.cam_loc[i] = c->camera.location[i]; f.fwd[i] = cb.fwd[i]; f.right[i] = cb.right[i]; f.up[i] = cb.up[i]; }
    f.sw_over_fl = cb.sw_over_fl; f.sh_over_fl = cb.sh_over_fl; f.res_x = c->camera.res_x; f.res_y = c->camera.res_y;
    f.lens_radius = cb.lens_radius; f.focal_distance = c->camera.focal_distance;
    f.width = (int)c->camera.res_x; f.height = (int)c->camera.res_y;
    f.soft_shadows = cfg->soft_shadows != 0; f.ss_samples = cfg->ss_samples; f.glossy = cfg->glossy_reflect != 0; f.gr_samples = cfg->gr_samples;
    f.anti_aliasing = cfg->anti_aliasing != 0; f.aa_n = AA; f.dof_on = cfg->dof != 0; f.dof_n = DOF; f.mb_on = cfg->motion_blur != 0; f.mb_n = MB;
    f.textures = cfg->textures != 0; f.use_bvh = cfg->bvh != 0; f.motion_blur = cfg->motion_blur != 0; f.max_depth = D;
    // the retry is exact and, through the file-order set's hierarchy, cheap at any scene size: on unless opted out
    int fb = cfg->fallback; if (fb == 0) fb = 1;
    f.fallback_full = (fb == 1);
    f.light_radius = cfg->light_radius; f.seed = cfg->seed; f.fast_samples = c->fast_samples ? 1 : 0; f.count_stats = count_stats ? 1 : 0; f.skip_unlit = c->skip_unlit ? 1 : 0;
    f.cull_cones = c->cull_cones ? 1 : 0; f.cull_own = c->cull_own ? 1 : 0; f.cull_umbra = c->cull_umbra ? 1 : 0;
    f.replicate = (cfg->soft_shadows && c->replicate_reflections) ? 1 : 0;
    // identical reflection rays are traced once and recorded gr_samples times (the packed hit record holds 12-bit copy numbers)
    f.bundle = (f.replicate && !cfg->glossy_reflect && cfg->gr_samples <= 4096 && c->bundle_reflections) ? 1 : 0;
    f.tiles_x = c->tiles_x; f.tiles_y = c->tiles_y; f.n_local_tiles = (int)c->local_tiles.size(); f.local_tiles = c->d_local_tiles.p;

    const long long n_slots = (long long)c->local_tiles.size() * RT_TILE_PIX;
    if (n_slots == 0) {
        // more ranks than tiles: this rank's block is all padding
        RT_HIP(c, hipSetDevice(c->device));
        const size_t pad = c->world > 1 ? (size_t)c->max_tiles * RT_TILE_PIX * 3 : (size_t)f.width * (size_t)f.height * 3;
        if (pad) RT_HIP(c, hipMemsetAsync(out_dev, 0, pad, c->stream));
        if (want_stats_zero && !noalloc) { RT_HIP(c, c->stats.resize(RT_S_TOTAL)); RT_HIP(c, hipMemsetAsync(c->stats.p, 0, RT_S_TOTAL * sizeof(unsigned long long), c->stream)); }
        c->err.clear(); c->last_batches = 0;
        return RT_OK;
    }
    hipStream_t s = c->stream;
    RT_HIP(c, hipSetDevice(c->device));

    // Fixed part of the work area (pixel sums, counters, statistics) and the budget left for the per-batch buffers
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t fixed_bytes = al((size_t)n_slots * sizeof(float4)) + al(RT_C_TOTAL * sizeof(int)) + al(RT_S_TOTAL * sizeof(unsigned long long));
    unsigned long long budget = c->workspace_limit;
    unsigned char* arena = nullptr;
    if (noalloc) {
        if (lookup_scratch(c->device) != hipSuccess) return c->fail(RT_ERR_HIP, "the library's scratch array has no address on device %d", c->device);
        arena = g_scratch_ptr[c->device];
        if (fixed_bytes + (1u << 20) > RT_NOALLOC_BYTES)
            return c->fail(RT_ERR_CAPACITY, "rt_render_device_noalloc: a film of %lld pixel slots needs %zu bytes of pixel sums; the scratch array holds %llu bytes "
                                            "(at most %llu pixel slots)", n_slots, fixed_bytes, (unsigned long long)RT_NOALLOC_BYTES,
                           (unsigned long long)((RT_NOALLOC_BYTES - (2u << 20)) / sizeof(float4)));
        budget = RT_NOALLOC_BYTES - fixed_bytes;
    }
    WorkPlan plan;
    RT_TRY(make_plan(c, cfg, D, L, f.textures != 0, n_slots, budget, plan));
    const long long B = plan.B;
    const int R = plan.R;
    const bool grouped = plan.grouped;
    const size_t cap_nodes = plan.cap_nodes, cap_links = plan.cap_links, cap_rays = plan.cap_rays, cap_pix = plan.cap_pix;
    const long long* node_off = plan.node_off; const long long* link_off = plan.link_off; const long long* rcap = plan.rcap;

    // Scenes with second-level hierarchies: the rays of every level >= 1 are sorted by origin cell and direction before
    // they are traced (k_sort_keys).  The sort runs over the level's whole queue, so it is only used while that queue is
    // of the order of the batch itself (not for fan-outs whose worst case dwarfs the rays that exist), and it costs a few
    // tenths of a millisecond per level whatever the count: measured, it pays where the scene's records no longer fit
    // the 256 MB memory-side cache (10^7 primitives: 18 % of the frame), changes nothing at 10^6 primitives and costs
    // 12 % of a 1080p frame of 20 000 - hence the threshold on the primitive count (rt_set_option).
    const bool sorts_scene = c->sort_rays && c->scene.sub_nodes != nullptr && (long long)c->np + c->nc + c->ns >= c->sort_rays_min_prims;
    const bool sorting = !noalloc && sorts_scene && D >= 1 && plan.fanout && cap_rays <= 8 * (size_t)B + 4096;
    size_t sort_temp_bytes = 0;
    if (sorting) RT_HIP(c, rt_sort_pairs_temp_bytes(cap_rays, &sort_temp_bytes));

    DevWork w; std::memset(&w, 0, sizeof w);
    if (noalloc) {
        // the frame fits: take the GPU's array unless another context's frames may still be in it
        rt_ctx* owner = nullptr;
        const bool took = g_scratch_owner[c->device].compare_exchange_strong(owner, c);
        if (!took && owner != c)
            return c->fail(RT_ERR_STATE, "rt_render_device_noalloc: another context's frame may still be using this GPU's scratch array "
                                         "(call rt_synchronize on that context first)");
        // carve the scratch block: fixed part first, then the batch buffers (sizes as in plan_bytes)
        unsigned char* q = arena;
        auto take = [&](size_t bytes) { unsigned char* r = q; q += al(bytes); return r; };
        w.accum = (float4*)take((size_t)n_slots * sizeof(float4));
        w.ctr = (int*)take(RT_C_TOTAL * sizeof(int));
        w.stats = (unsigned long long*)take(RT_S_TOTAL * sizeof(unsigned long long));
        for (int k = 0; k < 2; ++k) { w.q_a[k] = (float4*)take(cap_rays * 16); w.q_b[k] = (float4*)take(cap_rays * 16); }
        w.n_a = (float4*)take(cap_nodes * 16); w.n_b = (float4*)take(cap_nodes * 16); w.n_c = (float4*)take(cap_nodes * 16);
        w.n_col = (float4*)take(plan.cap_slots * 16);
        if (f.textures) w.n_uv = (float2*)take(cap_nodes * 8);
        w.n_pix = (int*)take(cap_pix * 4);
        w.shadow = (float*)take(plan.cap_shadow * 4);
        w.link = (int*)take(cap_links * 4);
        if (plan.cap_sq) w.sq = (int4*)take(plan.cap_sq * 32);
        if ((size_t)(q - arena) > RT_NOALLOC_BYTES) {
            if (took) release_scratch(c);
            return c->fail(RT_ERR_CAPACITY, "rt_render_device_noalloc: work plan exceeds the scratch array");
        }
    } else {
        const bool grow = cap_nodes > c->n_a.n || plan.cap_slots > c->n_col.n || cap_links > c->link.n || cap_rays > c->q_a[0].n || (size_t)n_slots > c->accum.n ||
                          (f.textures && cap_nodes > c->n_uv.n) || plan.cap_shadow > c->shadow.n || cap_pix > c->n_pix.n ||
                          (sorting && (cap_rays > c->sort_keys[0].n || sort_temp_bytes > c->sort_temp.n)) || 2 * plan.cap_sq > c->sq.n;
        if (grow) {
            // Recorded frames hold the addresses of the buffers that are about to be freed: drop every recording before any
            // reallocation, and wait for the frames still in flight on either stream.
            c->drop_graphs();
            RT_HIP(c, hipStreamSynchronize(s));
            if (c->aux_stream) RT_HIP(c, hipStreamSynchronize(c->aux_stream));
            if (c->aux_stream2) RT_HIP(c, hipStreamSynchronize(c->aux_stream2));
        }
        for (int q = 0; q < 2; ++q) { RT_HIP(c, c->q_a[q].resize(cap_rays)); RT_HIP(c, c->q_b[q].resize(cap_rays)); }
        RT_HIP(c, c->n_a.resize(cap_nodes)); RT_HIP(c, c->n_b.resize(cap_nodes)); RT_HIP(c, c->n_c.resize(cap_nodes));
        RT_HIP(c, c->n_col.resize(plan.cap_slots));
        if (f.textures) RT_HIP(c, c->n_uv.resize(cap_nodes));
        RT_HIP(c, c->n_pix.resize(cap_pix));
        RT_HIP(c, c->shadow.resize(plan.cap_shadow));
        RT_HIP(c, c->link.resize(cap_links));
        if (plan.cap_sq) RT_HIP(c, c->sq.resize(2 * plan.cap_sq));
        if (sorting) {
            for (int q = 0; q < 2; ++q) { RT_HIP(c, c->sort_keys[q].resize(cap_rays)); RT_HIP(c, c->sort_slots[q].resize(cap_rays)); }
            RT_HIP(c, c->sort_temp.resize(sort_temp_bytes));
        }
        RT_HIP(c, c->accum.resize((size_t)n_slots));
        RT_HIP(c, c->ctr.resize(RT_C_TOTAL));
        RT_HIP(c, c->stats.resize(RT_S_TOTAL));
        for (int q = 0; q < 2; ++q) { w.q_a[q] = c->q_a[q].p; w.q_b[q] = c->q_b[q].p; }
        w.n_a = c->n_a.p; w.n_b = c->n_b.p; w.n_c = c->n_c.p; w.n_uv = c->n_uv.p; w.n_col = c->n_col.p; w.n_pix = c->n_pix.p;
        w.shadow = c->shadow.p; w.link = c->link.p; w.accum = c->accum.p; w.ctr = c->ctr.p; w.stats = c->stats.p;
        w.sq = plan.cap_sq ? c->sq.p : nullptr;
    }
    w.sq_rcap = (int)plan.sq_rcap;
    w.cap_nodes = (unsigned)cap_nodes;
    c->scene.walk_stats = c->walk_stats ? w.stats + RT_S_WALK : nullptr;
    c->last_flag = w.stats + RT_S_ERR;
    for (int d = 0; d <= D + 2 && d < RT_MAX_LEVELS; ++d) {
        w.node_off[d] = (int)node_off[(size_t)std::min(d, D + 1)]; w.link_off[d] = (int)link_off[(size_t)d];
        w.node_rcap[d] = d <= D ? (int)rcap[(size_t)d] : 0; w.link_rcap[d] = (d >= 1 && d <= D) ? (int)rcap[(size_t)d] : 0;
        w.slot_off[d] = (int)plan.slot_off[(size_t)std::min(d, D + 1)]; w.slot_mul[d] = (int)plan.slot_mul[(size_t)std::min(d, D + 1)];
    }
    w.regions = R;

    const int tile_major = c->world > 1;
    const size_t out_bytes = tile_major ? (size_t)c->max_tiles * RT_TILE_PIX * 3 : (size_t)f.width * (size_t)f.height * 3;
    // One sample per pixel in a single batch: the level-0 shading writes the pixels itself; the frame start then zeroes
    // the image (pixels without a hit stay black) instead of the per-pixel sums, and there is no finalize pass.
    const bool direct = S == 1 && B >= n_slots && ((uintptr_t)out_dev % 16) == 0;
    f.direct_out = direct ? out_dev : nullptr; f.direct_tile_major = tile_major;
    if (direct) rt_launch_begin_frame(s, w, out_dev, out_bytes, want_stats_zero);
    else rt_launch_begin_frame(s, w, w.accum, (size_t)n_slots * sizeof(float4), want_stats_zero);

    unsigned long long batches = 0;
    StageTimer local_tm; if (!tm) { tm = &local_tm; tm->pool = &c->event_pool; }
    tm->s = s;
    const unsigned long long n_batches = (unsigned long long)((n_slots + B - 1) / B) * (unsigned long long)S;
    if (n_batches > 64) tm->on = false;
    tm->reserve((size_t)n_batches * (size_t)(2 * D + 4) * 2 + 2);
    const bool fused = c->fuse_simple && D == 0 && !cfg->soft_shadows;
    const bool split = L > 0 && D >= 1 && plan.fanout;                 // shadow rays in two launches: level 0, then the rest
    const bool overlap = split && c->overlap && !tm->on;         // stage times are only meaningful on one stream
    const bool fork1 = overlap && D >= 2 && c->shadow_streams >= 2;
    for (long long begin = 0; begin < n_slots; begin += B) {
        const int count = (int)std::min<long long>(B, n_slots - begin);
        f.pix_begin = (int)begin; f.pix_count = count;
        const int chunks0_batch = (count + RT_BLOCK - 1) / RT_BLOCK;
        // Workgroups per launch; more work than that is taken in rounds.  1024 is what is resident at once (256 CUs x 4
        // workgroups of 4 wavefronts): for kernels whose workgroups are cheap to start and whose rounds cost about the
        // same, a second generation of workgroups only adds start-up cost and an uneven tail.  Launches with very many
        // chunks of unequal cost (walks of second-level hierarchies, tens of thousands of chunks of glossy rays) do
        // better when the dispatcher hands out a second generation as workgroups finish; where chunk costs differ by an
        // order of magnitude - primary rays through second-level hierarchies, soft-shadow chunks of which some trace every
        // pair and others none - one workgroup per chunk or two (8192) evens the load out best.  Measured per kernel on
        // scene.json, bvh_vs_unaccel_5, cube_mb and 10^6 primitives; rt_set_option("grid_*") overrides.
        const bool sub = c->scene.sub_nodes != nullptr;
        const int g_primary = c->grid_primary ? c->grid_primary : (sub ? 8192 : (c->scene.lds_prims ? 1024 : 2048));   // 1024: a handful of primitives
        const int g_shade = c->grid_shade ? c->grid_shade : 1024;
        // (with the queue of pairs to trace the classifying launch's chunks cost about the same: one resident generation)
        const int g_shadow = c->grid_shadow ? c->grid_shadow : (plan.shadow_queue ? 1280 : ((!sub && cfg->soft_shadows && c->fast_samples) ? 8192 : 2048));
        const int g_drain = c->grid_shadow_drain ? c->grid_shadow_drain : 2048;
        auto g_secondary = [&](int level) {
            if (c->grid_secondary) return c->grid_secondary;
            const long long worst_chunks = (long long)w.regions * w.link_rcap[level] / RT_BLOCK;
            return sub ? 8192 : (worst_chunks > 16384 ? 2048 : 1024);
        };
        const int grid0 = std::max(1, std::min(g_primary, chunks0_batch));
        for (int smp = 0; smp < S; ++smp) {
            f.sample = smp;
            if (fused) {
                // no recursion, hard shadows: the whole batch is one launch (k_primary_shaded)
                tm->begin(ST_EXTEND_PRIMARY);
                // (chunks with and without hits differ widely in cost: more, shorter-lived workgroups than the extend kernel's;
                // 4K at 64 spp 22.8 -> 20.4 ms, the default frame 0.096 -> 0.090 ms alone - and 0.056 -> 0.062 with four in flight)
                const int g_fused = c->grid_primary ? c->grid_primary : (sub ? 8192 : 4096);
                rt_launch_primary_shaded(s, c->scene, f, w, std::max(1, std::min(g_fused, chunks0_batch)));
                tm->end();
                ++batches;
                continue;
            }
            if (batches > 0) RT_HIP(c, hipMemsetAsync(w.ctr, 0, RT_C_TOTAL * sizeof(int), s));   // counters of the batch
            if (!grouped) RT_HIP(c, hipMemsetAsync(w.shadow, 0, plan.cap_shadow * sizeof(float), s));
            for (int level = 0; level <= D; ++level) {
                tm->begin(level == 0 ? ST_EXTEND_PRIMARY : ST_EXTEND_SECONDARY);
                w.perm = nullptr;
                const size_t n_queue = level >= 1 ? (size_t)w.regions * (size_t)w.link_rcap[level] : 0;
                if (sorting && n_queue > 0) {
                    rt_launch_sort_keys(s, c->scene, w, level, c->sort_keys[0].p, c->sort_slots[0].p);
                    RT_HIP(c, rt_sort_pairs(s, c->sort_temp.p, c->sort_temp.n, c->sort_keys[0].p, c->sort_keys[1].p, c->sort_slots[0].p, c->sort_slots[1].p,
                                            n_queue, 30));
                    w.perm = c->sort_slots[1].p;
                }
                rt_launch_extend(s, c->scene, f, w, level, level == 0 ? grid0 : g_secondary(level));
                tm->end();
                if (level == 0 && split) {
                    // the primary hits are final: their shadow rays start now, beside the deeper levels' tracing
                    hipStream_t ss = overlap ? c->aux_stream : s;
                    if (overlap) { RT_HIP(c, hipEventRecord(c->ev_fork, s)); RT_HIP(c, hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0)); }
                    tm->begin(ST_SHADOW); tm->launches[ST_SHADOW] += rt_launch_shadow(ss, c->scene, f, w, 0, 0, g_shadow, g_drain) - 1; tm->end();
                    if (overlap) RT_HIP(c, hipEventRecord(c->ev_join, c->aux_stream));
                }
                if (level == 1 && fork1) {
                    // so are the first reflections' and refractions' hits, the other large level (the levels below hold a
                    // few per cent of a frame's hits): the launches that trace those, and the shading of the deep levels,
                    // wait on memory most of the time and run beside the shadow rays' arithmetic
                    RT_HIP(c, hipEventRecord(c->ev_fork2, s)); RT_HIP(c, hipStreamWaitEvent(c->aux_stream2, c->ev_fork2, 0));
                    tm->launches[ST_SHADOW] += rt_launch_shadow(c->aux_stream2, c->scene, f, w, 1, 1, g_shadow, g_drain);
                    RT_HIP(c, hipEventRecord(c->ev_join2, c->aux_stream2));
                }
            }
            if (L > 0) { tm->begin(ST_SHADOW); tm->launches[ST_SHADOW] += rt_launch_shadow(s, c->scene, f, w, fork1 ? 2 : (split ? 1 : 0), D, g_shadow, (fork1 && !c->shadow_queue_deep) ? 0 : g_drain) - 1; tm->end(); }
            for (int level = D; level >= 0; --level) {
                // only the shading of the levels whose shadow rays were forked needs those launches; the deeper levels
                // resolve beside them
                if (fork1 && level == 1) RT_HIP(c, hipStreamWaitEvent(s, c->ev_join2, 0));
                if (overlap && level == 0) RT_HIP(c, hipStreamWaitEvent(s, c->ev_join, 0));
                tm->begin(ST_SHADE);
                rt_launch_shade(s, c->scene, f, w, level, level == 0 ? std::max(1, std::min(g_shade, chunks0_batch)) : g_shade);
                tm->end();
            }
            ++batches;
        }
    }
    if (!direct) {
        if (tile_major) {
            size_t used = (size_t)n_slots * 3;
            if (out_bytes > used) RT_HIP(c, hipMemsetAsync(out_dev + used, 0, out_bytes - used, s));
        }
        tm->begin(ST_OTHER);
        rt_launch_finalize(s, f, w, out_dev, tile_major, (int)n_slots, S);
        tm->end();
    }
    RT_HIP(c, hipGetLastError());
    c->err.clear();
    c->last_batches = batches;
    return RT_OK;
}

static int finish_stats(rt_ctx* c, rt_stats* st, double ms, StageTimer* tm) {
    unsigned long long hs[RT_S_TOTAL];
    RT_HIP(c, hipMemcpyAsync(hs, c->stats.p, sizeof hs, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    if (st) {
        std::memset(st, 0, sizeof *st);
        st->primary_rays = hs[RT_S_PRIMARY]; st->shadow_rays = hs[RT_S_SHADOW]; st->reflection_rays = hs[RT_S_REFLECT];
        st->refraction_rays = hs[RT_S_REFRACT]; st->nodes_shaded = hs[RT_S_NODES];
        st->shadow_rays_ref = hs[RT_S_SHADOW_REF]; st->reflection_rays_ref = hs[RT_S_REFLECT_REF]; st->refraction_rays_ref = hs[RT_S_REFRACT_REF];
        st->batches = c->last_batches; st->ms_total = ms;
        double sm[ST_COUNT]; tm->collect(sm);
        st->ms_extend_primary = sm[ST_EXTEND_PRIMARY]; st->ms_extend_secondary = sm[ST_EXTEND_SECONDARY];
        st->ms_shadow = sm[ST_SHADOW]; st->ms_shade = sm[ST_SHADE]; st->ms_other = sm[ST_OTHER];
        st->launches_extend_primary = tm->launches[ST_EXTEND_PRIMARY]; st->launches_extend_secondary = tm->launches[ST_EXTEND_SECONDARY];
        st->launches_shadow = tm->launches[ST_SHADOW]; st->launches_shade = tm->launches[ST_SHADE];
    } else { double sm[ST_COUNT]; tm->collect(sm); }
    if (hs[RT_S_ERR]) return c->fail(RT_ERR_CAPACITY, "a work buffer overflowed during the frame; the image is incomplete");
    return RT_OK;
}

int rt_render_device(rt_ctx* c, const rt_config* cfg, void* rgb_out_dev, rt_stats* stats) {
    if (!c || !cfg || !rgb_out_dev) return RT_ERR_ARG;
    auto t0 = std::chrono::steady_clock::now();
    const long long spp = (long long)(cfg->anti_aliasing ? cfg->aa_samples : 1) * (cfg->dof ? cfg->dof_samples : 1) *
                          (cfg->motion_blur ? cfg->mb_samples : 1);
    bool remember = false;
    // (frames that sort their secondary rays are long and enqueue a library's sort per level: they are not recorded)
    const bool sorts = c->sort_rays && c->scene.sub_nodes != nullptr && (long long)c->np + c->nc + c->ns >= c->sort_rays_min_prims && cfg->reflect_refract_depth >= 1;
    if (!stats && c->use_graphs && spp >= 1 && spp <= 64 && !sorts) {
        rt_ctx::FrameGraph* g = nullptr;
        for (auto& e : c->graphs)
            if (e.out == rgb_out_dev && e.stream == c->stream && e.generation == c->generation && std::memcmp(&e.cfg, cfg, sizeof *cfg) == 0) { g = &e; break; }
        if (g && g->exec) { RT_HIP(c, hipSetDevice(c->device)); RT_HIP(c, hipGraphLaunch(g->exec, c->stream)); return RT_OK; }
        if (g && g->warm) {
            // second request: every buffer exists, so recording the frame allocates and synchronises nothing
            RT_HIP(c, hipSetDevice(c->device));
            hipGraph_t graph = nullptr;
            if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                const unsigned long long gen = c->generation;
                int rc = enqueue_frame(c, cfg, (unsigned char*)rgb_out_dev, true, nullptr, false);
                hipError_t e = hipStreamEndCapture(c->stream, &graph);
                if (gen != c->generation) {                  // the list was dropped under us (cannot happen: the buffers exist)
                    if (graph) (void)hipGraphDestroy(graph);
                    (void)hipGetLastError();
                    return rc != RT_OK ? rc : c->fail(RT_ERR_STATE, "work buffers changed while a frame was being recorded");
                }
                if (rc == RT_OK && e == hipSuccess && graph && c->last_batches <= 64 &&
                    hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                    (void)hipGraphDestroy(graph);
                    RT_HIP(c, hipGraphLaunch(g->exec, c->stream));
                    return RT_OK;
                }
                if (graph) (void)hipGraphDestroy(graph);
                g->exec = nullptr; g->warm = false;          // not capturable (or too long): plain launches from now on
                (void)hipGetLastError();
                if (rc != RT_OK) return rc;
            } else (void)hipGetLastError();
        } else if (!g) remember = true;
    }
    StageTimer tm; tm.on = stats != nullptr; tm.pool = &c->event_pool;
    int rc = enqueue_frame(c, cfg, (unsigned char*)rgb_out_dev, true, &tm, stats != nullptr && c->count_rays);
    if (rc != RT_OK) { double sm[ST_COUNT]; (void)hipStreamSynchronize(c->stream); tm.collect(sm); return rc; }
    if (remember) {
        // first request of this frame: its buffers exist now (a frame that had to grow them dropped every older
        // recording on the way), so the next request can be recorded without allocating or synchronising
        if (c->graphs.size() >= 8) { if (c->graphs.front().exec) (void)hipGraphExecDestroy(c->graphs.front().exec); c->graphs.erase(c->graphs.begin()); }
        rt_ctx::FrameGraph e{}; e.cfg = *cfg; e.out = rgb_out_dev; e.stream = c->stream; e.generation = c->generation; e.warm = true; e.exec = nullptr;
        c->graphs.push_back(e);
    }
    if (!stats) return RT_OK;
    RT_HIP(c, hipStreamSynchronize(c->stream));
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return finish_stats(c, stats, ms, &tm);
}

int rt_render_device_noalloc(rt_ctx* c, const rt_config* cfg, void* rgb_out_dev) {
    if (!c || !cfg || !rgb_out_dev) return RT_ERR_ARG;
    // plain launches (instantiating a frame graph allocates), no statistics (the event pool may have to grow)
    return enqueue_frame(c, cfg, (unsigned char*)rgb_out_dev, true, nullptr, false, true);
}

uint64_t rt_noalloc_capacity(void) { return RT_NOALLOC_BYTES; }

// Development aid, not part of include/rt_api.h: the step counters of the last frame rendered with the option
// "walk_stats" by a library whose kernels were built with -DRT_WALK_STATS (zeros otherwise).  out[k] = lanes, out[16 + k] = wavefronts.
int rt_debug_walk_stats(rt_ctx* c, uint64_t* out32) {
    if (!c || !out32) return RT_ERR_ARG;
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    if (!c->stats.p || c->stats.n < RT_S_TOTAL) { std::memset(out32, 0, 32 * sizeof(uint64_t)); return RT_OK; }
    RT_HIP(c, hipMemcpy(out32, c->stats.p + RT_S_WALK, 32 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_render(rt_ctx* c, const rt_config* cfg, uint8_t* rgb_out, rt_stats* stats) {
    if (!c || !cfg || !rgb_out) return RT_ERR_ARG;
    if (!c->have_scene) return c->fail(RT_ERR_STATE, "render before rt_upload_scene");
    RT_HIP(c, hipSetDevice(c->device));
    const size_t bytes = c->world > 1 ? (size_t)c->max_tiles * RT_TILE_PIX * 3
                                      : (size_t)((int)c->camera.res_x) * (size_t)((int)c->camera.res_y) * 3;
    if (bytes > c->out_tmp.n) { RT_HIP(c, hipStreamSynchronize(c->stream)); RT_HIP(c, c->out_tmp.resize(bytes)); }
    auto t0 = std::chrono::steady_clock::now();
    StageTimer tm; tm.on = stats != nullptr; tm.pool = &c->event_pool;
    int rc = enqueue_frame(c, cfg, c->out_tmp.p, true, &tm, stats != nullptr && c->count_rays);
    if (rc != RT_OK) { double sm[ST_COUNT]; (void)hipStreamSynchronize(c->stream); tm.collect(sm); return rc; }
    RT_HIP(c, hipStreamSynchronize(c->stream));
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    RT_HIP(c, hipMemcpy(rgb_out, c->out_tmp.p, bytes, hipMemcpyDeviceToHost));
    return finish_stats(c, stats, ms, &tm);
}

void* rt_alloc_image(uint64_t bytes) {
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

void rt_free_image(void* p) { if (p) (void)hipHostFree(p); }

int rt_assemble_tiles(rt_ctx* c, const void* tiles_dev, int32_t world, void* rgb_out_dev) {
    if (!c || !tiles_dev || !rgb_out_dev || world < 1) return RT_ERR_ARG;
    if (!c->have_scene) return c->fail(RT_ERR_STATE, "assemble before rt_upload_scene");
    if (world != c->world) return c->fail(RT_ERR_ARG, "world %d differs from the context's shard setting %d", world, c->world);
    RT_HIP(c, hipSetDevice(c->device));
    if (world == 1) {
        // an unsharded context writes the row-major image itself: its one "block" is copied, not re-ordered
        if (tiles_dev != rgb_out_dev)
            RT_HIP(c, hipMemcpyAsync(rgb_out_dev, tiles_dev, (size_t)c->camera.res_x * (size_t)c->camera.res_y * 3, hipMemcpyDeviceToDevice, c->stream));
        return RT_OK;
    }
    rt_launch_assemble(c->stream, (const unsigned char*)tiles_dev, world, c->max_tiles, c->tiles_x, c->tiles_y,
                       (int)c->camera.res_x, (int)c->camera.res_y, (unsigned char*)rgb_out_dev);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

} // extern "C"
