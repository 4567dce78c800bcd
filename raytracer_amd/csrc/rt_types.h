// Device-side layouts shared by the kernels (rt_kernels.hip) and the context (rt_context.hip).
// Everything lives in HBM as arrays of 16-byte words so that a lane's record is one or a few dwordx4 loads.
#ifndef RT_TYPES_H
#define RT_TYPES_H

#include <hip/hip_runtime.h>
#include <stdint.h>

// Scenes with moving primitives, 8 words per second-level node: where node h (heap index inside its group) keeps its
// time-0 words, relative to the group's first word; its motion words follow 8 words later.  The root has a block of 16
// words to itself, after it two siblings share one: both time-0 records (one 128-byte line), then both motion records.
__host__ __device__ inline size_t rt_sub_motion_offset(int h) {
    return h ? 16 + 16 * (size_t)((h - 1) >> 1) + 4 * (size_t)((h - 1) & 1) : 0;
}

#define RT_TILE        32            // image tiles are RT_TILE x RT_TILE pixels; a wavefront covers two tile rows
#define RT_TILE_PIX    (RT_TILE * RT_TILE)
#define RT_BLOCK       256           // threads per workgroup = 4 wavefronts
#define RT_MAX_LEVELS  34            // recursion levels 0..32 plus one end marker
#define RT_MAX_LDS_NODES   1024      // BVH nodes staged in LDS (32 B each)
#define RT_MAX_LDS_LEAVES  512       // leaf records staged in LDS (32 B each)
#define RT_SUB_LEAF        4         // second-level hierarchy: primitives per entry run
#define RT_SUB_MIN_GROUP   8         // groups of at most this many primitives are tested one by one
#define RT_SUB_STACK       40        // most second-level traversal stack entries per thread (LDS); see rt_sub_stack_entries
#define RT_SUB_MAX_DEPTH   24        // deepest second-level hierarchy (16M runs)

// ---- primitives ---------------------------------------------------------------------------------------
// A primitive set is one ordering of the scene's primitives: leaf order (BVH traversal) or file order
// (brute force, which must visit planes, cubes, spheres in the order of the scene file).
//   plane  : 3 words  w0 = (n.x, n.y, n.z, bits first_vert)
//                     w1 = (v0.x, v0.y, v0.z, bits n_verts | drop_axis << 16)   v0 = vertex the plane offset uses
//                     w2 = (bits gid, 0, 0, 0)
//   vertex : 1 word   (p, q, d, 0)  winding-sorted polygon vertices in the plane's 2D projection: d = the dropped
//                     coordinate, (p, q) = the other two in x, y, z order (drop x: y, z; drop y: x, z; drop z: x, y)
//   cube   : 6 words  w0 = (start.xyz, scale)  w1 = (end.xyz, 1/scale)
//                     w2 = (cos rx, sin rx, cos ry, sin ry)  w3 = (cos rz, sin rz, bits gid, 0)
//                     w4 = (cos -rz, sin -rz, cos -ry, sin -ry)  w5 = (cos -rx, sin -rx, 0, 0)
//   sphere : 2 words  w0 = (start.xyz, radius)  w1 = (end.xyz, bits gid)
// gid = index in scene-file order over planes, then cubes, then spheres; materials are indexed by it.
struct DevPrimSet {
    const float4* planes;
    const float4* verts;           // polygon vertices, projected: (kept, kept, dropped, 0); bounds are made from these
    const float4* edges;           // per vertex i, the edge from its predecessor j: (Pi.x, Pi.y, Pj.y, slope | NaN), see plane_test
    const float4* cubes;
    const float4* spheres;
    int n_planes, n_cubes, n_spheres;
};

// material: 3 words  m0 = (diffuse.rgb, shininess)  m1 = (specular.rgb, transparency)
//                    m2 = (ior, bits tex_id, reflectivity, 0)
// plane uv : 3 words per plane gid  (uAxis.xyz, uLen) (vAxis.xyz, vLen) (sorted v0.xyz, 0)
// texture  : descriptor (width, height, first texel, 0); texels are (r/255, g/255, b/255, 0)
// light    : (x, y, z, radiant intensity)
// BVH node : 2 words (min.xyz, bits skip) (max.xyz, bits leaf | -1), laid out in the order a traversal that
//            visits the right child first meets them; skip = index of the next node once this subtree is done.
// leaf     : 2 words of ints (plane_first, plane_count, cube_first, cube_count)
//            (sphere_first, sphere_count, first record of the leaf's second-level hierarchy | -1, depth of that hierarchy)
// second-level record : 8 words (16 in scenes with moving primitives), see sub_trace in rt_kernels.hip: the boxes of
//            four children plane by plane; a run's record also holds its (up to four) entries, class << 30 | index in the
//            primitive set or -1, and per entry class << 30 | the index the primitive had in the reference's testing
//            order, which decides exact ties.  The groups' primitive records are stored in hierarchy order (a run's
//            primitives are neighbours in their class arrays)
struct DevScene {
    const float4* nodes;  int n_nodes;
    const int4* leaves;   int n_leaves;
    DevPrimSet leafset;
    DevPrimSet fileset;
    DevPrimSet walkset;            // the file-order set as its second-level hierarchy reads it (records in hierarchy order); = fileset without one
    const float4* mats;
    const float4* plane_uv;
    const float4* plane_bound;     // per plane in file order: sphere around its vertices (centre, radius)
    const int4* tex_desc;
    const float4* texels;
    const float4* lights; int n_lights;
    const float4* sub_nodes;       // records of the second-level hierarchies of all groups (NULL when no group has one)
    const int* sub_entries;        // build products (entry slots, and per slot the primitive's place in the reference's testing
    const int* sub_order;          //   order, NULL: the entry itself is that); the kernels read the copies inside the run records
    int file_sub_root;             // first record of the hierarchy over the whole file-order set, or -1
    int file_sub_depth;            // depth of that hierarchy
    int sub_stride;                // words per second-level node: 4 (boxes over the whole motion) or 8 (boxes per time, see rt_sub_motion_offset)
    int sub_stack_entries;         // traversal stack entries per thread, see rt_sub_stack_entries: any hierarchy of the scene
    int sub_stack_entries_leaves;  // the same for walks that only enter the leaves' hierarchies (not the file-order set's)
    int n_verts;                   // polygon vertices of all planes
    int lds_prims;                 // tiny scene: kernels keep the primitive records and materials in LDS
    float center[3], radius;       // sphere around the true bounds of everything in the scene
    unsigned long long* walk_stats;  // development builds (RT_WALK_STATS): 32 counters behind the frame's statistics, else NULL
};

struct DevFrame {
    // camera (per-frame constants of Camera::convertPixelToRay)
    float cam_loc[3], fwd[3], right[3], up[3];
    float sw_over_fl, sh_over_fl, res_x, res_y, lens_radius, focal_distance;
    int width, height;
    // config
    int soft_shadows, ss_samples, glossy, gr_samples, anti_aliasing, aa_n, dof_on, dof_n, mb_on, mb_n;
    int textures, use_bvh, fallback_full, motion_blur, max_depth;
    float light_radius;
    uint32_t seed;
    int fast_samples;              // randomly drawn shadow rays use reciprocal arithmetic (distribution-level parity)
    int skip_unlit;                // shadow rays of (hit, light) pairs whose N.L is clearly negative are not traced
    int replicate;                 // soft shadows: the first reflection of a path is traced gr_samples times (see k_extend)
    int cull_cones;                // soft shadows of tiny scenes: (hit, light) pairs whose whole cone of sample rays misses everything are not traced
    int cull_own;                  // ... and a convex primitive (sphere, cube) is not searched for the sample rays that leave its own surface outwards
    int cull_umbra;                // ... and pairs whose every sample ray is certain to be blocked by one convex primitive (count 0) are not traced either
    int bundle;                    // ... as ONE traced ray whose node has gr_samples copies (replicate, no glossy perturbation)
    int count_stats;               // kernels add their ray counts to DevWork::stats (one atomic per workgroup and counter)
    // sharding
    int tiles_x, tiles_y, n_local_tiles;
    const int* local_tiles;        // global tile ids rendered by this rank, ascending
    // this batch
    int pix_begin, pix_count;      // range of local pixel slots (tile-major) covered by the batch
    int sample;                    // sample index s = (s_aa * dof_n + s_dof) * mb_n + s_mb
    // frames of one sample per pixel in one batch write their pixels from the level-0 shading (no sums, no finalize)
    unsigned char* direct_out; int direct_tile_major;
};

// ---- work buffers ---------------------------------------------------------------------------------------
// Hit-tree node (one per traced ray that hit), structure of arrays:
//   n_a = (hit point xyz, hit time)           n_b = (geometric normal xyz, bits gid | 1 << 31 when the node has more than one copy)
//   n_c = (bits first_link | -1, refraction weight, bits path id, bits n_reflection_children | has_refraction << 30), or, in
//         frames with bundles (DevFrame::bundle): bits n_reflection_children (13) | copies << 13 | 1 << 29 | has_refraction << 30
//   n_uv = (u, v) when texture mapping is on
//   n_pix (level 0 only) = local pixel slot.
// A node stands for `copies` identical hits of the reference (a bundle: its gr_samples identical reflection rays, and
// everything below them) that differ only in the shadow samples they draw.  What differs per copy lives in SLOTS:
//   n_col[slot] = resolved colour (r, g, b, 0)     shadow[slot * lights + light] = unblocked samples
//   slot of copy cp of the k-th node of region rg of `level` = slot_off[level] + (rg * slot_mul[level] + cp) * node_rcap[level] + k
//   (copy by copy inside a region: lanes that hold neighbouring nodes touch neighbouring slots whatever the copy)
// with slot_mul = 1 everywhere without bundles (slot == node), and gr_samples for the levels >= 1 of a frame with bundles
// (a node there has 1 or gr_samples copies; copy cp's path id is the node's for cp = 0, else mix32(id + 0x9E3779B1 * cp)).
// Every level's part of the node arena and of the ray queue is cut into `regions` (<= RT_REGIONS) regions of equal capacity, each
// filled from its start through its own counter (see ctr below):
//   node index = node_off[level] + region * node_rcap[level] + order of allocation inside the region
//   queue slot = region * link_rcap[level] + order of allocation inside the region
// Secondary-ray queue entry: q_a = (origin xyz, dir.x)  q_b = (dir.y, dir.z, bits path id, bits multiplicity | replicated << 31)
//   multiplicity = how many identical rays of the reference this one stands for (statistics); replicated = a reflection
//   above this ray was traced gr_samples times for real, so reflections below it are traced once (see k_extend)
// link[link_off[level] + queue slot] = node index hit by that ray of `level`, or -1
struct DevWork {
    float4* q_a[2]; float4* q_b[2];
    float4* n_a; float4* n_b; float4* n_c; float2* n_uv; float4* n_col; int* n_pix;
    float* shadow;
    int* link;
    const unsigned* perm;          // queue slots of the level about to be traced in the order they are traced (see k_sort_keys), or NULL = slot order
    float4* accum;                 // per local pixel slot: running colour sum over samples
    int* ctr;                      // counters, see below
    unsigned long long* stats;     // ray counters, see below
    unsigned cap_nodes;            // total node capacity = node_off[max_depth + 1]
    int node_off[RT_MAX_LEVELS];   // [level] first node of the level's region; [max_depth + 1] = end
    int link_off[RT_MAX_LEVELS];   // [level] first link of the rays traced at `level` (level >= 1); [max_depth + 1] = end
    int slot_off[RT_MAX_LEVELS];   // [level] first slot of the level's nodes, see above
    int slot_mul[RT_MAX_LEVELS];   // [level] slots per node of the level
    int node_rcap[RT_MAX_LEVELS];  // [level] nodes one region of the level can hold
    int link_rcap[RT_MAX_LEVELS];  // [level] rays one region of the level's queue can hold (level >= 1)
    int regions;                   // regions in use: 1, 2, 4 or RT_REGIONS (small batches do not need the spread)
    // Soft shadows of tiny scenes: (hit, light) pairs whose sample rays must be traced, queued by the launch that
    // classifies the pairs and traced by a second launch that deals them out evenly (see k_shadow).  One queue per
    // shadow launch of a batch (RT_SHADOWQ_LAUNCHES), each cut into `regions` parts of sq_rcap entries with their own
    // counters: entry = two words, (index into shadow[], copy, path id, hit time) and
    // (hit point, 0).  NULL / 0: pairs are traced by the workgroup that met them.
    int4* sq; int sq_rcap;
};

// ctr[] slots: allocation counters per (level, region), zeroed at the start of every batch.  Each recursion level owns
// a fixed part of the node and link arenas (offsets in DevWork, sized on the host for the worst-case fan-out), so
// no kernel ever needs another level's running count and there is no bookkeeping between launches.
// The hits recorded at `level` and the secondary rays they spawn (to be traced at level + 1) sit in the two halves
// of one aligned 64-bit word, so a wavefront allocates both with a single atomic.
// Why regions: returning atomics on one word - or on words a few hundred bytes apart, which one memory channel
// serves - complete at a fixed rate of roughly a hundred per microsecond, whatever the number of wavefronts; with one
// counter per level that rate, not the tracing, set the duration of the extend kernels.  A workgroup-round allocates
// in the region of its 256-ray chunk (chunk index mod RT_REGIONS for primary rays, the region the rays were queued
// in otherwise), and the regions' counters lie RT_CTR_REGION_INTS * 4 bytes apart.
#define RT_REGIONS 8
#define RT_CTR_REGION_INTS 4096
#define RT_CTR_NODE(level, region) ((region) * RT_CTR_REGION_INTS + 2 * (level))          // hits recorded at `level`
#define RT_CTR_LINK(level, region) ((region) * RT_CTR_REGION_INTS + 2 * (level) - 1)      // rays queued for `level` (>= 1)
#define RT_SHADOWQ_LAUNCHES 3
#define RT_CTR_SHADOWQ(k, region) ((region) * RT_CTR_REGION_INTS + 2 * RT_MAX_LEVELS + (k))  // pairs queued by shadow launch k of the batch
#define RT_CTR_WORDS_IN_USE (2 * RT_MAX_LEVELS + RT_SHADOWQ_LAUNCHES)
enum { RT_C_TOTAL = RT_REGIONS * RT_CTR_REGION_INTS };
// stats[] slots
// *_REF = what the reference would trace: without glossy perturbation it re-traces gr_samples identical reflection
// rays per hit, and with them their whole subtrees; every ray carries that multiplicity.
enum { RT_S_PRIMARY = 0, RT_S_SHADOW, RT_S_REFLECT, RT_S_REFRACT, RT_S_NODES, RT_S_SHADOW_REF, RT_S_REFLECT_REF, RT_S_REFRACT_REF,
       RT_S_ERR = 8,               // non-zero when a capacity was exceeded during the frame
       RT_S_WALK = 16,             // 32 step counters of development builds (see RT_WS in rt_kernels.hip)
       RT_S_TOTAL = 48 };

// 16-byte words the kernels copy into LDS when DevScene::lds_prims is set
#define RT_LDS_PRIM_MAX_WORDS 384
static inline int rt_lds_prim_words(const DevScene& sc) {
    return sc.n_verts + 3 * (sc.fileset.n_planes + sc.leafset.n_planes) + 6 * (sc.fileset.n_cubes + sc.leafset.n_cubes) +
           2 * (sc.fileset.n_spheres + sc.leafset.n_spheres) + 3 * (sc.fileset.n_planes + sc.fileset.n_cubes + sc.fileset.n_spheres);
}

// Stack entries a walk of a hierarchy of `depth` levels needs: a step on a record above the runs (two binary levels, or
// the step from the last level to the runs) pushes up to three siblings, a run's record pushes nothing, and there are
// ceil(depth / 2) records above the runs on a path from the root to a primitive.
static inline int rt_sub_stack_entries(int depth) { return 3 * ((depth + 1) / 2) + 2; }

// kernel launchers (rt_kernels.hip)
void rt_launch_extend(hipStream_t s, const DevScene& sc, const DevFrame& f, const DevWork& w, int level, int grid);
void rt_launch_primary_shaded(hipStream_t s, const DevScene& sc, const DevFrame& f, const DevWork& w, int grid);
// (returns the number of kernels launched: classification and tracing are two where the queue of pairs is in use)
int rt_launch_shadow(hipStream_t s, const DevScene& sc, const DevFrame& f, const DevWork& w, int lv_begin, int lv_end, int grid, int grid_drain);
void rt_launch_sort_keys(hipStream_t s, const DevScene& sc, const DevWork& w, int level, unsigned* keys, unsigned* slots);
void rt_launch_shade(hipStream_t s, const DevScene& sc, const DevFrame& f, const DevWork& w, int level, int grid);
void rt_launch_begin_frame(hipStream_t s, const DevWork& w, void* zero, size_t zero_bytes, bool zero_stats);
void rt_launch_finalize(hipStream_t s, const DevFrame& f, const DevWork& w, unsigned char* out, int tile_major, int n_slots, int samples);
void rt_launch_assemble(hipStream_t s, const unsigned char* tiles, int world, int max_tiles, int tiles_x, int tiles_y,
                        int width, int height, unsigned char* out);

#endif
