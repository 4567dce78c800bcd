// On-device construction of the second-level hierarchies (rt_build.hip) and of the leaf-order primitive arrays.
#ifndef RT_BUILD_H
#define RT_BUILD_H

#include "rt_types.h"

// One group of primitives that gets a hierarchy: class-wise ranges in a primitive set, and where its items,
// entries and nodes go.  A group of n items becomes a perfect binary tree with 2^depth runs of 2..RT_SUB_LEAF
// entries (depth = ceil(log2(ceil(n / RT_SUB_LEAF))) >= 1): 2^depth - 1 nodes in heap order starting at node_base, for
// which 2^depth node slots are reserved, and four entry slots per run starting at 4 * node_base.
struct BuildGroup {
    int p0, pn, c0, cn, s0, sn;
    int item_base;     // first item of the group in the build's item arrays
    int node_base;     // absolute index of the group's root in the node array
    int n, depth;
    int wide_base;     // first of the group's records in the array the walk reads (rt_build_wide_records)
};

inline int rt_build_depth(long long n) {
    long long runs = (n + RT_SUB_LEAF - 1) / RT_SUB_LEAF; int d = 0;
    while ((1ll << d) < runs) ++d;
    return d < 1 ? 1 : d;
}

// Union of the true bounds of every primitive of `set` (no sphere enlargement): out6 = min xyz, max xyz.
hipError_t rt_build_scene_bounds(hipStream_t s, const DevPrimSet& set, float out6[6]);

// Builds the hierarchies of `groups` (host array, items numbered group after group) over `set`.
// sphere_e = squared-distance rounding allowance of the sphere test; nodes / entries = device arrays (absolute node
// indices; a group's entries start at 4 * node_base).  When final6 != NULL it receives the union of the (enlarged) item boxes.
// motion = node records of 8 words (boxes at time 0 and their change until time 1) instead of 4 (boxes over the whole
// motion).  Synchronises the stream.
hipError_t rt_build_hierarchies(hipStream_t s, const DevPrimSet& set, const BuildGroup* groups, int n_groups, long long n_items,
                                double sphere_e, bool motion, float4* nodes, int* entries, float final6[6]);

// Records of the walk (sub_trace) a group of `depth` needs: (4^ceil(depth / 2) - 1) / 3 for every second level of its
// binary tree, and one per run.
inline long long rt_wide_records(int depth) { return ((1ll << (2 * ((depth + 1) >> 1))) - 1) / 3 + (1ll << depth); }

// The records the walk reads: a 4-ary view of the finished binary trees of `groups` in `nodes` (layout of
// rt_build_hierarchies, or of the host builder: 4 words per node) followed, per group, by one record per run with the
// boxes of the run's own primitives (of `set`, which the final `entries` index), the entries and their places in the
// reference's testing order (`order`, or NULL: the entries themselves).  Written to `wide` at 8 (motion: 16) words per
// record from every group's wide_base.  Synchronises the stream.
hipError_t rt_build_wide_records(hipStream_t s, const DevPrimSet& set, const BuildGroup* groups, int n_groups, bool motion, double sphere_e,
                                 const float4* nodes, const int* entries, const int* order, float4* wide);

// Records in hierarchy order.  `entries` / `order` = the slot arrays (absolute slots, a group's start at 4 * node_base):
// on return entries[slot] names the primitive's new place, class << 30 | index into out_planes / out_cubes / out_spheres
// (inside the group's own class ranges, so records outside the groups keep their indices: copy them beforehand), and
// order[slot] holds the entry as it was, i.e. the primitive's place in the reference's testing order, which decides
// exact ties.  The out arrays must not alias the set's.  Every slot of the groups must be written (-1 = unused).
// Synchronises the stream.
hipError_t rt_build_reorder(hipStream_t s, const DevPrimSet& set, const BuildGroup* groups, int n_groups, int* entries, int* order,
                            float4* out_planes, float4* out_cubes, float4* out_spheres);

// Leaf-order copies of the file-order primitive records.  leaf_desc (host): per leaf (first id in prim_ids, planes,
// cubes, spheres) and (first leaf-order plane, cube, sphere, 0).  prim_ids_dev holds validated indices.
struct LeafGather { int first_id, n_planes, n_cubes, n_spheres, lp0, lc0, ls0, pad; };
hipError_t rt_build_gather_leaves(hipStream_t s, const DevPrimSet& fileset, const int* prim_ids_dev, const LeafGather* leaves, int n_leaves,
                                  float4* leaf_planes, float4* leaf_cubes, float4* leaf_spheres);

// Radix sort of (key, value) pairs on the low `bits` bits of the keys (rocPRIM), out of place.  rt_sort_pairs_temp_bytes
// tells how much temporary device memory a sort of n pairs needs; rt_sort_pairs only enqueues work on the stream.
hipError_t rt_sort_pairs_temp_bytes(size_t n, size_t* bytes);
hipError_t rt_sort_pairs(hipStream_t s, void* temp, size_t temp_bytes, const unsigned* keys_in, unsigned* keys_out, const unsigned* vals_in, unsigned* vals_out,
                         size_t n, int bits);

#endif
