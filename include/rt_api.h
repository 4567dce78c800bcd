/* rt_api.h — C ABI of the MI355X render path.
 *
 * The reference is a single executable with no plugin/FFI surface; this boundary replaces the span between
 * its two timer reads (raytracer.cpp:109 and :192): "given the parsed scene, the built BVH and the config,
 * fill the image".  Everything on the far side of it is hand-written HIP for gfx950.
 *
 * Ownership: the caller owns every buffer it passes in; the context copies what it needs during the
 * rt_upload_* calls.  Nothing returned needs freeing except the context.
 * Errors: every call returns RT_OK (0) or a negative code and never throws or exits; rt_last_error() gives
 * the text.  Threading: calls on one context must be serialised by the caller; contexts are independent.
 */
#ifndef RT_API_H
#define RT_API_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    RT_OK = 0,
    RT_ERR_ARG = -1,        /* bad argument / inconsistent sizes */
    RT_ERR_HIP = -2,        /* a HIP runtime call failed (no device, out of memory, launch failure) */
    RT_ERR_STATE = -3,      /* call order: render before upload, ... */
    RT_ERR_CAPACITY = -4,   /* workspace too small for the requested depth / sample fan-out */
    RT_ERR_IO = -5,         /* host surface: file missing / malformed */
    RT_ERR_PARSE = -6       /* host surface: scene text the reference's scanner would throw on */
};

/* ---- scene PODs ------------------------------------------------------------------------------------ */

/* Camera fields as Camera::parseCameraDataFromJson leaves them (camera.h:14-22, camera.cpp:68-100):
 * focal_length and sensor already divided by 1000; aperture = 0 unless depth of field is enabled. */
typedef struct rt_camera {
    float location[3];
    float gaze[3];
    float aperture;
    float focal_length;
    float focal_distance;
    float sensor_width;
    float sensor_height;
    float res_x;
    float res_y;
} rt_camera;

/* pointlight.h:11-12 */
typedef struct rt_light {
    float location[3];
    float rad_intensity;
} rt_light;

/* shape.h:16-24.  tex_id = index into rt_scene.textures, or -1 when the shape's hasTex is false
 * (texture mapping off, or "texture": null; sphere.cpp:105-109). */
typedef struct rt_material {
    float diffuse[3];
    float specular[3];
    float shininess;
    float transparency;
    float ior;
    int32_t tex_id;
} rt_material;

/* sphere.h:14-26 */
typedef struct rt_sphere {
    float start[3];
    float end[3];
    float radius;
    rt_material mat;
} rt_sphere;

/* cube.h:14-30: rotation = XYZ Euler radians, scale = uniform half edge. */
typedef struct rt_cube {
    float start[3];
    float end[3];
    float rotation[3];
    float scale;
    rt_material mat;
} rt_cube;

/* plane.h:15-37: a planar polygon.  Vertices are in FILE order (the order of "corners" in scene.json, which
 * is Blender vertex order, not winding order); normal is what Plane::calculateNormal (plane.cpp:304-321)
 * computed from the first three of them.  The library derives the winding-sorted copy itself. */
typedef struct rt_plane {
    int32_t n_verts;
    int32_t first_vert;      /* index of this plane's first vertex in rt_scene.plane_verts (xyz triples) */
    float normal[3];
    rt_material mat;
} rt_plane;

/* A texture exactly as the reference's PPM reader produced it (image.cpp:13-67), i.e. for a P6 file with
 * every channel already moved down by one byte.  rgb = height*width*3 ints, row-major, top row first. */
typedef struct rt_texture {
    int32_t width;
    int32_t height;
    const int32_t* rgb;
} rt_texture;

typedef struct rt_scene {
    rt_camera camera;        /* cams[0]; the reference renders only the first camera (raytracer.cpp:107-136) */
    int32_t n_lights;   const rt_light* lights;
    int32_t n_spheres;  const rt_sphere* spheres;
    int32_t n_cubes;    const rt_cube* cubes;
    int32_t n_planes;   const rt_plane* planes;
    int32_t n_plane_verts; const float* plane_verts;   /* n_plane_verts * 3 floats */
    int32_t n_textures; const rt_texture* textures;
} rt_scene;

/* One node of the host-built pointer BVH (bvh.h:13-37), flattened in any order with the root at index 0.
 * left/right = child node index or -1.  A node with left == right == -1 is a leaf and owns
 * prim_ids[first_prim .. first_prim + n_planes + n_cubes + n_spheres): plane indices first, then cube
 * indices, then sphere indices, each in the order the reference's leaf vectors hold them (bvh.cpp:113-115). */
typedef struct rt_bvh_node {
    float box_min[3];
    float box_max[3];
    int32_t left;
    int32_t right;
    int32_t first_prim;
    int32_t n_planes;
    int32_t n_cubes;
    int32_t n_spheres;
} rt_bvh_node;

/* ---- render configuration ---------------------------------------------------------------------------- */

/* Every field of struct Config (config.h:8-27) except outputFile, in the same order, then the additions
 * this implementation needs.  Fill with rt_default_config() first. */
typedef struct rt_config {
    int32_t soft_shadows;         /* config.h:8   -ss   */
    int32_t ss_samples;           /* config.h:9   -sss  */
    int32_t glossy_reflect;       /* config.h:10  -gr   */
    int32_t gr_samples;           /* config.h:11  -grs  */
    float   light_radius;         /* config.h:12        */
    int32_t anti_aliasing;        /* config.h:14  -aa   */
    int32_t aa_samples;           /* config.h:15  -aas  */
    int32_t textures;             /* config.h:17  -t    */
    int32_t bvh;                  /* config.h:19  -u clears it */
    int32_t reflect_refract_depth;/* config.h:22  -r / -rd */
    int32_t dof;                  /* config.h:24  -dof  */
    int32_t dof_samples;          /* config.h:25  -dofs */
    int32_t motion_blur;          /* config.h:26  -m    */
    int32_t mb_samples;           /* config.h:27  -mbs  */
    /* additions (no counterpart in the reference) */
    uint32_t seed;                /* key of the counter-based generator; the reference never seeds rand() */
    int32_t fallback;             /* brute-force retry of secondary rays after a BVH miss (raytracer.cpp:424,484):
                                     0 or 1 = over every primitive, exactly as the reference (default),
                                     2 = skipped (faster, differs where the reference's retry finds a hit) */
} rt_config;

typedef struct rt_stats {
    uint64_t primary_rays;        /* camera samples traced */
    uint64_t shadow_rays;
    uint64_t reflection_rays;     /* traced; identical copies the reference would re-trace are not counted here */
    uint64_t refraction_rays;
    uint64_t nodes_shaded;        /* hits that went through Blinn-Phong */
    /* what the reference itself would trace for this frame: without -gr it re-traces gr_samples identical
       reflection rays per hit together with everything below them (raytracer.cpp:400-432) */
    uint64_t shadow_rays_ref;
    uint64_t reflection_rays_ref;
    uint64_t refraction_rays_ref;
    uint64_t batches;             /* pipeline passes (one per sample index per pixel chunk) */
    uint64_t launches_extend_primary, launches_extend_secondary, launches_shadow, launches_shade;
    double   ms_total;            /* rt_render*: enqueue -> results ready, host wall clock */
    /* device time per stage, summed over the frame, from HIP events on the render stream; filled only when
       the frame has at most 64 batches (0 otherwise) */
    double   ms_extend_primary, ms_extend_secondary, ms_shadow, ms_shade, ms_other;
} rt_stats;

typedef struct rt_ctx rt_ctx;

/* ---- calls ------------------------------------------------------------------------------------------- */

void rt_default_config(rt_config* cfg);

/* Creates a context on HIP device `device_id`.  workspace_bytes = upper bound for the ray queues and the
 * hit-tree arena (0 = default of 4 GiB); the work buffers are allocated by the first render and
 * re-allocated (never beyond that bound) when a later frame needs more: deeper recursion, more lights, a larger film.
 * A re-allocation waits for the frames in flight and drops every recorded frame graph. */
rt_ctx* rt_create(int device_id, uint64_t workspace_bytes);
void rt_destroy(rt_ctx* ctx);

/* Work is enqueued on `hip_stream` (a hipStream_t; NULL = the context's own stream). */
int rt_set_stream(rt_ctx* ctx, void* hip_stream);
/* The hipStream_t work is currently enqueued on (the context's own stream unless rt_set_stream replaced it), so that
 * a caller can order its own work against it without creating another stream: every extra stream of a process may
 * end up sharing one of the few hardware queues with a render stream. */
void* rt_get_stream(rt_ctx* ctx);

/* Implementation options (no counterpart in the reference; none of them changes a rendered value). Set before
 * rt_upload_scene.  "second_level_min_group" = N: groups of more than N primitives (a leaf of the reference's
 * depth-8 tree, or the whole scene for -u) are searched through a second-level hierarchy instead of one by
 * one; default 8, a huge value turns the hierarchy off.  "device_build" = 0/1: whether those hierarchies and the
 * leaf-order copies of the primitive records are made by kernels on the device (default 1) or on the host; set
 * before rt_upload_scene.  "motion_boxes" = 0/1 (default 1, before rt_upload_scene): device-built hierarchies of a scene in
 * which a sphere or cube moves keep boxes per time (at time 0 and their change until time 1, 128 bytes per node)
 * instead of boxes over the whole motion (64 bytes per node).  "graphs" = 0/1: whether rt_render_device (without
 * statistics) records a frame's launch sequence into a HIP graph on its second call with the same configuration,
 * output pointer and stream, and replays it afterwards; default 1.  "overlap" = 0/1: whether frames rendered
 * without statistics trace the shadow rays of the primary hits on a second, internal stream while the render
 * stream traces the deeper recursion levels (joined before shading; the caller sees one stream); default 1.
 * "skip_unlit" = 0/1: whether the shadow rays of a (hit, light) pair whose surface faces away from the light are left
 * untraced (default 1; raytracer.cpp:322-336 multiplies their result by max(0, N.L) = 0, so frames are identical;
 * rt_stats.shadow_rays counts traced rays, shadow_rays_ref what the reference traces).
 * "cull_cones" = 0/1 (soft shadows, scenes of at most 8 primitives): whether the sample rays of a (hit, light) pair are left
 * untraced when the whole cone from the hit to the light ball provably reaches no primitive (default 1; every sample is then
 * unblocked whatever the random numbers, so frames are identical; rt_stats.shadow_rays counts traced rays).
 * "cull_own" = 0/1 (with cull_cones): whether a sphere or cube is left out of that search for sample rays that all leave
 * its own surface outwards (a convex primitive cannot be met again; default 1, frames identical).
 * "cull_umbra" = 0/1 (with cull_cones): whether the sample rays of a pair are also left untraced when the whole cone passes
 * well inside one sphere or cube at rest that lies clearly before the light ball and that the search is certain to
 * get to (default 1; every sample is then blocked whatever the random numbers, raytracer.cpp:564, so frames are identical).
 * "shadow_queue" = 0/1 (with cull_cones): whether the pairs that do need tracing are handed, through a queue in device
 * memory, to a launch of their own that deals them out evenly, instead of being traced by the workgroup that met them
 * (default 1; "grid_shadow_drain" = that launch's workgroups).  "shadow_streams" = 1/2 (with overlap): how many recursion
 * levels' shadow rays run on internal streams of their own, level 0 or levels 0 and 1 (default 2).
 * "hierarchy_order" = 0/1 (before rt_upload_scene): whether the records of groups with a second-level hierarchy are
 * stored in hierarchy order, a run's primitives side by side (default 1; the reference's testing order, which decides
 * exact ties, is kept per entry).  "sort_rays" = 0/1 and "sort_rays_min_prims" = N: whether the secondary rays of scenes
 * with second-level hierarchies and at least N primitives (default 2 000 000) are sorted by origin cell and direction
 * before they are traced (default 1; only the order of tracing changes; such frames are not recorded into graphs).
 * "fuse_simple" = 0/1: whether frames without recursion and with hard shadows run as one kernel (trace, shadow rays,
 * shading and the pixel sum in the lane of the primary ray) instead of the extend / shadow / shade sequence; default 1,
 * frames and statistics identical.
 * "count_rays" = 0/1: whether rt_render_device with a statistics pointer counts rays (default 1) or only times the
 * stages, with the kernels plain frames use (the ray counts then read 0).
 * "max_grid" (all kernels) or "grid_primary" / "grid_secondary" / "grid_shadow" / "grid_shadow_drain" / "grid_shade" = workgroups per launch,
 * 0 = chosen per launch (default).  The one switch that does move rendered values, and only those
 * of soft-shadow renders, within their sampling noise: "fast_samples" = 0/1, whether the randomly drawn soft-shadow
 * sample rays are tested with hardware reciprocals (1 ulp) instead of IEEE division; default 1, 0 = the reference's
 * arithmetic.  Deterministic renders never use it.  "replicate_reflections" = 0/1 (soft-shadow renders without -gr only):
 * the reference traces gr_samples identical reflection rays per hit and each of their hits draws its own shadow samples
 * (raytracer.cpp:400-432 over :330-332); 1 (default) traces the first reflection of every path gr_samples times so that
 * a reflected penumbra has the reference's distribution, 0 traces it once and re-uses the colour (same mean, gr_samples
 * times the variance in reflected penumbrae, far fewer rays). */
int rt_set_option(rt_ctx* ctx, const char* name, int64_t value);

/* Copies the scene to the device (replaces raytracer.cpp:55-59 as the producer of the render loop's inputs). */
int rt_upload_scene(rt_ctx* ctx, const rt_scene* scene);

/* Copies the host-built BVH (the result of BVHNode::buildBVH, bvh.cpp:108-172) and re-lays it out for the
 * device.  n_nodes == 0 or nodes == NULL: no BVH (only -u renders are possible). Must follow rt_upload_scene. */
int rt_upload_bvh(rt_ctx* ctx, const rt_bvh_node* nodes, int32_t n_nodes, const int32_t* prim_ids, int32_t n_prim_ids);

/* Image sharding for one-process-per-GPU runs: this context renders the 32x32-pixel tiles t with
 * (tile_x + tile_y) % world == rank.  Default rank 0 of world 1 = whole image. */
int rt_set_shard(rt_ctx* ctx, int32_t rank, int32_t world);
/* Number of tiles this context renders / the largest count over all ranks (gather buffers are padded to it). */
int32_t rt_local_tiles(const rt_ctx* ctx);
int32_t rt_max_tiles_per_rank(const rt_ctx* ctx);

/* The render loop (raytracer.cpp:109-193).  world == 1: rgb_out receives res_y*res_x*3 bytes, row-major,
 * y down, rgb[y][x][c] = (int)(255*colour) exactly as raytracer.cpp:173-175.  world > 1: rgb_out receives this
 * rank's tiles, rt_max_tiles_per_rank()*32*32*3 bytes, tile-major (unused tail zeroed).
 * rt_render copies the result to host memory and waits; rt_render_device leaves it in device memory
 * (rgb_out_dev is a device pointer) and only enqueues, unless stats != NULL, in which case it also waits. */
int rt_render(rt_ctx* ctx, const rt_config* cfg, uint8_t* rgb_out, rt_stats* stats);
/* Page-locked host memory for rt_render's rgb_out (no counterpart in the reference, whose image lives in a
 * std::vector, image.h): into such a buffer the frame is copied by one direct DMA transfer, into ordinary memory it
 * goes through the driver's staging copies (6 MB of a 1080p frame: about 0.15 ms against about 0.6 ms).  NULL when the
 * allocation fails.  Free with rt_free_image. */
void* rt_alloc_image(uint64_t bytes);
void rt_free_image(void* p);
int rt_render_device(rt_ctx* ctx, const rt_config* cfg, void* rgb_out_dev, rt_stats* stats);

/* rt_render_device without statistics for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same inputs, same output
 * layout and the same kernels as rt_render_device, but every work buffer of the frame (ray queues, hit tree, shadow
 * counts, per-pixel sums, counters) is carved from one array that the library keeps in each GPU's memory for the life
 * of the process (rt_noalloc_capacity() bytes, 64 MiB), and the call performs no hipMalloc / hipFree, creates no
 * event and instantiates no graph: it only enqueues on the context's stream.  The film is rendered in as many batches
 * as that array allows; RT_ERR_CAPACITY when it cannot hold the film's per-pixel sums plus one wavefront of samples
 * (at most about 3.9 million pixel slots of 16 bytes).  All contexts of one GPU share the array and its previous
 * contents are overwritten, so the array has one owner at a time: the context that calls this function owns it until
 * its next rt_synchronize (or rt_destroy); frames of the owner follow each other on its stream, and any OTHER context
 * of that GPU that calls meanwhile gets RT_ERR_STATE and enqueues nothing.  (Changing the owner's stream between two
 * such frames without a synchronisation in between is the caller's race.)  The scene must have been uploaded before
 * (the uploads do allocate). */
int rt_render_device_noalloc(rt_ctx* ctx, const rt_config* cfg, void* rgb_out_dev);
uint64_t rt_noalloc_capacity(void);

/* Rank 0 after the gather: tiles_dev = world * rt_max_tiles_per_rank()*32*32*3 bytes (rank-major, each block
 * as rt_render_device wrote it); rgb_out_dev = res_y*res_x*3 bytes row-major.  Enqueue only.  world == 1: the one
 * block is the row-major image already and is copied (nothing happens when both pointers are equal). */
int rt_assemble_tiles(rt_ctx* ctx, const void* tiles_dev, int32_t world, void* rgb_out_dev);

/* Blocks until everything enqueued on the context's stream has finished.  RT_ERR_CAPACITY if the most recent frame found a
 * work buffer full (its image is then incomplete; the capacities are worst cases, so this indicates a defect). */
int rt_synchronize(rt_ctx* ctx);

const char* rt_last_error(const rt_ctx* ctx);   /* ctx may be NULL: error of the last failed rt_create */

/* ---- a device-resident scene handed from one context to others (no counterpart in the reference) ------------------
 * After its uploads a context holds every array the kernels read in its GPU's memory.  Another context - on another GPU
 * of the node, in this process or in another rank's - takes the scene over without parsing, building or uploading:
 *   source:    rt_export_scene_layout(src, buf, rt_scene_layout_bytes())     a small POD blob (no addresses in it)
 *   receivers: rt_adopt_scene_layout(dst, buf, bytes)                        allocates the arrays, contents undefined
 *   both:      rt_scene_array(ctx, k, &ptr, &bytes) for k < rt_scene_array_count(ctx): the k-th array on either side;
 *              the caller moves the bytes device to device (ncclBroadcast / torch.distributed.broadcast / hipMemcpy)
 *              and synchronises before the first render.
 * rt_adopt_scene_from does all of it between two contexts of one process with device copies.  An adopted scene
 * renders exactly the source's frames; rt_upload_bvh on it fails with RT_ERR_STATE (upload the tree on the source and
 * adopt again), rt_upload_scene replaces it.  The layout is only valid between builds of the same library. */
uint64_t rt_scene_layout_bytes(void);
int rt_export_scene_layout(rt_ctx* src, void* layout, uint64_t capacity);
int rt_adopt_scene_layout(rt_ctx* dst, const void* layout, uint64_t bytes);
int rt_scene_layout_resolution(const void* layout, uint64_t bytes, int32_t* width, int32_t* height);   /* film size the layout carries */
int32_t rt_scene_array_count(const rt_ctx* ctx);
int rt_scene_array(rt_ctx* ctx, int32_t k, void** dev_ptr, uint64_t* bytes);
int rt_adopt_scene_from(rt_ctx* dst, rt_ctx* src);
/* Several contexts of ONE GPU that render the same scene side by side (frames in flight): dst uses src's device arrays
 * in place - nothing is allocated or copied, and the caches hold one copy of the scene instead of one per context.
 * src owns the arrays: it must stay alive, and keep its scene, for as long as dst renders from them (rt_upload_scene
 * or an adoption on dst ends the sharing).  No counterpart in the reference (one scene, one thread). */
int rt_share_scene_from(rt_ctx* dst, rt_ctx* src);

/* ---- several GPUs of one node, one process (no counterpart in the reference, which is single-threaded) ------------
 * The render loop of raytracer.cpp:109-193 split over n devices: device k renders the 32x32 tiles with
 * (tile_x + tile_y) % n == k, the tile blocks are gathered to the first device of the list with RCCL (one group of
 * ncclRecv / ncclSend over xGMI per frame; librccl is loaded at run time, only by rt_multi_create) and re-ordered there
 * into the row-major frame.  The frame is bit-identical to the single-device one for any n.  A device id may appear more
 * than once (rehearsal on a machine with fewer GPUs): RCCL cannot span duplicates, the blocks then travel by
 * hipMemcpyAsync.  Calls on one rt_multi must be serialised by the caller. */
typedef struct rt_multi rt_multi;
rt_multi* rt_multi_create(int32_t n_devices, const int32_t* device_ids, uint64_t workspace_bytes_per_device);
void rt_multi_destroy(rt_multi* m);
int rt_multi_set_option(rt_multi* m, const char* name, int64_t value);             /* rt_set_option on every context; "device_broadcast" is rt_multi's own */
/* The scene and the tree of raytracer.cpp:55-79, once per frame set (nodes == NULL: no BVH): rt_upload_scene +
 * rt_upload_bvh on the first device, whose finished device arrays are then broadcast to the other devices (the layout
 * calls below + one ncclBroadcast per array over xGMI; device copies for a repeated device id).  With the option
 * "device_broadcast" set to 0 every device uploads and builds for itself instead, one host thread per device. */
int rt_multi_upload(rt_multi* m, const rt_scene* scene, const rt_bvh_node* nodes, int32_t n_nodes, const int32_t* prim_ids, int32_t n_prim_ids);
/* One frame: rgb_out (host) receives res_y*res_x*3 bytes as rt_render writes them.  rt_multi_render_device only
 * enqueues; the frame is in rt_multi_image() (memory of the first device) after rt_multi_synchronize(). */
int rt_multi_render(rt_multi* m, const rt_config* cfg, uint8_t* rgb_out);
int rt_multi_render_device(rt_multi* m, const rt_config* cfg);
int rt_multi_synchronize(rt_multi* m);
void* rt_multi_image(rt_multi* m);
rt_ctx* rt_multi_context(rt_multi* m, int32_t rank);        /* the context of device `rank` of the list (statistics, options) */
int32_t rt_multi_uses_rccl(const rt_multi* m);               /* 1 = blocks gathered with RCCL, 0 = device copies (duplicate ids) */
const char* rt_multi_last_error(const rt_multi* m);          /* m may be NULL: error of the last failed rt_multi_create */

#ifdef __cplusplus
}
#endif
#endif
