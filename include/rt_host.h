/* rt_host.h — the host surface kept from the reference, above the render boundary of rt_api.h:
 * command-line flags (config.cpp), scene.json loading (camera.cpp, pointlight.cpp, sphere.cpp, cube.cpp,
 * plane.cpp, shape.cpp), the host BVH build (bvh.cpp, aabb.cpp and the shapes' getAABB) and PPM I/O
 * (image.cpp).  Plain C ABI so that any host language can bind it; none of it touches the GPU.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include "rt_api.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Config::parseArgs, config.cpp:7-91: same flags, same defaults, unknown flags reported on stderr and
 * skipped; argv[0] is the program name.  out_name receives the -o value (default "output.ppm").
 * Extension flags (not in the reference): --seed N, --fallback N, --scene PATH, --textures DIR, --outdir DIR
 * are consumed only when `extensions` is non-zero; with extensions == 0 they are "Unknown flag"s too. */
int rt_host_parse_args(int argc, const char* const* argv, int extensions, rt_config* cfg, char* out_name, int out_name_cap);

/* A loaded scene: `scene` points into storage owned by the handle. */
typedef struct rt_host_scene {
    rt_scene scene;
    int32_t n_cameras;            /* how many cameras the file holds (only the first is rendered) */
    void* storage;                /* opaque */
} rt_host_scene;

/* The five parse...FromJson calls of raytracer.cpp:55-59 on `text` (the whole scene.json).  Textures named by
 * materials are read from textures_dir when cfg->textures is set (image.cpp:15 uses ../Textures/).
 * On failure returns NULL and writes a message to err (may be NULL). */
rt_host_scene* rt_host_parse_scene(const char* text, const char* textures_dir, const rt_config* cfg, char* err, int err_cap);
rt_host_scene* rt_host_load_scene(const char* json_path, const char* textures_dir, const rt_config* cfg, char* err, int err_cap);
void rt_host_free_scene(rt_host_scene* s);

/* BVHNode::buildBVH, bvh.cpp:108-172 (fixed depth 8, midpoint split on the widest axis), with the shape
 * boxes of sphere.cpp:177-188, cube.cpp:354-386 and plane.cpp:324-340 reproduced bit for bit.  The pointer
 * tree is returned flattened in pre-order (node, left subtree, right subtree), root first. */
typedef struct rt_host_bvh {
    rt_bvh_node* nodes;  int32_t n_nodes;
    int32_t* prim_ids;   int32_t n_prim_ids;
} rt_host_bvh;
int rt_host_build_bvh(const rt_scene* scene, int32_t max_depth, rt_host_bvh* out);
void rt_host_free_bvh(rt_host_bvh* b);

/* Image::readImage, image.cpp:13-67 (P3 and P6; the P6 path keeps the reference's one-byte channel shift).
 * *rgb is malloc'ed (height*width*3 ints); free with rt_host_free. */
int rt_host_read_ppm(const char* path, int32_t* width, int32_t* height, int32_t* maxval, int32_t** rgb);
/* Image::writeImage, image.cpp:83-99: "P3\nW H\nmax\n", then per row "r g b " per pixel and "\n". */
int rt_host_write_ppm(const char* path, int32_t width, int32_t height, int32_t maxval, const uint8_t* rgb);
/* No counterpart in the reference (image.cpp:83-99 only writes text): the same image as binary "P6\nW H\n255\n" + bytes,
 * 3 bytes per pixel instead of ~11 characters - for 4K frames, where the text form is ~100 MB.  Standard P6: the
 * reference's own reader shifts P6 channels by one byte (image.cpp:44-60) and is not the intended consumer. */
int rt_host_write_ppm_binary(const char* path, int32_t width, int32_t height, const uint8_t* rgb);
void rt_host_free(void* p);

/* Camera::convertPixelToRay with aperture 0 (camera.cpp:109-178): out6 = origin xyz, direction xyz. */
void rt_host_camera_ray(const rt_camera* cam, float x, float y, float* out6);

#ifdef __cplusplus
}
#endif
#endif
