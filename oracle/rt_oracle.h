/* CPU restatement of the reference ray tracer's render path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (raytracer_amd/, include/rt_api.h) never links or calls it.
 *
 * Pinning: the compiled, unmodified reference (oracle/_ref/raytracer_ref, built by oracle/Makefile from
 * /root/reference/Code) is run on the same scenes and flag sets; this restatement reproduces its PPM
 * byte-for-byte, stochastic flag sets included (same glibc rand() call order). See tests/test_oracle_*.py.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors struct Config, config.h:7-28 (outputFile omitted). */
typedef struct orc_config {
    int softShadows;          /* config.h:8  */
    int SSsamples;            /* config.h:9  */
    int glossyReflect;        /* config.h:10 */
    int GRsamples;            /* config.h:11 */
    float lightRadius;        /* config.h:12 */
    int antiAliasing;         /* config.h:14 */
    int AAsamples;            /* config.h:15 */
    int textures;             /* config.h:17 */
    int bvh;                  /* config.h:19 */
    int reflectRefractDepth;  /* config.h:22 */
    int dof;                  /* config.h:24 */
    int dofSamples;           /* config.h:25 */
    int motionBlur;           /* config.h:26 */
    int mbSamples;            /* config.h:27 */
} orc_config;

/* Rays = top-level calls of BVHNode::intersect / unacceleratedIntersection made by the render code
 * (a secondary ray that falls back to brute force after a BVH miss counts once). */
typedef struct orc_stats {
    long long primary, shadow, reflection, refraction;
    long long box_tests, prim_tests;
    double seconds;           /* wall time of the pixel loop only (raytracer.cpp:109-193) */
} orc_stats;

void orc_default_config(orc_config* c);
/* config.cpp:7-91. argv[0] is skipped like the reference does. Unknown flags are reported on stderr and ignored.
 * out_name (may be NULL) receives the -o value. Returns 0. */
int orc_parse_args(int argc, const char* const* argv, orc_config* c, char* out_name, int out_name_cap);

/* Parse scene text + render. rgb_out receives H*W*3 ints (row-major, y down, r,g,b) exactly as
 * raytracer.cpp:173-178 hands them to Image::modifyPixel. rgb_cap = capacity of rgb_out in ints.
 * rand_seed < 0: leave the C library generator as it is (the reference never calls srand, so a fresh
 * process starts from srand(1)); >= 0: call srand(rand_seed) first.
 * Returns 0, or a negative code with a message in orc_last_error(). */
int orc_render(const char* scene_text, const char* textures_dir, const orc_config* c, int rand_seed,
               int* rgb_out, long long rgb_cap, int* width, int* height, orc_stats* stats);

/* Restrict the following orc_render calls to the pixel window [x0,x1) x [y0,y1) (others stay 0). An exact crop for
 * deterministic flag sets only. orc_set_window(0, 0, 1 << 30, 1 << 30) restores the full frame. */
void orc_set_window(int x0, int y0, int x1, int y1);
void orc_set_stride(int sx, int sy);   /* inside the window, only every sx-th column / sy-th row */

/* Known-answer helpers (same parsing + build as orc_render). */
/* Camera::convertPixelToRay, camera.cpp:109-199, with aperture 0 (no lens sample). out6 = origin, direction. */
int orc_camera_ray(const char* scene_text, float x, float y, float* out6);
/* BVHNode::buildBVH, bvh.cpp:108-172: dumps the pointer tree in pre-order (node, left subtree, right subtree).
 * Per node 12 floats are written to out: min[3], max[3], has_left, has_right, n_planes, n_cubes, n_spheres, depth.
 * Returns the number of nodes (<= cap_nodes) or a negative error. */
int orc_bvh_dump(const char* scene_text, const orc_config* c, float* out, int cap_nodes);

/* Replays one known-answer record of oracle/ref_kat.cpp (kind, inputs as that file lays them out) through the
 * restatement's primitive tests and shape boxes; returns the number of floats written to out (<= 19) or -1. */
int orc_kat(int kind, const float* in, float* out);

const char* orc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
