/* CPU restatement of the reference ray tracer's render path (plain C, single thread, same recursion).
 *
 * TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * The product never links or calls anything in this file.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference/Code).
 * Arithmetic notes that matter for bit-parity with the g++-compiled reference on x86-64:
 *   - float expressions stay float (SSE, no excess precision, no FMA contraction);
 *   - C++ std::pow(float,int) promotes to double pow(double,double); std::sqrt of that stays double;
 *   - std::max(a,b) is (a<b)?b:a and std::min(a,b) is (b<a)?b:a  (NaN behaviour differs from fmaxf/fminf);
 *   - rand() is never seeded by the reference and is consumed in a fixed order which this file reproduces.
 * Pinned against the compiled reference itself: see rt_oracle.h.
 */
#define _GNU_SOURCE
#include "rt_oracle.h"

#include <ctype.h>
#include <errno.h>
#include <float.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static char g_err[512];
static void set_err(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
}
const char* orc_last_error(void) { return g_err; }

/* std::max / std::min exactly as libstdc++ defines them. */
static inline float smax(float a, float b) { return (a < b) ? b : a; }
static inline float smin(float a, float b) { return (b < a) ? b : a; }

/* ------------------------------------------------------------------------------------------------
 * Config (config.h:7-28, config.cpp:7-91)
 * ---------------------------------------------------------------------------------------------- */
void orc_default_config(orc_config* c) {
    c->softShadows = 0; c->SSsamples = 4; c->glossyReflect = 0; c->GRsamples = 16; c->lightRadius = 0.25f;
    c->antiAliasing = 0; c->AAsamples = 4; c->textures = 0; c->bvh = 1; c->reflectRefractDepth = 0;
    c->dof = 0; c->dofSamples = 16; c->motionBlur = 0; c->mbSamples = 16;
}

/* std::stoi as config.cpp:19,27,35,47,63,71 uses it: optional white space and sign, then digits; anything after the digits
 * is ignored; no digit at all (or a value outside int) throws, and the reference, which catches nothing, aborts.
 * Returns 0 and stores the value, or -1 for what would throw. */
static int stoi_like(const char* s, int* out) {
    char* end = NULL;
    errno = 0;
    long v = strtol(s, &end, 10);
    if (end == s || errno == ERANGE || v > 2147483647L || v < -2147483647L - 1) return -1;
    *out = (int)v;
    return 0;
}

/* Returns 0, or -1 where the reference would abort on an uncaught std::invalid_argument / std::out_of_range. */
int orc_parse_args(int argc, const char* const* argv, orc_config* c, char* out_name, int out_name_cap) {
    int bad = 0;
    orc_default_config(c);
    if (out_name && out_name_cap > 0) snprintf(out_name, (size_t)out_name_cap, "output.ppm");
#define IS(a, b) (strcmp(arg, a) == 0 || strcmp(arg, b) == 0)
    for (int i = 1; i < argc; ++i) {
        const char* arg = argv[i];
        if (IS("-ss", "--soft_shadows")) c->softShadows = 1;
        else if (IS("-sss", "--ss_samples") && i + 1 < argc) bad |= stoi_like(argv[++i], &c->SSsamples);
        else if (IS("-gr", "--glossy_reflect")) c->glossyReflect = 1;
        else if (IS("-grs", "--gr_samples") && i + 1 < argc) bad |= stoi_like(argv[++i], &c->GRsamples);
        else if (IS("-aa", "--antialiasing")) c->antiAliasing = 1;
        else if (IS("-aas", "--aa_samples") && i + 1 < argc) bad |= stoi_like(argv[++i], &c->AAsamples);
        else if (IS("-u", "--unaccelerated")) c->bvh = 0;
        else if (IS("-r", "--reflections")) c->reflectRefractDepth = 1;
        else if (IS("-rd", "--reflect_depth") && i + 1 < argc) bad |= stoi_like(argv[++i], &c->reflectRefractDepth);
        else if (IS("-t", "--texture_mapping")) c->textures = 1;
        else if (IS("-dof", "--depthoffield")) c->dof = 1;
        else if (IS("-dofs", "--dof_samples") && i + 1 < argc) bad |= stoi_like(argv[++i], &c->dofSamples);
        else if (IS("-m", "--motion-blur")) c->motionBlur = 1;
        else if (IS("-mbs", "--mb_samples") && i + 1 < argc) bad |= stoi_like(argv[++i], &c->mbSamples);
        else if (IS("-o", "--output") && i + 1 < argc) {
            ++i;
            if (out_name && out_name_cap > 0) snprintf(out_name, (size_t)out_name_cap, "%s", argv[i]);
        } else fprintf(stderr, "Unknown flag: %s\n", arg);
    }
#undef IS
    return bad ? -1 : 0;
}

/* ------------------------------------------------------------------------------------------------
 * Scene types (shape.h:14-24, sphere.h, cube.h, plane.h, camera.h:14-22, pointlight.h:11-12)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { int w, h, maxv; int* px; char name[256]; } Tex;                 /* image.h:14-18 */
typedef struct { float diffuse[3], specular[3], shininess, transparency, ior; int hasTex; int tex; } Mat;
typedef struct { float start[3], end[3], radius; Mat m; } Sphere;
typedef struct { float start[3], end[3], rot[3], scale; Mat m; } Cube;
typedef struct {
    int n;
    float (*vfile)[3];    /* vertices in file order: what every BVH-leaf copy starts from (bvh.cpp:70) */
    float (*vsorted)[3];  /* the same vertices after Plane::sortVerticesWinding (plane.cpp:254-301) */
    int main_sorted;      /* has the object in main()'s vector been sorted yet? (plane.cpp:158 mutates it) */
    float normal[3];
    Mat m;
} Plane;
typedef struct { float loc[3], gaze[3], aperture, focal_length, focal_distance, sw, sh, res_x, res_y; } Camera;
typedef struct { float loc[3], intensity; } Light;

typedef struct Node {
    float mn[3], mx[3];
    struct Node *left, *right;
    int *planes, np, *cubes, nc, *spheres, ns;
} Node;

typedef struct {
    Camera* cams; int ncams;
    Light* lights; int nlights;
    Sphere* spheres; int nspheres;
    Cube* cubes; int ncubes;
    Plane* planes; int nplanes;
    Tex* tex; int ntex;
    Node* bvh;
} Scene;

/* hit_struct.h:10-25. `time` starts at 0 here; the reference leaves it uninitialised for plane hits of a
 * ray with time == 0 (plane.cpp:194-196). */
typedef struct { float p[3], t, n[3]; const Mat* m; float time, u, v; int hasTex; } Hit;
/* ray.h:8-14: only hs.back() is ever read (raytracer.cpp:150,364,564), so keep the last push and a count. */
typedef struct { float o[3], d[3], time; Hit last; int nh; } Ray;

static orc_stats* g_stats;
/* Optional pixel window (x0, y0, x1, y1; x1/y1 exclusive): pixels outside it are skipped and left 0.  For
 * deterministic flag sets a pixel's value does not depend on the others, so a window is an exact crop. */
static int g_win[4] = { 0, 0, 1 << 30, 1 << 30 };
void orc_set_window(int x0, int y0, int x1, int y1) { g_win[0] = x0; g_win[1] = y0; g_win[2] = x1; g_win[3] = y1; }
/* Inside the window only every sx-th column and sy-th row (from the window's corner) is rendered: a sparse sample of a
 * frame that is out of reach as a whole.  Exact for deterministic flag sets (no pixel depends on another). */
static int g_stride[2] = { 1, 1 };
void orc_set_stride(int sx, int sy) { g_stride[0] = sx > 0 ? sx : 1; g_stride[1] = sy > 0 ? sy : 1; }

/* ------------------------------------------------------------------------------------------------
 * Vector helpers (raytracer.cpp:577-631, 671-673)
 * ---------------------------------------------------------------------------------------------- */
static inline void normalise(float v[3]) {                    /* raytracer.cpp:577-585 */
    float len = (float)sqrt(pow((double)v[0], 2.0) + pow((double)v[1], 2.0) + pow((double)v[2], 2.0));
    v[0] /= len; v[1] /= len; v[2] /= len;
}
static inline float dotp(const float a[3], const float b[3]) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }
static inline void cross(const float a[3], const float b[3], float r[3]) {   /* raytracer.cpp:595-602 */
    r[0] = a[1]*b[2] - a[2]*b[1]; r[1] = a[2]*b[0] - a[0]*b[2]; r[2] = a[0]*b[1] - a[1]*b[0];
}
static inline float lerpf(float a, float b, float t) { return a + t * (b - a); }   /* raytracer.cpp:671 */

/* ------------------------------------------------------------------------------------------------
 * "JSON" scanning (camera.cpp:205-284, pointlight.cpp:80-159, shape.cpp:14-135): substring search.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { const char* p; size_t n; } Str;

static long sfind(Str s, const char* key, size_t from) {
    size_t k = strlen(key);
    if (k == 0 || s.n < k) return -1;
    for (size_t i = from; i + k <= s.n; ++i)
        if (s.p[i] == key[0] && memcmp(s.p + i, key, k) == 0) return (long)i;
    return -1;
}
static long sfindc(Str s, char c, size_t from) {
    for (size_t i = from; i < s.n; ++i) if (s.p[i] == c) return (long)i;
    return -1;
}
static inline char sat(Str s, size_t i) { return i < s.n ? s.p[i] : '\0'; }   /* std::string[size()] == '\0' */

/* getFloat: camera.cpp:205-227 / shape.cpp:14-37 (allow_e=1), pointlight.cpp:80-102 (allow_e=0). */
static int get_float(Str s, const char* key, int allow_e, float* out) {
    long pos = sfind(s, key, 0);
    if (pos < 0) { set_err("Key not found in JSON string: %s", key); return -1; }
    long st = sfindc(s, ':', (size_t)pos);
    if (st < 0) { set_err("Invalid JSON string"); return -1; }
    size_t start = (size_t)st + 1;
    while (start < s.n && isspace((unsigned char)s.p[start])) start++;
    size_t end = start;
    for (;;) {
        char ch = sat(s, end);
        int ok = (end < s.n && isdigit((unsigned char)ch)) || ch == '.' || ch == '-' || (allow_e && ch == 'e');
        if (!ok) break;
        end++;
    }
    char buf[128];
    size_t len = end - start; if (len >= sizeof buf) len = sizeof buf - 1;
    memcpy(buf, s.p + start, len); buf[len] = 0;
    char* endp = NULL; errno = 0;
    float v = strtof(buf, &endp);                   /* std::stof -> strtof */
    if (endp == buf) { set_err("stof: no conversion for key %s", key); return -1; }
    if (errno == ERANGE) { set_err("stof: out of range for key %s", key); return -1; }
    *out = v; return 0;
}

/* getJSONObject: camera.cpp:230-256, shape.cpp:77-108. The brace counter starts at 1 AND counts the opening
 * brace again, so the slice runs on to the close of the ENCLOSING object; later lookups rely on first match. */
static int get_object(Str s, const char* key, Str* out) {
    long pos = sfind(s, key, 0);
    if (pos < 0) { set_err("Key not found in JSON string: %s", key); return -1; }
    long st = sfindc(s, '{', (size_t)pos);
    if (st < 0) { set_err("Invalid JSON string"); return -1; }
    size_t start = (size_t)st, end = start; int braces = 1;
    while (end < s.n) {
        if (s.p[end] == '{') braces++;
        else if (s.p[end] == '}') { braces--; if (braces == 0) break; }
        end++;
    }
    size_t len = end - start + 1; if (start + len > s.n) len = s.n - start;
    out->p = s.p + start; out->n = len; return 0;
}

/* getJSONArray: camera.cpp:260-284, shape.cpp:111-135. Includes the outer brackets. */
static int get_array(Str s, const char* key, Str* out) {
    long pos = sfind(s, key, 0);
    if (pos < 0) { set_err("Key not found in JSON string: %s", key); return -1; }
    long st = sfindc(s, '[', (size_t)pos);
    if (st < 0) { set_err("Invalid JSON string: '[' not found"); return -1; }
    size_t start = (size_t)st, end = start + 1; int br = 1;
    while (end < s.n && br > 0) { if (s.p[end] == '[') br++; else if (s.p[end] == ']') br--; end++; }
    if (br != 0) { set_err("Mismatched brackets in JSON string"); return -1; }
    out->p = s.p + start; out->n = end - start; return 0;
}

/* getString: shape.cpp:40-74. */
static int get_string(Str s, const char* key, char* out, size_t cap) {
    out[0] = 0;
    long pos = sfind(s, key, 0);
    if (pos < 0) { set_err("Key not found in JSON string: %s", key); return -1; }
    long st = sfindc(s, ':', (size_t)pos);
    if (st < 0) { set_err("Invalid JSON string"); return -1; }
    size_t start = (size_t)st + 1;
    while (start < s.n && (s.p[start] == ' ' || s.p[start] == '\t' || s.p[start] == '\n' || s.p[start] == '\r')) start++;
    if (start >= s.n) return 0;
    if (s.n - start >= 4 && memcmp(s.p + start, "null", 4) == 0) return 0;
    if (s.p[start] == '"') {
        long e = sfindc(s, '"', start + 1);
        if (e < 0) { set_err("Unterminated string value in JSON"); return -1; }
        size_t len = (size_t)e - start - 1; if (len >= cap) len = cap - 1;
        memcpy(out, s.p + start + 1, len); out[len] = 0;
    }
    return 0;
}

/* Strip outer brackets and split on top-level braces (camera.cpp:33-52 and the four copies of it). */
static int split_objects(Str arr, Str** out) {
    if (arr.n >= 2 && arr.p[0] == '[' && arr.p[arr.n - 1] == ']') { arr.p++; arr.n -= 2; }
    int cap = 16, n = 0; Str* v = (Str*)malloc(sizeof(Str) * (size_t)cap);
    int braces = 0; size_t start = 0;
    for (size_t i = 0; i < arr.n; ++i) {
        if (arr.p[i] == '{') { if (braces == 0) start = i; braces++; }
        else if (arr.p[i] == '}') {
            braces--;
            if (braces == 0) {
                if (n == cap) { cap *= 2; v = (Str*)realloc(v, sizeof(Str) * (size_t)cap); }
                v[n].p = arr.p + start; v[n].n = i - start + 1; n++;
            }
        }
    }
    *out = v; return n;
}

static int get_xyz(Str s, const char* key, const char* a, const char* b, const char* c, int allow_e, float out[3]) {
    Str o;
    if (get_object(s, key, &o)) return -1;
    if (get_float(o, a, allow_e, &out[0]) || get_float(o, b, allow_e, &out[1]) || get_float(o, c, allow_e, &out[2])) return -1;
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Textures (image.cpp:13-67). P6 reader keeps the reference's one-byte shift: the single whitespace after
 * maxval is not consumed, so it becomes the first red value and every channel moves down by one.
 * ---------------------------------------------------------------------------------------------- */
static int read_ppm(const char* dir, const char* name, Tex* t) {
    char path[1024]; snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE* f = fopen(path, "rb");
    if (!f) { set_err("Failed to open the image file %s", path); return -1; }
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    unsigned char* buf = (unsigned char*)malloc((size_t)sz + 1);
    if (fread(buf, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(buf); set_err("short read %s", path); return -1; }
    fclose(f); buf[sz] = 0;
    long i = 0;
#define SKIPWS while (i < sz && isspace(buf[i])) i++
    SKIPWS; long h0 = i; while (i < sz && !isspace(buf[i])) i++;
    int p6;
    if (i - h0 == 2 && buf[h0] == 'P' && buf[h0 + 1] == '3') p6 = 0;
    else if (i - h0 == 2 && buf[h0] == 'P' && buf[h0 + 1] == '6') p6 = 1;
    else { free(buf); set_err("Only P3 and P6 .ppm files are supported"); return -1; }
    int hdr[3];
    for (int k = 0; k < 3; ++k) {
        SKIPWS; long v = 0; int any = 0;
        while (i < sz && isdigit(buf[i])) { v = v * 10 + (buf[i] - '0'); i++; any = 1; }
        if (!any) { free(buf); set_err("bad ppm header"); return -1; }
        hdr[k] = (int)v;
    }
    t->w = hdr[0]; t->h = hdr[1]; t->maxv = hdr[2];
    size_t n = (size_t)t->w * (size_t)t->h * 3;
    t->px = (int*)malloc(sizeof(int) * n);
    if (!p6) {
        for (size_t k = 0; k < n; ++k) {
            SKIPWS; long v = 0; int any = 0, neg = 0;
            if (i < sz && (buf[i] == '-' || buf[i] == '+')) { neg = buf[i] == '-'; i++; }
            while (i < sz && isdigit(buf[i])) { v = v * 10 + (buf[i] - '0'); i++; any = 1; }
            if (!any) { free(buf); free(t->px); set_err("Invalid pixel values"); return -1; }
            t->px[k] = (int)(neg ? -v : v);
        }
    } else {
        /* stream sits right after the maxval digits: byte i is the separator and is read as data */
        if ((size_t)(sz - i) < n) { free(buf); free(t->px); set_err("Invalid pixel values"); return -1; }
        for (size_t k = 0; k < n; ++k) t->px[k] = buf[i + (long)k];
    }
#undef SKIPWS
    free(buf);
    snprintf(t->name, sizeof t->name, "%s", name);
    return 0;
}

static int tex_lookup(Scene* sc, const char* dir, const char* name) {   /* raytracer.cpp:273-285 */
    for (int i = 0; i < sc->ntex; ++i) if (strcmp(sc->tex[i].name, name) == 0) return i;
    sc->tex = (Tex*)realloc(sc->tex, sizeof(Tex) * (size_t)(sc->ntex + 1));
    if (read_ppm(dir ? dir : ".", name, &sc->tex[sc->ntex])) return -1;
    return sc->ntex++;
}

/* ------------------------------------------------------------------------------------------------
 * Parsing (camera.cpp:16-105, pointlight.cpp:11-76, sphere.cpp:19-115, cube.cpp:20-122, plane.cpp:20-140)
 * ---------------------------------------------------------------------------------------------- */
static int parse_material(Scene* sc, Str obj, const orc_config* cfg, const char* texdir, Mat* m) {
    Str ms, ds, ss;
    if (get_object(obj, "\"material\"", &ms)) return -1;
    if (get_object(ms, "\"diffuse\"", &ds)) return -1;
    if (get_float(ds, "\"r\"", 1, &m->diffuse[0]) || get_float(ds, "\"g\"", 1, &m->diffuse[1]) || get_float(ds, "\"b\"", 1, &m->diffuse[2])) return -1;
    if (get_object(ms, "\"specular\"", &ss)) return -1;
    if (get_float(ss, "\"r\"", 1, &m->specular[0]) || get_float(ss, "\"g\"", 1, &m->specular[1]) || get_float(ss, "\"b\"", 1, &m->specular[2])) return -1;
    if (get_float(ms, "\"shininess\"", 1, &m->shininess)) return -1;
    if (get_float(ms, "\"transparency\"", 1, &m->transparency)) return -1;
    if (get_float(ms, "\"ior\"", 1, &m->ior)) return -1;
    char name[256];
    if (get_string(ms, "\"texture\"", name, sizeof name)) return -1;
    m->hasTex = 0; m->tex = -1;
    if (name[0] && cfg->textures) {                 /* sphere.cpp:106-109 */
        m->hasTex = 1;
        m->tex = tex_lookup(sc, texdir, name);
        if (m->tex < 0) return -1;
    }
    return 0;
}

static void plane_calc_normal(Plane* p) {           /* plane.cpp:304-321 */
    float (*v)[3] = p->vfile;
    float a[3] = { v[1][0] - v[0][0], v[1][1] - v[0][1], v[1][2] - v[0][2] };
    float b[3] = { v[2][0] - v[0][0], v[2][1] - v[0][1], v[2][2] - v[0][2] };
    p->normal[0] = a[1]*b[2] - a[2]*b[1];
    p->normal[1] = a[2]*b[0] - a[0]*b[2];
    p->normal[2] = a[0]*b[1] - a[1]*b[0];
    float len = sqrtf(p->normal[0]*p->normal[0] + p->normal[1]*p->normal[1] + p->normal[2]*p->normal[2]);
    p->normal[0] /= len; p->normal[1] /= len; p->normal[2] /= len;
}

static int drop_axis(const float n[3]) {            /* plane.cpp:207-216, 267-273 */
    float nx = fabsf(n[0]), ny = fabsf(n[1]), nz = fabsf(n[2]);
    if (nx > ny && nx > nz) return 0;
    else if (ny > nz) return 1;
    return 2;
}

static void plane_sort_winding(Plane* p) {          /* plane.cpp:254-301 */
    int n = p->n;
    float c[3] = {0.0f, 0.0f, 0.0f};
    for (int i = 0; i < n; ++i) { c[0] += p->vfile[i][0]; c[1] += p->vfile[i][1]; c[2] += p->vfile[i][2]; }
    float fn = (float)(size_t)n;
    c[0] /= fn; c[1] /= fn; c[2] /= fn;
    int da = drop_axis(p->normal);
    float* ang = (float*)malloc(sizeof(float) * (size_t)n);
    int* idx = (int*)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        const float* v = p->vfile[i];
        float x, y, cx, cy;
        if (da == 0) { x = v[1]; y = v[2]; cx = c[1]; cy = c[2]; }
        else if (da == 1) { x = v[0]; y = v[2]; cx = c[0]; cy = c[2]; }
        else { x = v[0]; y = v[1]; cx = c[0]; cy = c[1]; }
        ang[i] = atan2f(y - cy, x - cx);
        idx[i] = i;
    }
    /* std::sort on <= 16 elements is an insertion sort that never moves an element past an equal one */
    for (int i = 1; i < n; ++i) {
        int k = idx[i]; float a = ang[k]; int j = i - 1;
        while (j >= 0 && a < ang[idx[j]]) { idx[j + 1] = idx[j]; j--; }
        idx[j + 1] = k;
    }
    for (int i = 0; i < n; ++i) memcpy(p->vsorted[i], p->vfile[idx[i]], sizeof(float) * 3);
    free(ang); free(idx);
}

static void free_node(Node* n) {
    if (!n) return;
    free_node(n->left); free_node(n->right);
    free(n->planes); free(n->cubes); free(n->spheres); free(n);
}
static void free_scene(Scene* sc) {
    for (int i = 0; i < sc->nplanes; ++i) { free(sc->planes[i].vfile); free(sc->planes[i].vsorted); }
    for (int i = 0; i < sc->ntex; ++i) free(sc->tex[i].px);
    free(sc->cams); free(sc->lights); free(sc->spheres); free(sc->cubes); free(sc->planes); free(sc->tex);
    free_node(sc->bvh);
    memset(sc, 0, sizeof *sc);
}

static int parse_scene(const char* text, const orc_config* cfg, const char* texdir, Scene* sc) {
    memset(sc, 0, sizeof *sc);
    Str all = { text, strlen(text) }, props, arr; Str* objs = NULL; int n;

    /* cameras: camera.cpp:30-104 */
    if (get_object(all, "\"properties\"", &props)) return -1;
    if (get_array(props, "\"cameras\"", &arr)) return -1;
    n = split_objects(arr, &objs);
    if (n == 0) fprintf(stderr, "No cameras.\n");
    sc->cams = (Camera*)calloc((size_t)(n > 0 ? n : 1), sizeof(Camera)); sc->ncams = n;
    for (int i = 0; i < n; ++i) {
        Camera* c = &sc->cams[i]; Str o = objs[i], s2;
        if (get_xyz(o, "\"location\"", "\"x\"", "\"y\"", "\"z\"", 1, c->loc)) goto fail;
        if (get_xyz(o, "\"gaze_vector\"", "\"x\"", "\"y\"", "\"z\"", 1, c->gaze)) goto fail;
        if (cfg->dof) {
            if (get_float(o, "\"aperture\"", 1, &c->aperture)) goto fail;
            if (get_float(o, "\"focal_distance\"", 1, &c->focal_distance)) goto fail;
        } else { c->aperture = 0.0f; c->focal_distance = 0.0f; }
        float fl; if (get_float(o, "\"focal_length\"", 1, &fl)) goto fail;
        c->focal_length = fl / 1000.0f;
        if (get_object(o, "\"sensor\"", &s2)) goto fail;
        float w, h;
        if (get_float(s2, "\"width\"", 1, &w) || get_float(s2, "\"height\"", 1, &h)) goto fail;
        c->sw = w / 1000.0f; c->sh = h / 1000.0f;
        if (get_object(o, "\"film_resolution\"", &s2)) goto fail;
        if (get_float(s2, "\"width\"", 1, &c->res_x) || get_float(s2, "\"height\"", 1, &c->res_y)) goto fail;
    }
    free(objs); objs = NULL;

    /* lights: pointlight.cpp:25-75 (number scanner without 'e') */
    if (get_array(props, "\"point_lights\"", &arr)) return -1;
    n = split_objects(arr, &objs);
    if (n == 0) fprintf(stderr, "No lights.\n");
    sc->lights = (Light*)calloc((size_t)(n > 0 ? n : 1), sizeof(Light)); sc->nlights = n;
    for (int i = 0; i < n; ++i) {
        if (get_xyz(objs[i], "\"location\"", "\"x\"", "\"y\"", "\"z\"", 0, sc->lights[i].loc)) goto fail;
        if (get_float(objs[i], "\"radiant_intensity\"", 0, &sc->lights[i].intensity)) goto fail;
    }
    free(objs); objs = NULL;

    /* spheres: sphere.cpp:33-112 */
    if (get_array(props, "\"spheres\"", &arr)) return -1;
    n = split_objects(arr, &objs);
    if (n == 0) fprintf(stderr, "No spheres.\n");
    sc->spheres = (Sphere*)calloc((size_t)(n > 0 ? n : 1), sizeof(Sphere)); sc->nspheres = n;
    for (int i = 0; i < n; ++i) {
        Sphere* s = &sc->spheres[i];
        if (get_xyz(objs[i], "\"start_location\"", "\"x\"", "\"y\"", "\"z\"", 1, s->start)) goto fail;
        if (get_xyz(objs[i], "\"end_location\"", "\"x\"", "\"y\"", "\"z\"", 1, s->end)) goto fail;
        if (get_float(objs[i], "\"radius\"", 1, &s->radius)) goto fail;
        if (parse_material(sc, objs[i], cfg, texdir, &s->m)) goto fail;
    }
    free(objs); objs = NULL;

    /* cubes: cube.cpp:34-119 */
    if (get_array(props, "\"cubes\"", &arr)) return -1;
    n = split_objects(arr, &objs);
    if (n == 0) fprintf(stderr, "No cubes.\n");
    sc->cubes = (Cube*)calloc((size_t)(n > 0 ? n : 1), sizeof(Cube)); sc->ncubes = n;
    for (int i = 0; i < n; ++i) {
        Cube* c = &sc->cubes[i];
        if (get_xyz(objs[i], "\"start_location\"", "\"x\"", "\"y\"", "\"z\"", 1, c->start)) goto fail;
        if (get_xyz(objs[i], "\"end_location\"", "\"x\"", "\"y\"", "\"z\"", 1, c->end)) goto fail;
        if (get_xyz(objs[i], "\"rotation\"", "\"x\"", "\"y\"", "\"z\"", 1, c->rot)) goto fail;
        if (get_float(objs[i], "\"scale\"", 1, &c->scale)) goto fail;
        if (parse_material(sc, objs[i], cfg, texdir, &c->m)) goto fail;
    }
    free(objs); objs = NULL;

    /* planes: plane.cpp:34-137 */
    if (get_array(props, "\"planes\"", &arr)) return -1;
    n = split_objects(arr, &objs);
    if (n == 0) fprintf(stderr, "No planes.\n");
    sc->planes = (Plane*)calloc((size_t)(n > 0 ? n : 1), sizeof(Plane)); sc->nplanes = n;
    for (int i = 0; i < n; ++i) {
        Plane* p = &sc->planes[i]; Str carr; Str* cs = NULL;
        if (get_array(objs[i], "\"corners\"", &carr)) goto fail;
        int nc = split_objects(carr, &cs);
        if (nc < 3) { free(cs); set_err("plane %d has %d corners", i, nc); goto fail; }
        p->n = nc;
        p->vfile = (float(*)[3])calloc((size_t)nc, sizeof(float[3]));
        p->vsorted = (float(*)[3])calloc((size_t)nc, sizeof(float[3]));
        for (int k = 0; k < nc; ++k) {
            if (get_float(cs[k], "\"x\"", 1, &p->vfile[k][0]) || get_float(cs[k], "\"y\"", 1, &p->vfile[k][1]) ||
                get_float(cs[k], "\"z\"", 1, &p->vfile[k][2])) { free(cs); goto fail; }
        }
        free(cs);
        plane_calc_normal(p);
        plane_sort_winding(p);
        p->main_sorted = 0;
        if (parse_material(sc, objs[i], cfg, texdir, &p->m)) goto fail;
    }
    free(objs);
    return 0;
fail:
    free(objs);
    free_scene(sc);
    return -1;
}

/* ------------------------------------------------------------------------------------------------
 * Shape AABBs (sphere.cpp:177-188, cube.cpp:354-386, plane.cpp:324-340, aabb.cpp:67-156)
 * ---------------------------------------------------------------------------------------------- */
static void box_init(float mn[3], float mx[3]) {
    for (int i = 0; i < 3; ++i) { mn[i] = INFINITY; mx[i] = -INFINITY; }
}
static void box_grow_pt(float mn[3], float mx[3], const float p[3]) {
    for (int i = 0; i < 3; ++i) { if (p[i] < mn[i]) mn[i] = p[i]; if (p[i] > mx[i]) mx[i] = p[i]; }
}

static void sphere_aabb(const Sphere* s, float mn[3], float mx[3]) {     /* sphere.cpp:177-188 + aabb.cpp:135-156 */
    float r = s->radius;
    box_init(mn, mx);
    for (int i = 0; i < 3; ++i) {
        float smin_ = s->start[i] - r, smax_ = s->start[i] + r, emin = s->end[i] - r, emax = s->end[i] + r;
        if (smin_ < mn[i]) mn[i] = smin_;
        if (emin < mn[i]) mn[i] = emin;
        if (smax_ > mx[i]) mx[i] = smax_;
        if (emax > mx[i]) mx[i] = emax;
    }
}

static void rotate_xyz(float v[3], const float rot[3]) {                 /* cube.cpp:297-318 */
    float x = v[0], y = v[1], z = v[2];
    float cx = cosf(rot[0]), sx = sinf(rot[0]);
    float y1 = y * cx - z * sx, z1 = y * sx + z * cx; y = y1; z = z1;
    float cy = cosf(rot[1]), sy = sinf(rot[1]);
    float x2 = x * cy + z * sy, z2 = -x * sy + z * cy; x = x2; z = z2;
    float cz = cosf(rot[2]), sz = sinf(rot[2]);
    float x3 = x * cz - y * sz, y3 = x * sz + y * cz;
    v[0] = x3; v[1] = y3; v[2] = z;
}
static void rotate_xyz_inv(float v[3], const float rot[3]) {             /* cube.cpp:320-351 */
    float ix = -rot[0], iy = -rot[1], iz = -rot[2];
    { float x = v[0], y = v[1]; float cz = cosf(iz), sz = sinf(iz); v[0] = x * cz - y * sz; v[1] = x * sz + y * cz; }
    { float x = v[0], z = v[2]; float cy = cosf(iy), sy = sinf(iy); v[0] = x * cy + z * sy; v[2] = -x * sy + z * cy; }
    { float y = v[1], z = v[2]; float cx = cosf(ix), sx = sinf(ix); v[1] = y * cx - z * sx; v[2] = y * sx + z * cx; }
}

static void cube_aabb(const Cube* c, float mn[3], float mx[3]) {         /* cube.cpp:354-386 */
    static const float sg[2] = { -1.0f, 1.0f };
    box_init(mn, mx);
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int d = 0; d < 2; ++d) {
        float sc_[3] = { sg[a], sg[b], sg[d] };
        rotate_xyz(sc_, c->rot);
        float p[3] = { sc_[0] * c->scale + c->start[0], sc_[1] * c->scale + c->start[1], sc_[2] * c->scale + c->start[2] };
        box_grow_pt(mn, mx, p);
        /* end corner: rotated but NOT scaled (cube.cpp:376-379 assigns the scaled value to startC) */
        float ec[3] = { sg[a], sg[b], sg[d] };
        rotate_xyz(ec, c->rot);
        float q[3] = { ec[0] + c->end[0], ec[1] + c->end[1], ec[2] + c->end[2] };
        box_grow_pt(mn, mx, q);
    }
}

static void plane_aabb(const Plane* p, float mn[3], float mx[3]) {       /* plane.cpp:324-340: max starts at FLT_MIN */
    for (int i = 0; i < 3; ++i) { mn[i] = FLT_MAX; mx[i] = FLT_MIN; }
    for (int k = 0; k < p->n; ++k)
        for (int i = 0; i < 3; ++i) {
            if (p->vfile[k][i] < mn[i]) mn[i] = p->vfile[k][i];
            if (p->vfile[k][i] > mx[i]) mx[i] = p->vfile[k][i];
        }
}

/* ------------------------------------------------------------------------------------------------
 * BVH build (bvh.cpp:108-172)
 * ---------------------------------------------------------------------------------------------- */
static void group_aabb(const Scene* sc, const int* pl, int np, const int* cu, int nc, const int* sp, int ns,
                       float mn[3], float mx[3]) {                        /* aabb.cpp:67-111 */
    float a[3], b[3];
    box_init(mn, mx);
    for (int i = 0; i < np; ++i) { plane_aabb(&sc->planes[pl[i]], a, b);
        for (int k = 0; k < 3; ++k) { if (a[k] < mn[k]) mn[k] = a[k]; if (b[k] > mx[k]) mx[k] = b[k]; } }
    for (int i = 0; i < nc; ++i) { cube_aabb(&sc->cubes[cu[i]], a, b);
        for (int k = 0; k < 3; ++k) { if (a[k] < mn[k]) mn[k] = a[k]; if (b[k] > mx[k]) mx[k] = b[k]; } }
    for (int i = 0; i < ns; ++i) { sphere_aabb(&sc->spheres[sp[i]], a, b);
        for (int k = 0; k < 3; ++k) { if (a[k] < mn[k]) mn[k] = a[k]; if (b[k] > mx[k]) mx[k] = b[k]; } }
}

static int* dup_ints(const int* v, int n) {
    int* r = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    if (n > 0) memcpy(r, v, sizeof(int) * (size_t)n);
    return r;
}

static Node* build_bvh(const Scene* sc, const int* pl, int np, const int* cu, int nc, const int* sp, int ns, int maxDepth) {
    Node* node = (Node*)calloc(1, sizeof(Node));
    group_aabb(sc, pl, np, cu, nc, sp, ns, node->mn, node->mx);
    if (maxDepth == 0) {                                                  /* bvh.cpp:109-117 */
        node->planes = dup_ints(pl, np); node->np = np;
        node->cubes = dup_ints(cu, nc); node->nc = nc;
        node->spheres = dup_ints(sp, ns); node->ns = ns;
        return node;
    }
    const float* mn = node->mn; const float* mx = node->mx;
    int axis = 0;                                                         /* bvh.cpp:123-129 */
    if (mx[1] - mn[1] > mx[0] - mn[0]) axis = 1;
    if (mx[2] - mn[2] > mx[0] - mn[0] && mx[2] - mn[2] > mx[1] - mn[1]) axis = 2;
    float split = (mn[axis] + mx[axis]) / 2.0f;                           /* bvh.cpp:139 */
    int* lp = (int*)malloc(sizeof(int) * (size_t)(np + 1)); int* rp = (int*)malloc(sizeof(int) * (size_t)(np + 1));
    int* lc = (int*)malloc(sizeof(int) * (size_t)(nc + 1)); int* rc = (int*)malloc(sizeof(int) * (size_t)(nc + 1));
    int* ls = (int*)malloc(sizeof(int) * (size_t)(ns + 1)); int* rs = (int*)malloc(sizeof(int) * (size_t)(ns + 1));
    int nlp = 0, nrp = 0, nlc = 0, nrc = 0, nls = 0, nrs = 0;
    float a[3], b[3];
    for (int i = 0; i < np; ++i) { plane_aabb(&sc->planes[pl[i]], a, b);
        if ((a[axis] + b[axis]) / 2.0f < split) lp[nlp++] = pl[i]; else rp[nrp++] = pl[i]; }
    for (int i = 0; i < nc; ++i) { cube_aabb(&sc->cubes[cu[i]], a, b);
        if ((a[axis] + b[axis]) / 2.0f < split) lc[nlc++] = cu[i]; else rc[nrc++] = cu[i]; }
    for (int i = 0; i < ns; ++i) { sphere_aabb(&sc->spheres[sp[i]], a, b);
        if ((a[axis] + b[axis]) / 2.0f < split) ls[nls++] = sp[i]; else rs[nrs++] = sp[i]; }
    if (nlp || nlc || nls) node->left = build_bvh(sc, lp, nlp, lc, nlc, ls, nls, maxDepth - 1);
    if (nrp || nrc || nrs) node->right = build_bvh(sc, rp, nrp, rc, nrc, rs, nrs, maxDepth - 1);
    free(lp); free(rp); free(lc); free(rc); free(ls); free(rs);
    return node;
}

static void scene_build_bvh(Scene* sc) {                                  /* raytracer.cpp:72-75 */
    int* pl = (int*)malloc(sizeof(int) * (size_t)(sc->nplanes + 1));
    int* cu = (int*)malloc(sizeof(int) * (size_t)(sc->ncubes + 1));
    int* sp = (int*)malloc(sizeof(int) * (size_t)(sc->nspheres + 1));
    for (int i = 0; i < sc->nplanes; ++i) pl[i] = i;
    for (int i = 0; i < sc->ncubes; ++i) cu[i] = i;
    for (int i = 0; i < sc->nspheres; ++i) sp[i] = i;
    sc->bvh = build_bvh(sc, pl, sc->nplanes, cu, sc->ncubes, sp, sc->nspheres, 8);
    free(pl); free(cu); free(sp);
}

/* ------------------------------------------------------------------------------------------------
 * Primitive intersection
 * ---------------------------------------------------------------------------------------------- */
static void hit_reset(Hit* h) { memset(h, 0, sizeof *h); }

static int sphere_intersect(const Sphere* s, const Ray* ray, Hit* hs, const orc_config* cfg) {   /* sphere.cpp:118-174 */
    float loc[3];
    if (cfg->motionBlur) {                                                /* shape.cpp:138-144 */
        loc[0] = lerpf(s->start[0], s->end[0], ray->time);
        loc[1] = lerpf(s->start[1], s->end[1], ray->time);
        loc[2] = lerpf(s->start[2], s->end[2], ray->time);
    } else { loc[0] = s->start[0]; loc[1] = s->start[1]; loc[2] = s->start[2]; }
    float l[3] = { loc[0] - ray->o[0], loc[1] - ray->o[1], loc[2] - ray->o[2] };
    float tca = l[0] * ray->d[0] + l[1] * ray->d[1] + l[2] * ray->d[2];
    if (tca < 0) return 0;
    float d2 = l[0]*l[0] + l[1]*l[1] + l[2]*l[2] - tca*tca;
    if (d2 > s->radius * s->radius) return 0;
    float thc = sqrtf(s->radius * s->radius - d2);
    float t0 = tca - thc, t1 = tca + thc;
    float t = (t0 > 0) ? t0 : t1;
    if (t < 0) return 0;
    float ip[3] = { ray->o[0] + t * ray->d[0], ray->o[1] + t * ray->d[1], ray->o[2] + t * ray->d[2] };
    if (s->m.hasTex) {
        float nn[3] = { ip[0] - loc[0], ip[1] - loc[1], ip[2] - loc[2] };
        normalise(nn);
        float u = (float)(0.5f + atan2f(nn[2], nn[0]) / (2 * M_PI));
        float v = (float)(0.5f - asinf(nn[1]) / M_PI);
        hs->u = u; hs->v = v; hs->hasTex = 1;
    }
    hs->p[0] = ip[0]; hs->p[1] = ip[1]; hs->p[2] = ip[2];
    hs->n[0] = (ip[0] - loc[0]) / s->radius; hs->n[1] = (ip[1] - loc[1]) / s->radius; hs->n[2] = (ip[2] - loc[2]) / s->radius;
    hs->t = t; hs->m = &s->m; hs->time = ray->time;
    return 1;
}

static int cube_intersect(const Cube* c, const Ray* ray, Hit* hs, const orc_config* cfg) {       /* cube.cpp:126-292 */
    float loc[3];
    if (cfg->motionBlur) {
        loc[0] = lerpf(c->start[0], c->end[0], ray->time);
        loc[1] = lerpf(c->start[1], c->end[1], ray->time);
        loc[2] = lerpf(c->start[2], c->end[2], ray->time);
    } else { loc[0] = c->start[0]; loc[1] = c->start[1]; loc[2] = c->start[2]; }
    float o[3] = { ray->o[0] - loc[0], ray->o[1] - loc[1], ray->o[2] - loc[2] };
    float d[3] = { ray->d[0], ray->d[1], ray->d[2] };
    rotate_xyz_inv(o, c->rot);
    rotate_xyz_inv(d, c->rot);
    float inv = 1.0f / c->scale;
    float lo[3] = { o[0] * inv, o[1] * inv, o[2] * inv };
    float ld[3] = { d[0] * inv, d[1] * inv, d[2] * inv };
    normalise(ld);
    float tmin = -INFINITY, tmax = INFINITY;
    for (int i = 0; i < 3; ++i) {
        if (fabsf(ld[i]) < 1e-5f) {
            if (lo[i] < -1 || lo[i] > 1) return 0;
        } else {
            float t1 = (-1 - lo[i]) / ld[i];
            float t2 = ( 1 - lo[i]) / ld[i];
            if (t1 > t2) { float tt = t1; t1 = t2; t2 = tt; }
            tmin = smax(tmin, t1);
            tmax = smin(tmax, t2);
            if (tmin > tmax) return 0;
        }
    }
    const float T_EPS = 1e-4f;
    float t = -1.0f;
    if (tmin > T_EPS) t = tmin; else if (tmax > T_EPS) t = tmax;
    if (t < 0.0f) return 0;
    float hl[3] = { lo[0] + ld[0] * t, lo[1] + ld[1] * t, lo[2] + ld[2] * t };
    float nl[3] = { 0.0f, 0.0f, 0.0f };
    float eps = 1e-5f;
    float ax = fabsf(hl[0]), ay = fabsf(hl[1]), az = fabsf(hl[2]);
    if (ax >= ay - eps && ax >= az - eps) nl[0] = (hl[0] > 0) ? 1.0f : -1.0f;
    else if (ay >= ax - eps && ay >= az - eps) nl[1] = (hl[1] > 0) ? 1.0f : -1.0f;
    else nl[2] = (hl[2] > 0) ? 1.0f : -1.0f;
    if (c->m.hasTex) {                                                    /* cube.cpp:197-263 */
        float u = 0.0f, v = 0.0f;
        float maxA = smax(ax, smax(ay, az));
        if (ax >= maxA - eps) {
            if (hl[0] > 0.0f) { u = (hl[2] + 1.0f) * 0.5f; v = (hl[1] + 1.0f) * 0.5f; }
            else { u = (1.0f - hl[2]) * 0.5f; v = (hl[1] + 1.0f) * 0.5f; }
        } else if (ay >= maxA - eps) {
            if (hl[1] > 0.0f) { u = (hl[0] + 1.0f) * 0.5f; v = (hl[2] + 1.0f) * 0.5f; }
            else { u = (hl[0] + 1.0f) * 0.5f; v = (1.0f - hl[2]) * 0.5f; }
        } else {
            if (hl[2] > 0.0f) { u = (hl[0] + 1.0f) * 0.5f; v = (hl[1] + 1.0f) * 0.5f; }
            else { u = (1.0f - hl[0]) * 0.5f; v = (hl[1] + 1.0f) * 0.5f; }
        }
        u = fmodf(u, 1.0f); v = fmodf(v, 1.0f);
        if (u < 0.0f) u += 1.0f;
        if (v < 0.0f) v += 1.0f;
        if (u < 0.0f) u = 0.0f;
        if (u > 1.0f) u = 1.0f;
        if (v < 0.0f) v = 0.0f;
        if (v > 1.0f) v = 1.0f;
        hs->u = u; hs->v = v; hs->hasTex = 1;
    }
    float hw[3] = { hl[0] * c->scale, hl[1] * c->scale, hl[2] * c->scale };
    rotate_xyz(hw, c->rot);
    hw[0] += loc[0]; hw[1] += loc[1]; hw[2] += loc[2];
    float vth[3] = { hw[0] - ray->o[0], hw[1] - ray->o[1], hw[2] - ray->o[2] };
    float worldT = dotp(vth, ray->d);
    worldT = smax(worldT, 1e-4f);
    rotate_xyz(nl, c->rot);
    normalise(nl);
    hs->p[0] = hw[0]; hs->p[1] = hw[1]; hs->p[2] = hw[2];
    hs->n[0] = nl[0]; hs->n[1] = nl[1]; hs->n[2] = nl[2];
    hs->t = worldT; hs->m = &c->m; hs->time = ray->time;
    return 1;
}

static int point_in_polygon(const float (*v)[3], int n, const float normal[3], const float pt[3]) {   /* plane.cpp:205-251 */
    int da = drop_axis(normal);
    int i0 = (da == 0) ? 1 : 0;
    int i1 = (da == 2) ? 1 : 2;
    float Px = pt[i0], Py = pt[i1];
    int inside = 0;
    for (int i = 0, j = n - 1; i < n; j = i++) {
        float Pix = v[i][i0], Piy = v[i][i1], Pjx = v[j][i0], Pjy = v[j][i1];
        if (fabsf(Pjy - Piy) < 1e-8f) continue;
        int iy = ((Piy > Py) != (Pjy > Py));
        if (iy) {
            float slope = (Pjx - Pix) / (Pjy - Piy);
            float xi = slope * (Py - Piy) + Pix;
            if (Px < xi) inside = !inside;
        }
    }
    return inside;
}

/* plane.cpp:143-201. v0 = the vertex used for t (file-order first vertex for a fresh copy, or whatever is first
 * in the mutated main-vector object); vs = the sorted polygon used for the inside test and for UVs. */
static int plane_intersect_core(const Plane* p, const float v0[3], const Ray* ray, Hit* hs) {
    const float* nrm = p->normal;
    float denom = nrm[0] * ray->d[0] + nrm[1] * ray->d[1] + nrm[2] * ray->d[2];
    if ((double)fabsf(denom) < 1e-10) return 0;
    float t = (nrm[0] * (v0[0] - ray->o[0]) + nrm[1] * (v0[1] - ray->o[1]) + nrm[2] * (v0[2] - ray->o[2])) / denom;
    if (t < 0) return 0;
    float ip[3] = { ray->o[0] + t * ray->d[0], ray->o[1] + t * ray->d[1], ray->o[2] + t * ray->d[2] };
    const float (*vs)[3] = (const float (*)[3])p->vsorted;
    if (!point_in_polygon(vs, p->n, nrm, ip)) return 0;
    if (p->m.hasTex) {                                                    /* plane.cpp:163-184 */
        float us[3] = { vs[1][0] - vs[0][0], vs[1][1] - vs[0][1], vs[1][2] - vs[0][2] };
        float ulen = (float)sqrt(pow((double)us[0], 2.0) + pow((double)us[1], 2.0) + pow((double)us[2], 2.0));
        normalise(us);
        const float* vb = vs[p->n - 1];
        float vsd[3] = { vb[0] - vs[0][0], vb[1] - vs[0][1], vb[2] - vs[0][2] };
        float vlen = (float)sqrt(pow((double)vsd[0], 2.0) + pow((double)vsd[1], 2.0) + pow((double)vsd[2], 2.0));
        normalise(vsd);
        float loc[3] = { ip[0] - vs[0][0], ip[1] - vs[0][1], ip[2] - vs[0][2] };
        float u = dotp(loc, us) / ulen;
        float v = dotp(loc, vsd) / vlen;
        u = fmodf(smax(u, 0.0f), 1.0f);
        v = fmodf(smax(v, 0.0f), 1.0f);
        hs->u = u; hs->v = v; hs->hasTex = 1;
    }
    hs->p[0] = ip[0]; hs->p[1] = ip[1]; hs->p[2] = ip[2];
    hs->n[0] = nrm[0]; hs->n[1] = nrm[1]; hs->n[2] = nrm[2];
    hs->t = t; hs->m = &p->m;
    if (ray->time) hs->time = ray->time;                                  /* plane.cpp:194-196 */
    return 1;
}
/* Called on a per-visit copy of the leaf's plane (bvh.cpp:70): t always uses the file-order first vertex. */
static int plane_intersect_copy(const Plane* p, const Ray* ray, Hit* hs) {
    return plane_intersect_core(p, p->vfile[0], ray, hs);
}
/* Called on the object in main()'s vector (raytracer.cpp:213): the first call sees file order, then sorts in place. */
static int plane_intersect_main(Plane* p, const Ray* ray, Hit* hs) {
    const float* v0 = p->main_sorted ? p->vsorted[0] : p->vfile[0];
    float denom = p->normal[0] * ray->d[0] + p->normal[1] * ray->d[1] + p->normal[2] * ray->d[2];
    int r = plane_intersect_core(p, v0, ray, hs);
    /* sortVerticesWinding runs only if the parallel and t<0 early-outs were passed (plane.cpp:145-158) */
    if (!p->main_sorted && !((double)fabsf(denom) < 1e-10)) {
        float t = (p->normal[0] * (v0[0] - ray->o[0]) + p->normal[1] * (v0[1] - ray->o[1]) + p->normal[2] * (v0[2] - ray->o[2])) / denom;
        if (!(t < 0)) p->main_sorted = 1;
    }
    return r;
}

/* ------------------------------------------------------------------------------------------------
 * AABB slab test (aabb.cpp:31-63) and traversal (bvh.cpp:35-105)
 * ---------------------------------------------------------------------------------------------- */
static int aabb_intersect(const float mn[3], const float mx[3], const Ray* ray, float* tMin, float* tMax) {
    float tmn[3], tmx[3];
    if (g_stats) g_stats->box_tests++;
    for (int i = 0; i < 3; ++i) {
        if (ray->d[i] >= 0) { tmn[i] = (mn[i] - ray->o[i]) / ray->d[i]; tmx[i] = (mx[i] - ray->o[i]) / ray->d[i]; }
        else                { tmn[i] = (mx[i] - ray->o[i]) / ray->d[i]; tmx[i] = (mn[i] - ray->o[i]) / ray->d[i]; }
    }
    *tMin = smax(tmn[0], smax(tmn[1], tmn[2]));
    *tMax = smin(tmx[0], smin(tmx[1], tmx[2]));
    return *tMin <= *tMax;
}

static int bvh_intersect(const Scene* sc, const Node* nd, Ray* ray, float* tMin, float* tMax, const orc_config* cfg) {
    if (!aabb_intersect(nd->mn, nd->mx, ray, tMin, tMax)) return 0;
    int hit = 0;
    float tlmin = *tMin, tlmax = *tMax, trmin = *tMin, trmax = *tMax;
    if (nd->left && bvh_intersect(sc, nd->left, ray, &tlmin, &tlmax, cfg)) { *tMin = tlmin; *tMax = tlmax; hit = 1; }
    if (nd->right && bvh_intersect(sc, nd->right, ray, &trmin, &trmax, cfg)) {
        if (!hit || trmin < *tMin) { *tMin = trmin; *tMax = trmax; }
        hit = 1;
    }
    if (hit) return 1;
    if (!nd->left && !nd->right) {                                        /* bvh.cpp:65-102 */
        Hit closest; hit_reset(&closest);
        float closestDist = INFINITY;
        for (int i = 0; i < nd->np; ++i) {
            Hit hs; hit_reset(&hs);
            if (g_stats) g_stats->prim_tests++;
            if (plane_intersect_copy(&sc->planes[nd->planes[i]], ray, &hs) && hs.t >= 0.0f && hs.t < closestDist) {
                hit = 1; closestDist = hs.t; closest = hs;
            }
        }
        for (int i = 0; i < nd->nc; ++i) {
            Hit hs; hit_reset(&hs);
            if (g_stats) g_stats->prim_tests++;
            if (cube_intersect(&sc->cubes[nd->cubes[i]], ray, &hs, cfg) && hs.t >= 0.0f && hs.t < closestDist) {
                hit = 1; closestDist = hs.t; closest = hs;
            }
        }
        for (int i = 0; i < nd->ns; ++i) {
            Hit hs; hit_reset(&hs);
            if (g_stats) g_stats->prim_tests++;
            if (sphere_intersect(&sc->spheres[nd->spheres[i]], ray, &hs, cfg) && hs.t >= 0.0f && hs.t < closestDist) {
                hit = 1; closestDist = hs.t; closest = hs;
            }
        }
        if (hit) { ray->last = closest; ray->nh++; *tMax = closestDist; return 1; }
    }
    return 0;
}

static int unaccelerated(Scene* sc, Ray* ray, const orc_config* cfg) {   /* raytracer.cpp:202-256 */
    Hit hs; hit_reset(&hs);
    int hit = 0; float closestDist = FLT_MAX;
    for (int p = 0; p < sc->nplanes; ++p) {
        Hit temp; hit_reset(&temp);
        if (g_stats) g_stats->prim_tests++;
        if (plane_intersect_main(&sc->planes[p], ray, &temp) && temp.t < closestDist) { closestDist = temp.t; hs = temp; hit = 1; }
    }
    for (int c = 0; c < sc->ncubes; ++c) {
        Hit temp; hit_reset(&temp);
        if (g_stats) g_stats->prim_tests++;
        if (cube_intersect(&sc->cubes[c], ray, &temp, cfg) && temp.t < closestDist) { closestDist = temp.t; hs = temp; hit = 1; }
    }
    for (int s = 0; s < sc->nspheres; ++s) {
        Hit temp; hit_reset(&temp);
        if (g_stats) g_stats->prim_tests++;
        if (sphere_intersect(&sc->spheres[s], ray, &temp, cfg) && temp.t < closestDist) { closestDist = temp.t; hs = temp; hit = 1; }
    }
    if (hit) { ray->last = hs; ray->nh++; return 1; }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Random helpers (raytracer.cpp:634-668, camera.cpp:286-312)
 * ---------------------------------------------------------------------------------------------- */
static inline float rnd01(void) { return (float)rand() / (float)RAND_MAX; }   /* static_cast<float>(rand()) / RAND_MAX */

static void rnd_unit_sphere(float out[3]) {                               /* raytracer.cpp:634-644 */
    float u = rnd01();
    float v = rnd01();
    float theta = (float)(u * 2.0f * M_PI);
    float phi = acosf(2.0f * v - 1.0f);
    float r = cbrtf(rnd01());
    out[0] = r * sinf(phi) * cosf(theta);
    out[1] = r * sinf(phi) * sinf(theta);
    out[2] = r * cosf(phi);
}

static void rnd_cone_direction(const float dir[3], float angleRad, float out[3]) {   /* raytracer.cpp:647-668 */
    float rv[3]; rnd_unit_sphere(rv);               /* drawn and discarded */
    float cosTheta = cosf(angleRad);
    float z = cosTheta + (1.0f - cosTheta) * ((rand() % 1000) / 1000.0f);
    float phi = (float)(2.0f * M_PI * ((rand() % 1000) / 1000.0f));
    float r = sqrtf(1.0f - z * z);
    rv[0] = r * cosf(phi); rv[1] = r * sinf(phi); rv[2] = z;
    const float up[3] = { 0.0f, 1.0f, 0.0f };
    float u[3]; cross(up, dir, u); normalise(u);
    float v[3]; cross(dir, u, v);
    out[0] = rv[0] * u[0] + rv[1] * v[0] + rv[2] * dir[0];
    out[1] = rv[0] * u[1] + rv[1] * v[1] + rv[2] * dir[1];
    out[2] = rv[0] * u[2] + rv[1] * v[2] + rv[2] * dir[2];
    normalise(out);
}

static void sample_disk(float* sx, float* sy) {                           /* camera.cpp:286-312 */
    float x = 2.0f * rnd01() - 1.0f;
    float y = 2.0f * rnd01() - 1.0f;
    if (x == 0.0f && y == 0.0f) { *sx = 0.0f; *sy = 0.0f; return; }
    float theta, r;
    if (fabsf(x) > fabsf(y)) { r = x; theta = (float)((M_PI / 4.0f) * (y / x)); }
    else { r = y; theta = (float)((M_PI / 2.0f) - (M_PI / 4.0f) * (x / y)); }
    *sx = r * cosf(theta);
    *sy = r * sinf(theta);
}

/* ------------------------------------------------------------------------------------------------
 * Camera (camera.cpp:109-199)
 * ---------------------------------------------------------------------------------------------- */
static void camera_ray(const Camera* c, float x, float y, Ray* ray) {
    memset(ray, 0, sizeof *ray);
    ray->time = rnd01();                                                  /* camera.cpp:112 */
    float u = (x + 0.5f) / c->res_x;
    float v = (y + 0.5f) / c->res_y;
    float cam_x = -(u - 0.5f) * (c->sw / c->focal_length);
    float cam_y = (0.5f - v) * (c->sh / c->focal_length);
    float cam_z = -1.0f;
    float fwd[3] = { c->gaze[0], c->gaze[1], c->gaze[2] };
    normalise(fwd);
    float wup[3] = { 0.0f, 0.0f, 1.0f };
    float right[3] = { fwd[1]*wup[2] - fwd[2]*wup[1], fwd[2]*wup[0] - fwd[0]*wup[2], fwd[0]*wup[1] - fwd[1]*wup[0] };
    float rlen = sqrtf(right[0]*right[0] + right[1]*right[1] + right[2]*right[2]);
    if (rlen < 1e-6f) {
        right[0] = wup[1]*fwd[2] - wup[2]*fwd[1];
        right[1] = wup[2]*fwd[0] - wup[0]*fwd[2];
        right[2] = wup[0]*fwd[1] - wup[1]*fwd[0];
    }
    normalise(right);
    float up[3] = { fwd[1]*right[2] - fwd[2]*right[1], fwd[2]*right[0] - fwd[0]*right[2], fwd[0]*right[1] - fwd[1]*right[0] };
    ray->d[0] = cam_x*right[0] + cam_y*up[0] + cam_z*fwd[0];
    ray->d[1] = cam_x*right[1] + cam_y*up[1] + cam_z*fwd[1];
    ray->d[2] = cam_x*right[2] + cam_y*up[2] + cam_z*fwd[2];
    float dlen = sqrtf(ray->d[0]*ray->d[0] + ray->d[1]*ray->d[1] + ray->d[2]*ray->d[2]);
    if (dlen < 1e-6f) { ray->d[0] = 0.0f; ray->d[1] = 0.0f; ray->d[2] = 1.0f; }
    normalise(ray->d);
    ray->d[0] = -ray->d[0]; ray->d[1] = -ray->d[1]; ray->d[2] = -ray->d[2];
    ray->o[0] = c->loc[0]; ray->o[1] = c->loc[1]; ray->o[2] = c->loc[2];
    if (c->aperture > 0.0f) {                                             /* camera.cpp:181-196 */
        float lens_radius = c->focal_length / (2.0f * c->aperture);
        float sx, sy; sample_disk(&sx, &sy);
        sx *= lens_radius; sy *= lens_radius;
        float lo[3] = { right[0]*sx + up[0]*sy, right[1]*sx + up[1]*sy, right[2]*sx + up[2]*sy };
        float fp[3] = { ray->o[0] + ray->d[0]*c->focal_distance, ray->o[1] + ray->d[1]*c->focal_distance, ray->o[2] + ray->d[2]*c->focal_distance };
        ray->o[0] += lo[0]; ray->o[1] += lo[1]; ray->o[2] += lo[2];
        ray->d[0] = fp[0] - ray->o[0]; ray->d[1] = fp[1] - ray->o[1]; ray->d[2] = fp[2] - ray->o[2];
        normalise(ray->d);
    }
}

/* ------------------------------------------------------------------------------------------------
 * Shading (raytracer.cpp:262-571)
 * ---------------------------------------------------------------------------------------------- */
static float hard_shadow(Scene* sc, const float hp[3], float time, const float norm[3], const Light* light, const orc_config* cfg) {
    /* raytracer.cpp:509-530: tMin/tMax handed to the traversal are overwritten by the first slab test */
    float toL[3] = { light->loc[0] - hp[0], light->loc[1] - hp[1], light->loc[2] - hp[2] };
    normalise(toL);
    Ray sr; memset(&sr, 0, sizeof sr);
    sr.o[0] = hp[0] + norm[0] * 1e-2f; sr.o[1] = hp[1] + norm[1] * 1e-2f; sr.o[2] = hp[2] + norm[2] * 1e-2f;
    sr.d[0] = toL[0]; sr.d[1] = toL[1]; sr.d[2] = toL[2];
    sr.time = time;
    if (g_stats) g_stats->shadow++;
    int hit;
    if (cfg->bvh) { float a = 1e-4f, b = 0.0f; hit = bvh_intersect(sc, sc->bvh, &sr, &a, &b, cfg); }
    else hit = unaccelerated(sc, &sr, cfg);
    return hit ? 0.0f : 1.0f;
}

static float soft_shadow(Scene* sc, const float hp[3], float time, const Light* light, const orc_config* cfg) {
    /* raytracer.cpp:534-571 */
    int unblocked = 0;
    for (int i = 0; i < cfg->SSsamples; ++i) {
        float j[3]; rnd_unit_sphere(j);
        float lp[3] = { light->loc[0] + j[0] * cfg->lightRadius, light->loc[1] + j[1] * cfg->lightRadius, light->loc[2] + j[2] * cfg->lightRadius };
        float h2l[3] = { lp[0] - hp[0], lp[1] - hp[1], lp[2] - hp[2] };
        float ld[3] = { h2l[0], h2l[1], h2l[2] };
        normalise(ld);
        Ray sr; memset(&sr, 0, sizeof sr);
        sr.o[0] = hp[0] + ld[0] * 0.001f; sr.o[1] = hp[1] + ld[1] * 0.001f; sr.o[2] = hp[2] + ld[2] * 0.001f;
        sr.d[0] = ld[0]; sr.d[1] = ld[1]; sr.d[2] = ld[2];
        sr.time = time;
        float lightDist = (float)sqrt(pow((double)h2l[0], 2.0) + pow((double)h2l[1], 2.0) + pow((double)h2l[2], 2.0));
        if (g_stats) g_stats->shadow++;
        int hit;
        if (cfg->bvh) { float a = 0.0f, b = lightDist; hit = bvh_intersect(sc, sc->bvh, &sr, &a, &b, cfg); }
        else hit = unaccelerated(sc, &sr, cfg);
        if (!(hit && sr.last.t < lightDist)) unblocked++;
    }
    return (float)unblocked / (float)cfg->SSsamples;
}

static void blinn_phong(Scene* sc, const Hit* hs, const Camera* cam, const orc_config* cfg, float colour[3]) {
    /* raytracer.cpp:262-347 */
    float ka = 0.25f;
    float shiny = hs->m->shininess;
    float mat[3];
    if (hs->hasTex) {
        const Tex* tx = &sc->tex[hs->m->tex];
        float u = fmodf(hs->u, 1.0f); if (hs->u < 0) u += 1.0f;
        float v = fmodf(hs->v, 1.0f); if (hs->v < 0) v += 1.0f;
        int xi = (int)(u * (tx->w - 1)); if (tx->w - 1 < xi) xi = tx->w - 1;
        int yi = (int)((1 - v) * (tx->h - 1)); if (tx->h - 1 < yi) yi = tx->h - 1;
        if (xi < 0) xi = 0;   /* memory safety only: the reference indexes out of bounds here (NaN u/v) */
        if (yi < 0) yi = 0;
        const int* px = &tx->px[((size_t)yi * (size_t)tx->w + (size_t)xi) * 3];
        mat[0] = px[0] / 255.0f; mat[1] = px[1] / 255.0f; mat[2] = px[2] / 255.0f;
    } else { mat[0] = hs->m->diffuse[0]; mat[1] = hs->m->diffuse[1]; mat[2] = hs->m->diffuse[2]; }
    const float* N = hs->n;
    float V[3] = { cam->loc[0] - hs->p[0], cam->loc[1] - hs->p[1], cam->loc[2] - hs->p[2] };
    normalise(V);
    colour[0] = ka * mat[0]; colour[1] = ka * mat[1]; colour[2] = ka * mat[2];
    for (int li = 0; li < sc->nlights; ++li) {
        const Light* light = &sc->lights[li];
        float L[3] = { light->loc[0] - hs->p[0], light->loc[1] - hs->p[1], light->loc[2] - hs->p[2] };
        float L_len = (float)sqrt(pow((double)L[0], 2.0) + pow((double)L[1], 2.0) + pow((double)L[2], 2.0));
        normalise(L);
        float irr = (float)(light->intensity / (1.0f + pow((double)L_len, 2.0)));
        float H[3] = { L[0] + V[0], L[1] + V[1], L[2] + V[2] };
        normalise(H);
        float N_L = smax(0.0f, dotp(N, L));
        float N_H = smax(0.0f, dotp(N, H));
        float shadow;
        if (cfg->softShadows) shadow = soft_shadow(sc, hs->p, hs->time, light, cfg);
        else shadow = hard_shadow(sc, hs->p, hs->time, N, light, cfg);
        float pw = powf(N_H, shiny);
        colour[0] += shadow * mat[0] * N_L * irr + 0.5f * hs->m->specular[0] * pw * irr;
        colour[1] += shadow * mat[1] * N_L * irr + 0.5f * hs->m->specular[1] * pw * irr;
        colour[2] += shadow * mat[2] * N_L * irr + 0.5f * hs->m->specular[2] * pw * irr;
    }
    colour[0] = smin(1.0f, smax(colour[0], 0.0f));
    colour[1] = smin(1.0f, smax(colour[1], 0.0f));
    colour[2] = smin(1.0f, smax(colour[2], 0.0f));
}

static void reflect_refract(Scene* sc, Ray* ray, const Camera* cam, const orc_config* cfg, int depth, int maxDepth, float colour[3]) {
    /* raytracer.cpp:352-506 */
    const float EPS = 1e-4f;
    if (ray->nh == 0) { colour[0] = colour[1] = colour[2] = 0.0f; return; }
    Hit hs = ray->last;
    blinn_phong(sc, &hs, cam, cfg, colour);
    if (depth >= maxDepth) return;
    const float* spec = hs.m->specular;
    float reflectivity = (float)(((spec[0] + spec[1] + spec[2]) / 3.0f) * 0.5);
    reflectivity = smin(smax(reflectivity, 0.0f), 1.0f);
    float ior = hs.m->ior > 0.0f ? hs.m->ior : 1.0f;
    float transparency = hs.m->transparency;
    float V[3] = { cam->loc[0] - hs.p[0], cam->loc[1] - hs.p[1], cam->loc[2] - hs.p[2] };
    normalise(V);
    float N[3] = { hs.n[0], hs.n[1], hs.n[2] };
    float N_rayDir = dotp(ray->d, N);
    if (N_rayDir > 0.0f) { N[0] = -N[0]; N[1] = -N[1]; N[2] = -N[2]; }
    float reflectColour[3] = { 0.0f, 0.0f, 0.0f };
    if (reflectivity > 0) {
        float incoming_N = dotp(ray->d, N);
        for (int s = 0; s < cfg->GRsamples; ++s) {
            float rd[3] = { ray->d[0] - 2.0f * incoming_N * N[0], ray->d[1] - 2.0f * incoming_N * N[1], ray->d[2] - 2.0f * incoming_N * N[2] };
            normalise(rd);
            if (cfg->glossyReflect && hs.m->shininess > 0.0f) {
                float glossAngle = (float)(smax(0.01f, (1.0f - smin(1.0f, hs.m->shininess / 128.0f))) * (M_PI / 6.0f));
                float nd[3]; rnd_cone_direction(rd, glossAngle, nd);
                rd[0] = nd[0]; rd[1] = nd[1]; rd[2] = nd[2];
            }
            Ray rr; memset(&rr, 0, sizeof rr);
            rr.o[0] = hs.p[0] + N[0] * EPS; rr.o[1] = hs.p[1] + N[1] * EPS; rr.o[2] = hs.p[2] + N[2] * EPS;
            rr.d[0] = rd[0]; rr.d[1] = rd[1]; rr.d[2] = rd[2];
            float sample[3] = { 0.0f, 0.0f, 0.0f };
            float a = 0.0f, b = FLT_MAX;
            if (g_stats) g_stats->reflection++;
            if ((cfg->bvh && bvh_intersect(sc, sc->bvh, &rr, &a, &b, cfg)) || unaccelerated(sc, &rr, cfg))
                reflect_refract(sc, &rr, cam, cfg, depth + 1, maxDepth, sample);
            reflectColour[0] += sample[0]; reflectColour[1] += sample[1]; reflectColour[2] += sample[2];
        }
    }
    float invn = 1.0f / cfg->GRsamples;
    for (int k = 0; k < 3; ++k) {
        reflectColour[k] = reflectColour[k] * invn;
        reflectColour[k] = reflectColour[k] * reflectivity;
        colour[k] = colour[k] + reflectColour[k];
    }
    if (transparency > 0) {
        float n1 = 1.0f, n2 = ior;
        float cos_i = dotp(ray->d, N);
        if (cos_i > 0) { float tt = n1; n1 = n2; n2 = tt; }
        float eta = n1 / n2;
        cos_i = -smax(-1.0f, smin(1.0f, cos_i));
        float k = 1.0f - eta * eta * (1.0f - cos_i * cos_i);
        if (!(k < 0.0f)) {
            float sq = sqrtf(k);
            float fd[3] = { (ray->d[0] * eta) + (N[0] * (eta * cos_i - sq)),
                            (ray->d[1] * eta) + (N[1] * (eta * cos_i - sq)),
                            (ray->d[2] * eta) + (N[2] * (eta * cos_i - sq)) };
            normalise(fd);
            Ray fr; memset(&fr, 0, sizeof fr);
            fr.o[0] = hs.p[0] + fd[0] * EPS; fr.o[1] = hs.p[1] + fd[1] * EPS; fr.o[2] = hs.p[2] + fd[2] * EPS;
            fr.d[0] = fd[0]; fr.d[1] = fd[1]; fr.d[2] = fd[2];
            float rc[3] = { 0.0f, 0.0f, 0.0f };
            float a = 0.0f, b = FLT_MAX;
            if (g_stats) g_stats->refraction++;
            if ((cfg->bvh && bvh_intersect(sc, sc->bvh, &fr, &a, &b, cfg)) || unaccelerated(sc, &fr, cfg))
                reflect_refract(sc, &fr, cam, cfg, depth + 1, maxDepth, rc);
            float cosTheta = fabsf(dotp(V, N));
            float r0 = (float)pow((double)((n1 - n2) / (n1 + n2)), 2.0);
            float fresnel = (float)(r0 + (1.0f - r0) * pow((double)(1.0f - cosTheta), 5.0));
            float weight = transparency * (1.0f - fresnel);
            colour[0] = colour[0] + rc[0] * weight;
            colour[1] = colour[1] + rc[1] * weight;
            colour[2] = colour[2] + rc[2] * weight;
        }
    }
    colour[0] = smin(1.0f, smax(0.0f, colour[0]));
    colour[1] = smin(1.0f, smax(0.0f, colour[1]));
    colour[2] = smin(1.0f, smax(0.0f, colour[2]));
}

/* ------------------------------------------------------------------------------------------------
 * Pixel loop (raytracer.cpp:81-193)
 * ---------------------------------------------------------------------------------------------- */
int orc_render(const char* scene_text, const char* textures_dir, const orc_config* cfg, int rand_seed,
               int* rgb_out, long long rgb_cap, int* width, int* height, orc_stats* stats) {
    Scene sc;
    g_err[0] = 0;
    if (stats) memset(stats, 0, sizeof *stats);
    if (parse_scene(scene_text, cfg, textures_dir, &sc)) return -1;
    if (sc.ncams < 1) { set_err("no camera"); free_scene(&sc); return -2; }
    if (rand_seed >= 0) srand((unsigned)rand_seed);
    const Camera* cam = &sc.cams[0];
    int W = (int)cam->res_x, H = (int)cam->res_y;
    if (width) *width = W;
    if (height) *height = H;
    if ((long long)W * H * 3 > rgb_cap) { set_err("rgb_out too small: need %lld ints", (long long)W * H * 3); free_scene(&sc); return -3; }
    if (cfg->bvh) scene_build_bvh(&sc);
    int AA = cfg->antiAliasing ? cfg->AAsamples : 1;
    int DOF = cfg->dof ? cfg->dofSamples : 1;
    int MB = cfg->motionBlur ? cfg->mbSamples : 1;
    g_stats = stats;
    struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
    /* the loop bounds compare an int against the float resolution (raytracer.cpp:111-112) */
    for (int y = 0; y < cam->res_y; ++y) {
        for (int x = 0; x < cam->res_x; ++x) {
            if (x < g_win[0] || x >= g_win[2] || y < g_win[1] || y >= g_win[3]) continue;
            if ((x - g_win[0]) % g_stride[0] != 0 || (y - g_win[1]) % g_stride[1] != 0) continue;
            float pc[3] = { 0.0f, 0.0f, 0.0f };
            float u, v;
            for (int s = 0; s < AA; ++s) {
                if (cfg->antiAliasing) { u = x + rnd01(); v = y + rnd01(); }
                else { u = (float)x; v = (float)y; }
                for (int t = 0; t < DOF; ++t) {
                    Ray ray; camera_ray(cam, u, v, &ray);
                    for (int m = 0; m < MB; ++m) {
                        ray.nh = 0;
                        ray.time = (m + rnd01()) / MB;
                        if (stats) stats->primary++;
                        int hit;
                        if (cfg->bvh) { float a = 0.0f, b = FLT_MAX; hit = bvh_intersect(&sc, sc.bvh, &ray, &a, &b, cfg); }
                        else hit = unaccelerated(&sc, &ray, cfg);
                        if (hit) {
                            float col[3];
                            reflect_refract(&sc, &ray, cam, cfg, 0, cfg->reflectRefractDepth, col);
                            pc[0] = pc[0] + col[0]; pc[1] = pc[1] + col[1]; pc[2] = pc[2] + col[2];
                        }
                    }
                }
            }
            float inv = 1.0f / (AA * DOF * MB);
            pc[0] = pc[0] * inv; pc[1] = pc[1] * inv; pc[2] = pc[2] * inv;
            int* o = &rgb_out[((size_t)y * (size_t)W + (size_t)x) * 3];
            o[0] = (int)(255 * pc[0]); o[1] = (int)(255 * pc[1]); o[2] = (int)(255 * pc[2]);
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (stats) stats->seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    g_stats = NULL;
    free_scene(&sc);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Known-answer helpers
 * ---------------------------------------------------------------------------------------------- */
int orc_camera_ray(const char* scene_text, float x, float y, float* out6) {
    orc_config cfg; orc_default_config(&cfg);
    Scene sc;
    if (parse_scene(scene_text, &cfg, NULL, &sc)) return -1;
    if (sc.ncams < 1) { free_scene(&sc); set_err("no camera"); return -2; }
    Ray r; camera_ray(&sc.cams[0], x, y, &r);
    for (int i = 0; i < 3; ++i) { out6[i] = r.o[i]; out6[3 + i] = r.d[i]; }
    free_scene(&sc);
    return 0;
}

/* Known-answer replay: one record of oracle/ref_kat.cpp (same input and output layout) through the functions
 * above.  kind 0 sphere, 1 cube, 2 plane (a fresh object, i.e. what a BVH leaf copy is), 3 box slab test. */
static void kat_hit(int hit, const Hit* h, int tex, float* out) {
    out[0] = hit ? 1.0f : 0.0f;
    for (int i = 1; i < 10; ++i) out[i] = 0.0f;
    if (!hit) return;
    out[1] = h->t;
    for (int i = 0; i < 3; ++i) { out[2 + i] = h->p[i]; out[5 + i] = h->n[i]; }
    if (tex) { out[8] = h->u; out[9] = h->v; }
}
int orc_kat(int kind, const float* in, float* out) {
    Ray r; memset(&r, 0, sizeof r);
    for (int i = 0; i < 3; ++i) { r.o[i] = in[i]; r.d[i] = in[3 + i]; }
    r.time = in[6];
    const float* a = in + 7;
    orc_config cfg; orc_default_config(&cfg);
    Hit h; hit_reset(&h);
    if (kind == 0) {
        Sphere s; memset(&s, 0, sizeof s);
        for (int i = 0; i < 3; ++i) { s.start[i] = a[i]; s.end[i] = a[3 + i]; }
        s.radius = a[6]; cfg.motionBlur = a[7] != 0.0f; s.m.hasTex = a[8] != 0.0f; cfg.textures = s.m.hasTex;
        kat_hit(sphere_intersect(&s, &r, &h, &cfg), &h, s.m.hasTex, out);
        sphere_aabb(&s, out + 10, out + 13);
        return 16;
    }
    if (kind == 1) {
        Cube c; memset(&c, 0, sizeof c);
        for (int i = 0; i < 3; ++i) { c.start[i] = a[i]; c.end[i] = a[3 + i]; c.rot[i] = a[6 + i]; }
        c.scale = a[9]; cfg.motionBlur = a[10] != 0.0f; c.m.hasTex = a[11] != 0.0f; cfg.textures = c.m.hasTex;
        kat_hit(cube_intersect(&c, &r, &h, &cfg), &h, c.m.hasTex, out);
        cube_aabb(&c, out + 10, out + 13);
        return 16;
    }
    if (kind == 2) {
        float vf[4][3], vs[4][3];
        Plane p; memset(&p, 0, sizeof p);
        p.n = 4; p.vfile = vf; p.vsorted = vs;
        for (int k = 0; k < 4; ++k) for (int i = 0; i < 3; ++i) vf[k][i] = a[3 * k + i];
        p.m.hasTex = a[12] != 0.0f; cfg.textures = p.m.hasTex;
        plane_calc_normal(&p);
        plane_sort_winding(&p);
        kat_hit(plane_intersect_copy(&p, &r, &h), &h, p.m.hasTex, out);
        for (int i = 0; i < 3; ++i) out[10 + i] = p.normal[i];
        plane_aabb(&p, out + 13, out + 16);
        return 19;
    }
    if (kind == 3) {
        float tmin = 0.0f, tmax = 0.0f;
        int hit = aabb_intersect(a, a + 3, &r, &tmin, &tmax);
        out[0] = hit ? 1.0f : 0.0f; out[1] = tmin; out[2] = tmax;
        return 3;
    }
    return -1;
}

static int dump_node(const Node* n, int depth, float* out, int cap, int idx) {
    if (!n) return idx;
    if (idx < cap) {
        float* o = out + (size_t)idx * 12;
        for (int i = 0; i < 3; ++i) { o[i] = n->mn[i]; o[3 + i] = n->mx[i]; }
        o[6] = n->left ? 1.0f : 0.0f; o[7] = n->right ? 1.0f : 0.0f;
        o[8] = (float)n->np; o[9] = (float)n->nc; o[10] = (float)n->ns; o[11] = (float)depth;
    }
    idx++;
    idx = dump_node(n->left, depth + 1, out, cap, idx);
    idx = dump_node(n->right, depth + 1, out, cap, idx);
    return idx;
}

int orc_bvh_dump(const char* scene_text, const orc_config* c, float* out, int cap_nodes) {
    Scene sc;
    orc_config cfg = *c; cfg.textures = 0;
    if (parse_scene(scene_text, &cfg, NULL, &sc)) return -1;
    scene_build_bvh(&sc);
    int n = dump_node(sc.bvh, 0, out, cap_nodes, 0);
    free_scene(&sc);
    return n;
}
