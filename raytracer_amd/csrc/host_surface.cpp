// Host surface of the renderer: flags, scene.json, host BVH build, PPM I/O.  No GPU code here.
//
// Behaviour follows the reference's host code (cited per function, paths relative to its Code/ directory);
// the implementation does not: one pass over the scene text with string_views, shape boxes computed once,
// and an index-partitioning BVH build instead of copying shape vectors per level.
#include "rt_host.h"
#include "rt_parallel.h"

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>

namespace {

using sv = std::string_view;

struct ParseError : std::runtime_error { using std::runtime_error::runtime_error; };

// ---------------------------------------------------------------------------------------------------
// Scene-text scanner.  Not a JSON parser, on purpose: the reference finds keys by first substring match and
// cuts "objects" with a brace counter that starts one too high, and files written for it rely on that.
// ---------------------------------------------------------------------------------------------------
struct Scan {
    // camera.cpp:230-256 / shape.cpp:77-108: slice from the '{' after `key` to the close of the ENCLOSING
    // object (the counter starts at 1 and counts the opening brace again).
    static sv object(sv s, sv key) {
        size_t pos = s.find(key);
        if (pos == sv::npos) throw ParseError("Key not found in JSON string: " + std::string(key));
        size_t start = s.find('{', pos);
        if (start == sv::npos) throw ParseError("Invalid JSON string");
        int depth = 1; size_t end = start;
        for (; end < s.size(); ++end) {
            if (s[end] == '{') ++depth;
            else if (s[end] == '}' && --depth == 0) break;
        }
        return s.substr(start, end - start + 1);
    }
    // camera.cpp:260-284: '[' ... matching ']' inclusive.
    static sv array(sv s, sv key) {
        size_t pos = s.find(key);
        if (pos == sv::npos) throw ParseError("Key not found in JSON string: " + std::string(key));
        size_t start = s.find('[', pos);
        if (start == sv::npos) throw ParseError("Invalid JSON string: '[' not found");
        int depth = 1; size_t end = start + 1;
        while (end < s.size() && depth > 0) { if (s[end] == '[') ++depth; else if (s[end] == ']') --depth; ++end; }
        if (depth != 0) throw ParseError("Mismatched brackets in JSON string");
        return s.substr(start, end - start);
    }
    // camera.cpp:33-52: top-level {...} items of an array slice.
    static std::vector<sv> items(sv arr) {
        if (arr.size() >= 2 && arr.front() == '[' && arr.back() == ']') arr = arr.substr(1, arr.size() - 2);
        std::vector<sv> out; int depth = 0; size_t start = 0;
        for (size_t i = 0; i < arr.size(); ++i) {
            if (arr[i] == '{') { if (depth++ == 0) start = i; }
            else if (arr[i] == '}') { if (--depth == 0) out.push_back(arr.substr(start, i - start + 1)); }
        }
        return out;
    }
    // camera.cpp:205-227, shape.cpp:14-37 (exponent allowed) and pointlight.cpp:80-102 (not allowed):
    // the longest run of [0-9 . - e] after the colon, handed to std::stof.
    static float number(sv s, sv key, bool allow_exponent) {
        size_t pos = s.find(key);
        if (pos == sv::npos) throw ParseError("Key not found in JSON string: " + std::string(key));
        size_t b = s.find(':', pos);
        if (b == sv::npos) throw ParseError("Invalid JSON string");
        ++b;
        while (b < s.size() && std::isspace((unsigned char)s[b])) ++b;
        size_t e = b;
        auto at = [&](size_t i) { return i < s.size() ? s[i] : '\0'; };
        while ((e < s.size() && std::isdigit((unsigned char)s[e])) || at(e) == '.' || at(e) == '-' || (allow_exponent && at(e) == 'e')) ++e;
        std::string tok(s.substr(b, e - b));
        char* endp = nullptr; errno = 0;
        float v = std::strtof(tok.c_str(), &endp);
        if (endp == tok.c_str()) throw ParseError("stof: no number after key " + std::string(key));
        if (errno == ERANGE) throw ParseError("stof: out of range after key " + std::string(key));
        return v;
    }
    // shape.cpp:40-74: "..." -> contents, null / anything else -> "".
    static std::string string(sv s, sv key) {
        size_t pos = s.find(key);
        if (pos == sv::npos) throw ParseError("Key not found in JSON string: " + std::string(key));
        size_t b = s.find(':', pos);
        if (b == sv::npos) throw ParseError("Invalid JSON string");
        b = s.find_first_not_of(" \t\n\r", b + 1);
        if (b == sv::npos) return "";
        if (s.compare(b, 4, "null") == 0) return "";
        if (s[b] == '"') {
            size_t e = s.find('"', b + 1);
            if (e == sv::npos) throw ParseError("Unterminated string value in JSON");
            return std::string(s.substr(b + 1, e - b - 1));
        }
        return "";
    }
    static void xyz(sv s, sv key, bool allow_exponent, float out[3], sv a = "\"x\"", sv b = "\"y\"", sv c = "\"z\"") {
        sv o = object(s, key);
        out[0] = number(o, a, allow_exponent); out[1] = number(o, b, allow_exponent); out[2] = number(o, c, allow_exponent);
    }
};

// ---------------------------------------------------------------------------------------------------
// PPM reader (image.cpp:13-67)
// ---------------------------------------------------------------------------------------------------
struct Ppm { int w = 0, h = 0, maxv = 0; std::vector<int32_t> rgb; };

Ppm read_ppm_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) throw std::runtime_error("Failed to open the image file " + path);
    std::string data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    size_t i = 0; const size_t n = data.size();
    auto skip_ws = [&] { while (i < n && std::isspace((unsigned char)data[i])) ++i; };
    auto token = [&] { skip_ws(); size_t b = i; while (i < n && !std::isspace((unsigned char)data[i])) ++i; return data.substr(b, i - b); };
    auto integer = [&](const char* what) {
        skip_ws(); size_t b = i; bool neg = false;
        if (i < n && (data[i] == '-' || data[i] == '+')) { neg = data[i] == '-'; ++i; }
        long v = 0; size_t d0 = i;
        while (i < n && std::isdigit((unsigned char)data[i])) { v = v * 10 + (data[i] - '0'); ++i; }
        if (i == d0) { i = b; throw std::runtime_error(what); }
        return (int)(neg ? -v : v);
    };
    std::string magic = token();
    if (magic != "P3" && magic != "P6") throw std::runtime_error("Only P3 and P6 .ppm files are supported");
    Ppm img;
    img.w = integer("bad PPM header"); img.h = integer("bad PPM header"); img.maxv = integer("bad PPM header");
    if (img.w <= 0 || img.h <= 0) throw std::runtime_error("bad PPM size");
    size_t count = (size_t)img.w * (size_t)img.h * 3;
    img.rgb.resize(count);
    if (magic == "P3") {
        for (size_t k = 0; k < count; ++k) img.rgb[k] = integer("Invalid pixel values");
    } else {
        // The reference starts reading bytes right after the maxval digits, so the separator becomes the first
        // red value and every later channel is taken from one byte earlier than the format says.
        if (n - i < count) throw std::runtime_error("Invalid pixel values");
        for (size_t k = 0; k < count; ++k) img.rgb[k] = (unsigned char)data[i + k];
    }
    return img;
}

// ---------------------------------------------------------------------------------------------------
// Scene storage
// ---------------------------------------------------------------------------------------------------
struct SceneStore {
    std::vector<rt_light> lights;
    std::vector<rt_sphere> spheres;
    std::vector<rt_cube> cubes;
    std::vector<rt_plane> planes;
    std::vector<float> verts;
    std::vector<std::string> tex_names;
    std::vector<Ppm> tex_data;
    std::vector<rt_texture> textures;
};

int texture_id(SceneStore& st, const std::string& dir, const std::string& name) {
    for (size_t i = 0; i < st.tex_names.size(); ++i) if (st.tex_names[i] == name) return (int)i;
    st.tex_data.push_back(read_ppm_file(dir + "/" + name));
    st.tex_names.push_back(name);
    return (int)st.tex_names.size() - 1;
}

rt_material parse_material(sv obj, const rt_config& cfg, SceneStore& st, const std::string& texdir, const char* kind, size_t idx) {
    rt_material m{};
    sv ms = Scan::object(obj, "\"material\"");
    Scan::xyz(ms, "\"diffuse\"", true, m.diffuse, "\"r\"", "\"g\"", "\"b\"");
    Scan::xyz(ms, "\"specular\"", true, m.specular, "\"r\"", "\"g\"", "\"b\"");
    m.shininess = Scan::number(ms, "\"shininess\"", true);
    m.transparency = Scan::number(ms, "\"transparency\"", true);
    m.ior = Scan::number(ms, "\"ior\"", true);
    std::string tex = Scan::string(ms, "\"texture\"");
    m.tex_id = -1;
    if (!tex.empty() && cfg.textures) {                       // sphere.cpp:105-109, cube.cpp:112-116, plane.cpp:130-134
        std::printf("%s %zu texture: %s\n", kind, idx, tex.c_str());
        m.tex_id = texture_id(st, texdir, tex);
    }
    return m;
}

// plane.cpp:304-321
void plane_normal(const float* v, float n[3]) {
    float a[3] = { v[3] - v[0], v[4] - v[1], v[5] - v[2] };
    float b[3] = { v[6] - v[0], v[7] - v[1], v[8] - v[2] };
    n[0] = a[1] * b[2] - a[2] * b[1];
    n[1] = a[2] * b[0] - a[0] * b[2];
    n[2] = a[0] * b[1] - a[1] * b[0];
    float len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    n[0] /= len; n[1] /= len; n[2] /= len;
}

rt_host_scene* parse_scene_impl(sv text, const std::string& texdir, const rt_config& cfg) {
    auto st = std::make_unique<SceneStore>();
    auto hs = std::make_unique<rt_host_scene>();
    std::memset(hs.get(), 0, sizeof(rt_host_scene));
    sv props = Scan::object(text, "\"properties\"");

    // cameras: camera.cpp:30-104
    {
        auto cams = Scan::items(Scan::array(props, "\"cameras\""));
        if (cams.empty()) { std::fprintf(stderr, "No cameras.\n"); throw ParseError("scene has no camera"); }
        hs->n_cameras = (int)cams.size();
        sv o = cams[0];
        rt_camera& c = hs->scene.camera;
        Scan::xyz(o, "\"location\"", true, c.location);
        Scan::xyz(o, "\"gaze_vector\"", true, c.gaze);
        if (cfg.dof) {
            c.aperture = Scan::number(o, "\"aperture\"", true);
            c.focal_distance = Scan::number(o, "\"focal_distance\"", true);
        } else { c.aperture = 0.0f; c.focal_distance = 0.0f; }
        c.focal_length = Scan::number(o, "\"focal_length\"", true) / 1000.0f;
        sv sensor = Scan::object(o, "\"sensor\"");
        c.sensor_width = Scan::number(sensor, "\"width\"", true) / 1000.0f;
        c.sensor_height = Scan::number(sensor, "\"height\"", true) / 1000.0f;
        sv film = Scan::object(o, "\"film_resolution\"");
        c.res_x = Scan::number(film, "\"width\"", true);
        c.res_y = Scan::number(film, "\"height\"", true);
    }
    // lights: pointlight.cpp:25-75
    {
        auto items = Scan::items(Scan::array(props, "\"point_lights\""));
        if (items.empty()) std::fprintf(stderr, "No lights.\n");
        for (sv o : items) {
            rt_light l{};
            Scan::xyz(o, "\"location\"", false, l.location);
            l.rad_intensity = Scan::number(o, "\"radiant_intensity\"", false);
            st->lights.push_back(l);
        }
    }
    // spheres: sphere.cpp:33-112
    {
        auto items = Scan::items(Scan::array(props, "\"spheres\""));
        if (items.empty()) std::fprintf(stderr, "No spheres.\n");
        for (size_t i = 0; i < items.size(); ++i) {
            rt_sphere s{};
            Scan::xyz(items[i], "\"start_location\"", true, s.start);
            Scan::xyz(items[i], "\"end_location\"", true, s.end);
            s.radius = Scan::number(items[i], "\"radius\"", true);
            s.mat = parse_material(items[i], cfg, *st, texdir, "Sphere", i);
            st->spheres.push_back(s);
        }
    }
    // cubes: cube.cpp:34-119
    {
        auto items = Scan::items(Scan::array(props, "\"cubes\""));
        if (items.empty()) std::fprintf(stderr, "No cubes.\n");
        for (size_t i = 0; i < items.size(); ++i) {
            rt_cube c{};
            Scan::xyz(items[i], "\"start_location\"", true, c.start);
            Scan::xyz(items[i], "\"end_location\"", true, c.end);
            Scan::xyz(items[i], "\"rotation\"", true, c.rotation);
            c.scale = Scan::number(items[i], "\"scale\"", true);
            c.mat = parse_material(items[i], cfg, *st, texdir, "Cube", i);
            st->cubes.push_back(c);
        }
    }
    // planes: plane.cpp:34-137
    {
        auto items = Scan::items(Scan::array(props, "\"planes\""));
        if (items.empty()) std::fprintf(stderr, "No planes.\n");
        for (size_t i = 0; i < items.size(); ++i) {
            rt_plane p{};
            auto corners = Scan::items(Scan::array(items[i], "\"corners\""));
            if (corners.size() < 3) throw ParseError("plane with fewer than 3 corners");
            p.n_verts = (int)corners.size();
            p.first_vert = (int)(st->verts.size() / 3);
            for (sv c : corners) {
                st->verts.push_back(Scan::number(c, "\"x\"", true));
                st->verts.push_back(Scan::number(c, "\"y\"", true));
                st->verts.push_back(Scan::number(c, "\"z\"", true));
            }
            plane_normal(&st->verts[(size_t)p.first_vert * 3], p.normal);
            p.mat = parse_material(items[i], cfg, *st, texdir, "Plane", i);
            st->planes.push_back(p);
        }
    }
    for (auto& t : st->tex_data) st->textures.push_back(rt_texture{ t.w, t.h, t.rgb.data() });

    rt_scene& sc = hs->scene;
    sc.n_lights = (int)st->lights.size();   sc.lights = st->lights.data();
    sc.n_spheres = (int)st->spheres.size(); sc.spheres = st->spheres.data();
    sc.n_cubes = (int)st->cubes.size();     sc.cubes = st->cubes.data();
    sc.n_planes = (int)st->planes.size();   sc.planes = st->planes.data();
    sc.n_plane_verts = (int)(st->verts.size() / 3); sc.plane_verts = st->verts.data();
    sc.n_textures = (int)st->textures.size(); sc.textures = st->textures.data();
    hs->storage = st.release();
    return hs.release();
}

// ---------------------------------------------------------------------------------------------------
// Shape boxes and the BVH build
// ---------------------------------------------------------------------------------------------------
struct Box { float mn[3], mx[3]; };

inline void box_empty(Box& b) { for (int i = 0; i < 3; ++i) { b.mn[i] = INFINITY; b.mx[i] = -INFINITY; } }
inline void box_add(Box& b, const Box& o) {              // aabb.cpp:71-108: strict compares
    for (int i = 0; i < 3; ++i) { if (o.mn[i] < b.mn[i]) b.mn[i] = o.mn[i]; if (o.mx[i] > b.mx[i]) b.mx[i] = o.mx[i]; }
}

// cube.cpp:297-318 (float sin/cos of the float angles)
inline void rotate_xyz(float v[3], const float rot[3]) {
    float x = v[0], y = v[1], z = v[2];
    float cx = std::cos(rot[0]), sx = std::sin(rot[0]);
    float y1 = y * cx - z * sx, z1 = y * sx + z * cx; y = y1; z = z1;
    float cy = std::cos(rot[1]), sy = std::sin(rot[1]);
    float x2 = x * cy + z * sy, z2 = -x * sy + z * cy; x = x2; z = z2;
    float cz = std::cos(rot[2]), sz = std::sin(rot[2]);
    v[0] = x * cz - y * sz; v[1] = x * sz + y * cz; v[2] = z;
}

Box sphere_box(const rt_sphere& s) {                       // sphere.cpp:177-188 via AABB::unionOf, aabb.cpp:135-156
    Box b; box_empty(b);
    for (int i = 0; i < 3; ++i) {
        float a0 = s.start[i] - s.radius, a1 = s.start[i] + s.radius, e0 = s.end[i] - s.radius, e1 = s.end[i] + s.radius;
        if (a0 < b.mn[i]) b.mn[i] = a0;
        if (e0 < b.mn[i]) b.mn[i] = e0;
        if (a1 > b.mx[i]) b.mx[i] = a1;
        if (e1 > b.mx[i]) b.mx[i] = e1;
    }
    return b;
}

Box cube_box(const rt_cube& c) {                           // cube.cpp:354-386
    Box b; box_empty(b);
    // cube.cpp:297-318 evaluates these six float sines/cosines once per corner; they do not depend on the corner
    const float cx = std::cos(c.rotation[0]), sx = std::sin(c.rotation[0]), cy = std::cos(c.rotation[1]), sy = std::sin(c.rotation[1]);
    const float cz = std::cos(c.rotation[2]), sz = std::sin(c.rotation[2]);
    const float sg[2] = { -1.0f, 1.0f };
    for (float x : sg) for (float y : sg) for (float z : sg) {
        float y1 = y * cx - z * sx, z1 = y * sx + z * cx;
        float x2 = x * cy + z1 * sy, z2 = -x * sy + z1 * cy;
        float r[3] = { x2 * cz - y1 * sz, x2 * sz + y1 * cz, z2 };
        float p[3] = { r[0] * c.scale + c.start[0], r[1] * c.scale + c.start[1], r[2] * c.scale + c.start[2] };
        // The end-position corner is the rotated UNIT corner plus end_location: the reference stores the scaled
        // value into the wrong variable (cube.cpp:378), and ray/box agreement depends on reproducing that.
        float q[3] = { r[0] + c.end[0], r[1] + c.end[1], r[2] + c.end[2] };
        for (int i = 0; i < 3; ++i) {
            if (p[i] < b.mn[i]) b.mn[i] = p[i];
            if (p[i] > b.mx[i]) b.mx[i] = p[i];
            if (q[i] < b.mn[i]) b.mn[i] = q[i];
            if (q[i] > b.mx[i]) b.mx[i] = q[i];
        }
    }
    return b;
}

Box plane_box(const rt_plane& p, const float* verts) {     // plane.cpp:324-340: max starts at FLT_MIN (smallest positive)
    Box b; for (int i = 0; i < 3; ++i) { b.mn[i] = FLT_MAX; b.mx[i] = FLT_MIN; }
    for (int k = 0; k < p.n_verts; ++k) {
        const float* v = verts + (size_t)(p.first_vert + k) * 3;
        for (int i = 0; i < 3; ++i) { if (v[i] < b.mn[i]) b.mn[i] = v[i]; if (v[i] > b.mx[i]) b.mx[i] = v[i]; }
    }
    return b;
}

struct Builder {
    const std::vector<Box>& pbox; const std::vector<Box>& cbox; const std::vector<Box>& sbox;
    std::vector<rt_bvh_node> nodes;
    std::vector<int32_t> prim_ids;

    // One level of BVHNode::buildBVH (bvh.cpp:108-172).  pl/cu/sp hold primitive indices in the order the
    // reference's vectors would (stable partition keeps file order inside every node).
    int build(std::vector<int32_t>& pl, std::vector<int32_t>& cu, std::vector<int32_t>& sp, int depth_left) {
        Box box; box_empty(box);                            // AABB::fromPoints(planes, cubes, spheres), aabb.cpp:67-111
        for (int i : pl) box_add(box, pbox[i]);
        for (int i : cu) box_add(box, cbox[i]);
        for (int i : sp) box_add(box, sbox[i]);
        int me = (int)nodes.size();
        nodes.push_back(rt_bvh_node{});
        {
            rt_bvh_node& n = nodes[me];
            for (int i = 0; i < 3; ++i) { n.box_min[i] = box.mn[i]; n.box_max[i] = box.mx[i]; }
            n.left = n.right = -1; n.first_prim = 0; n.n_planes = n.n_cubes = n.n_spheres = 0;
        }
        if (depth_left == 0) {                              // bvh.cpp:109-117
            rt_bvh_node& n = nodes[me];
            n.first_prim = (int)prim_ids.size();
            n.n_planes = (int)pl.size(); n.n_cubes = (int)cu.size(); n.n_spheres = (int)sp.size();
            prim_ids.insert(prim_ids.end(), pl.begin(), pl.end());
            prim_ids.insert(prim_ids.end(), cu.begin(), cu.end());
            prim_ids.insert(prim_ids.end(), sp.begin(), sp.end());
            return me;
        }
        int axis = 0;                                       // bvh.cpp:123-129
        if (box.mx[1] - box.mn[1] > box.mx[0] - box.mn[0]) axis = 1;
        if (box.mx[2] - box.mn[2] > box.mx[0] - box.mn[0] && box.mx[2] - box.mn[2] > box.mx[1] - box.mn[1]) axis = 2;
        float split = (box.mn[axis] + box.mx[axis]) / 2.0f; // bvh.cpp:139
        auto part = [&](std::vector<int32_t>& v, const std::vector<Box>& boxes, std::vector<int32_t>& l, std::vector<int32_t>& r) {
            for (int i : v) { if ((boxes[i].mn[axis] + boxes[i].mx[axis]) / 2.0f < split) l.push_back(i); else r.push_back(i); }
            std::vector<int32_t>().swap(v);                 // parent lists are not needed below this point
        };
        std::vector<int32_t> lp, rp, lc, rc, ls, rs;
        part(pl, pbox, lp, rp); part(cu, cbox, lc, rc); part(sp, sbox, ls, rs);
        if (!lp.empty() || !lc.empty() || !ls.empty()) { int c = build(lp, lc, ls, depth_left - 1); nodes[me].left = c; }
        if (!rp.empty() || !rc.empty() || !rs.empty()) { int c = build(rp, rc, rs, depth_left - 1); nodes[me].right = c; }
        return me;
    }
};

// The same tree, built level by level over all primitives at once (for scenes of 10^5..10^7 primitives): a primitive
// carries the path of its node (one bit per level, 1 = right), every level bounds its nodes (per-worker tables merged
// in primitive order, so the strict compares of aabb.cpp:71-108 see the objects in the reference's order), picks axis
// and split point per node and extends the paths.  Leaves come out in path order = the recursion's left-to-right
// order, primitives inside a leaf in file order (a stable partition, like the reference's push_backs).
struct LevelBuilder {
    const std::vector<Box>& pbox; const std::vector<Box>& cbox; const std::vector<Box>& sbox;
    std::vector<rt_bvh_node> nodes;
    std::vector<int32_t> prim_ids;

    void run(int depth) {
        const std::vector<Box>* cls[3] = { &pbox, &cbox, &sbox };
        const size_t n[3] = { pbox.size(), cbox.size(), sbox.size() };
        const int workers = rt_worker_count();
        std::vector<uint16_t> path[3];
        for (int k = 0; k < 3; ++k) path[k].assign(n[k], 0);
        std::vector<std::vector<Box>> level_box((size_t)depth + 1);       // [level][path]
        std::vector<std::vector<uint32_t>> level_cnt((size_t)depth + 1);
        std::vector<Box> tb; std::vector<uint32_t> tc;
        for (int level = 0; level <= depth; ++level) {
            const size_t width = (size_t)1 << level;
            std::vector<Box>& nb = level_box[(size_t)level]; nb.resize(width);
            std::vector<uint32_t>& cnt = level_cnt[(size_t)level]; cnt.assign(width, 0);
            for (Box& b : nb) box_empty(b);
            for (int k = 0; k < 3; ++k) {
                tb.resize(width * (size_t)workers); tc.assign(width * (size_t)workers, 0);
                for (Box& b : tb) box_empty(b);
                const std::vector<Box>& bx = *cls[k]; const std::vector<uint16_t>& pa = path[k];
                rt_parallel_for(n[k], 8192, [&](size_t lo, size_t hi, int w) {
                    Box* my = &tb[(size_t)w * width]; uint32_t* mc = &tc[(size_t)w * width];
                    for (size_t i = lo; i < hi; ++i) { box_add(my[pa[i]], bx[i]); ++mc[pa[i]]; }
                });
                for (int w = 0; w < workers; ++w)
                    for (size_t q = 0; q < width; ++q)
                        if (tc[(size_t)w * width + q]) { box_add(nb[q], tb[(size_t)w * width + q]); cnt[q] += tc[(size_t)w * width + q]; }
            }
            if (level == depth) break;
            std::vector<int> axis(width, 0); std::vector<float> split(width, 0.0f);
            for (size_t q = 0; q < width; ++q) {
                const Box& box = nb[q];
                int a = 0;                                      // bvh.cpp:123-129
                if (box.mx[1] - box.mn[1] > box.mx[0] - box.mn[0]) a = 1;
                if (box.mx[2] - box.mn[2] > box.mx[0] - box.mn[0] && box.mx[2] - box.mn[2] > box.mx[1] - box.mn[1]) a = 2;
                axis[q] = a; split[q] = (box.mn[a] + box.mx[a]) / 2.0f;   // bvh.cpp:139
            }
            for (int k = 0; k < 3; ++k) {
                const std::vector<Box>& bx = *cls[k]; std::vector<uint16_t>& pa = path[k];
                rt_parallel_for(n[k], 8192, [&](size_t lo, size_t hi, int) {
                    for (size_t i = lo; i < hi; ++i) {
                        const int a = axis[pa[i]];
                        const bool left = (bx[i].mn[a] + bx[i].mx[a]) / 2.0f < split[pa[i]];
                        pa[i] = (uint16_t)(pa[i] * 2 + (left ? 0 : 1));
                    }
                });
            }
        }
        // leaves: stable counting sort of every class by path
        const size_t leaves = (size_t)1 << depth;
        const std::vector<uint32_t>& leaf_cnt = level_cnt[(size_t)depth];
        std::vector<size_t> first(leaves + 1, 0);
        for (size_t q = 0; q < leaves; ++q) first[q + 1] = first[q] + leaf_cnt[q];
        prim_ids.assign(first[leaves], 0);
        std::vector<uint32_t> class_cnt[3];
        std::vector<size_t> cursor(leaves);
        for (size_t q = 0; q < leaves; ++q) cursor[q] = first[q];
        for (int k = 0; k < 3; ++k) {
            // per-worker counts -> per-worker start offsets inside every leaf
            const std::vector<uint16_t>& pa = path[k];
            std::vector<size_t> off(leaves * (size_t)workers, 0);
            const size_t chunk_workers = (size_t)workers;
            rt_parallel_for(n[k], 8192, [&](size_t lo, size_t hi, int w) {
                size_t* my = &off[(size_t)w * leaves];
                for (size_t i = lo; i < hi; ++i) ++my[pa[i]];
            });
            class_cnt[k].assign(leaves, 0);
            for (size_t q = 0; q < leaves; ++q) {
                size_t at = cursor[q];
                for (size_t w = 0; w < chunk_workers; ++w) { size_t c = off[w * leaves + q]; off[w * leaves + q] = at; at += c; }
                class_cnt[k][q] = (uint32_t)(at - cursor[q]);
                cursor[q] = at;
            }
            rt_parallel_for(n[k], 8192, [&](size_t lo, size_t hi, int w) {
                size_t* my = &off[(size_t)w * leaves];
                for (size_t i = lo; i < hi; ++i) prim_ids[my[pa[i]]++] = (int32_t)i;
            });
        }
        // nodes in the recursion's order (node, left subtree, right subtree); a child exists when it owns a primitive
        struct Todo { int level; size_t q; int parent; bool right; };
        std::vector<Todo> st; st.push_back({ 0, 0, -1, false });
        while (!st.empty()) {
            Todo t = st.back(); st.pop_back();
            const int me = (int)nodes.size();
            rt_bvh_node nd{};
            const Box& box = level_box[(size_t)t.level][t.q];
            for (int i = 0; i < 3; ++i) { nd.box_min[i] = box.mn[i]; nd.box_max[i] = box.mx[i]; }
            nd.left = nd.right = -1;
            if (t.level == depth) {
                nd.first_prim = (int32_t)first[t.q];
                nd.n_planes = (int32_t)class_cnt[0][t.q]; nd.n_cubes = (int32_t)class_cnt[1][t.q]; nd.n_spheres = (int32_t)class_cnt[2][t.q];
            }
            nodes.push_back(nd);
            if (t.parent >= 0) { if (t.right) nodes[(size_t)t.parent].right = me; else nodes[(size_t)t.parent].left = me; }
            if (t.level < depth) {
                const std::vector<uint32_t>& cc = level_cnt[(size_t)t.level + 1];
                if (cc[2 * t.q + 1]) st.push_back({ t.level + 1, 2 * t.q + 1, me, true });    // popped after the left subtree
                if (cc[2 * t.q]) st.push_back({ t.level + 1, 2 * t.q, me, false });
            }
        }
    }
};

void copy_err(char* err, int cap, const char* msg) { if (err && cap > 0) std::snprintf(err, (size_t)cap, "%s", msg); }

} // namespace

// ===================================================================================================
// C ABI
// ===================================================================================================
extern "C" {

void rt_default_config(rt_config* c) {                      // config.h:8-27
    std::memset(c, 0, sizeof *c);
    c->soft_shadows = 0; c->ss_samples = 4; c->glossy_reflect = 0; c->gr_samples = 16; c->light_radius = 0.25f;
    c->anti_aliasing = 0; c->aa_samples = 4; c->textures = 0; c->bvh = 1; c->reflect_refract_depth = 0;
    c->dof = 0; c->dof_samples = 16; c->motion_blur = 0; c->mb_samples = 16;
    c->seed = 1; c->fallback = 0;
}

int rt_host_parse_args(int argc, const char* const* argv, int extensions, rt_config* cfg, char* out_name, int out_name_cap) {
    rt_default_config(cfg);
    if (out_name && out_name_cap > 0) std::snprintf(out_name, (size_t)out_name_cap, "output.ppm");
    auto is = [](const char* a, const char* s, const char* l) { return std::strcmp(a, s) == 0 || std::strcmp(a, l) == 0; };
    // std::stoi throws on a non-number and the reference then aborts; here the flag is reported and skipped.
    auto integer = [](const char* s, int32_t* out) {
        char* e = nullptr; long v = std::strtol(s, &e, 10);
        if (e == s) { std::fprintf(stderr, "Not a number: %s\n", s); return; }
        *out = (int32_t)v;
    };
    for (int i = 1; i < argc; ++i) {
        const char* a = argv[i];
        bool more = i + 1 < argc;
        if (is(a, "-ss", "--soft_shadows")) cfg->soft_shadows = 1;
        else if (is(a, "-sss", "--ss_samples") && more) integer(argv[++i], &cfg->ss_samples);
        else if (is(a, "-gr", "--glossy_reflect")) cfg->glossy_reflect = 1;
        else if (is(a, "-grs", "--gr_samples") && more) integer(argv[++i], &cfg->gr_samples);
        else if (is(a, "-aa", "--antialiasing")) cfg->anti_aliasing = 1;
        else if (is(a, "-aas", "--aa_samples") && more) integer(argv[++i], &cfg->aa_samples);
        else if (is(a, "-u", "--unaccelerated")) cfg->bvh = 0;
        else if (is(a, "-r", "--reflections")) cfg->reflect_refract_depth = 1;
        else if (is(a, "-rd", "--reflect_depth") && more) integer(argv[++i], &cfg->reflect_refract_depth);
        else if (is(a, "-t", "--texture_mapping")) cfg->textures = 1;
        else if (is(a, "-dof", "--depthoffield")) cfg->dof = 1;
        else if (is(a, "-dofs", "--dof_samples") && more) integer(argv[++i], &cfg->dof_samples);
        else if (is(a, "-m", "--motion-blur")) cfg->motion_blur = 1;
        else if (is(a, "-mbs", "--mb_samples") && more) integer(argv[++i], &cfg->mb_samples);
        else if (is(a, "-o", "--output") && more) { ++i; if (out_name && out_name_cap > 0) std::snprintf(out_name, (size_t)out_name_cap, "%s", argv[i]); }
        else if (extensions && std::strcmp(a, "--seed") == 0 && more) { int32_t v = 1; integer(argv[++i], &v); cfg->seed = (uint32_t)v; }
        else if (extensions && std::strcmp(a, "--fallback") == 0 && more) integer(argv[++i], &cfg->fallback);
        else if (extensions && (std::strcmp(a, "--scene") == 0 || std::strcmp(a, "--textures") == 0 || std::strcmp(a, "--outdir") == 0 ||
                                std::strcmp(a, "--gpu") == 0) && more) ++i;   // consumed by the caller
        else std::fprintf(stderr, "Unknown flag: %s\n", a);
    }
    return RT_OK;
}

rt_host_scene* rt_host_parse_scene(const char* text, const char* textures_dir, const rt_config* cfg, char* err, int err_cap) {
    try {
        if (!text || !cfg) throw std::runtime_error("null argument");
        return parse_scene_impl(sv(text), textures_dir ? textures_dir : "../Textures", *cfg);
    } catch (const std::exception& e) { copy_err(err, err_cap, e.what()); return nullptr; }
}

rt_host_scene* rt_host_load_scene(const char* json_path, const char* textures_dir, const rt_config* cfg, char* err, int err_cap) {
    try {
        if (!json_path || !cfg) throw std::runtime_error("null argument");
        std::ifstream f(json_path, std::ios::binary);
        if (!f) throw std::runtime_error(std::string("Error opening JSON file ") + json_path);
        std::stringstream ss; ss << f.rdbuf();
        std::string text = ss.str();
        return parse_scene_impl(sv(text), textures_dir ? textures_dir : "../Textures", *cfg);
    } catch (const std::exception& e) { copy_err(err, err_cap, e.what()); return nullptr; }
}

void rt_host_free_scene(rt_host_scene* s) {
    if (!s) return;
    delete static_cast<SceneStore*>(s->storage);
    delete s;
}

int rt_host_build_bvh(const rt_scene* sc, int32_t max_depth, rt_host_bvh* out) {
    if (!sc || !out || max_depth < 0 || max_depth > 30) return RT_ERR_ARG;
    try {
        std::vector<Box> pbox((size_t)sc->n_planes), cbox((size_t)sc->n_cubes), sbox((size_t)sc->n_spheres);
        rt_parallel_for(pbox.size(), 8192, [&](size_t lo, size_t hi, int) { for (size_t i = lo; i < hi; ++i) pbox[i] = plane_box(sc->planes[i], sc->plane_verts); });
        rt_parallel_for(cbox.size(), 2048, [&](size_t lo, size_t hi, int) { for (size_t i = lo; i < hi; ++i) cbox[i] = cube_box(sc->cubes[i]); });
        rt_parallel_for(sbox.size(), 8192, [&](size_t lo, size_t hi, int) { for (size_t i = lo; i < hi; ++i) sbox[i] = sphere_box(sc->spheres[i]); });
        struct Result { std::vector<rt_bvh_node> nodes; std::vector<int32_t> prim_ids; } b;
        const size_t total = pbox.size() + cbox.size() + sbox.size();
        if (total >= 65536 && max_depth >= 1 && max_depth <= 12 && rt_worker_count() > 1) {
            LevelBuilder lb{ pbox, cbox, sbox, {}, {} };
            lb.run(max_depth);
            b.nodes.swap(lb.nodes); b.prim_ids.swap(lb.prim_ids);
        } else {
            Builder rb{ pbox, cbox, sbox, {}, {} };
            std::vector<int32_t> pl((size_t)sc->n_planes), cu((size_t)sc->n_cubes), sp((size_t)sc->n_spheres);
            for (int i = 0; i < sc->n_planes; ++i) pl[i] = i;
            for (int i = 0; i < sc->n_cubes; ++i) cu[i] = i;
            for (int i = 0; i < sc->n_spheres; ++i) sp[i] = i;
            rb.build(pl, cu, sp, max_depth);
            b.nodes.swap(rb.nodes); b.prim_ids.swap(rb.prim_ids);
        }
        out->n_nodes = (int)b.nodes.size();
        out->nodes = (rt_bvh_node*)std::malloc(sizeof(rt_bvh_node) * std::max<size_t>(1, b.nodes.size()));
        std::memcpy(out->nodes, b.nodes.data(), sizeof(rt_bvh_node) * b.nodes.size());
        out->n_prim_ids = (int)b.prim_ids.size();
        out->prim_ids = (int32_t*)std::malloc(sizeof(int32_t) * std::max<size_t>(1, b.prim_ids.size()));
        std::memcpy(out->prim_ids, b.prim_ids.data(), sizeof(int32_t) * b.prim_ids.size());
        return RT_OK;
    } catch (const std::exception&) { return RT_ERR_ARG; }
}

void rt_host_free_bvh(rt_host_bvh* b) {
    if (!b) return;
    std::free(b->nodes); std::free(b->prim_ids);
    b->nodes = nullptr; b->prim_ids = nullptr; b->n_nodes = b->n_prim_ids = 0;
}

int rt_host_read_ppm(const char* path, int32_t* width, int32_t* height, int32_t* maxval, int32_t** rgb) {
    try {
        Ppm p = read_ppm_file(path);
        *width = p.w; *height = p.h; if (maxval) *maxval = p.maxv;
        *rgb = (int32_t*)std::malloc(sizeof(int32_t) * p.rgb.size());
        std::memcpy(*rgb, p.rgb.data(), sizeof(int32_t) * p.rgb.size());
        return RT_OK;
    } catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return RT_ERR_IO; }
}

int rt_host_write_ppm(const char* path, int32_t width, int32_t height, int32_t maxval, const uint8_t* rgb) {
    FILE* f = std::fopen(path, "wb");
    if (!f) { std::fprintf(stderr, "Failed to open the output file %s\n", path); return RT_ERR_IO; }
    // One formatted row at a time: a 4K frame is ~100 MB of text and per-value stream inserts dominate otherwise.
    static const char digits[] = "0123456789";
    std::fprintf(f, "P3\n%d %d\n%d\n", width, height, maxval);
    std::string row; row.reserve((size_t)width * 12 + 2);
    for (int y = 0; y < height; ++y) {
        row.clear();
        const uint8_t* p = rgb + (size_t)y * (size_t)width * 3;
        for (int k = 0; k < width * 3; ++k) {
            unsigned v = p[k];
            if (v >= 100) { row.push_back(digits[v / 100]); row.push_back(digits[(v / 10) % 10]); }
            else if (v >= 10) row.push_back(digits[v / 10]);
            row.push_back(digits[v % 10]);
            row.push_back(' ');
        }
        row.push_back('\n');
        std::fwrite(row.data(), 1, row.size(), f);
    }
    std::fclose(f);
    return RT_OK;
}

int rt_host_write_ppm_binary(const char* path, int32_t width, int32_t height, const uint8_t* rgb) {
    if (!path || !rgb || width <= 0 || height <= 0) return RT_ERR_ARG;
    FILE* f = std::fopen(path, "wb");
    if (!f) { std::fprintf(stderr, "Failed to open the output file %s\n", path); return RT_ERR_IO; }
    std::fprintf(f, "P6\n%d %d\n255\n", width, height);
    const size_t n = (size_t)width * (size_t)height * 3;
    const bool ok = std::fwrite(rgb, 1, n, f) == n;
    return (std::fclose(f) == 0 && ok) ? RT_OK : RT_ERR_IO;
}

void rt_host_free(void* p) { std::free(p); }

void rt_host_camera_ray(const rt_camera* c, float x, float y, float* out6);   // defined with the camera basis below

} // extern "C"

// ---------------------------------------------------------------------------------------------------
// Camera basis shared with the device upload (camera.cpp:109-178).  The per-frame constant part of
// Camera::convertPixelToRay is evaluated here once, with the same float/double steps the reference takes
// per ray, so the device only does the per-pixel part.
// ---------------------------------------------------------------------------------------------------
#include "camera_basis.h"

extern "C" void rt_host_camera_ray(const rt_camera* c, float x, float y, float* out6) {
    rt_camera_basis b = rt_make_camera_basis(*c);
    float u = (x + 0.5f) / c->res_x, v = (y + 0.5f) / c->res_y;
    float cam_x = -(u - 0.5f) * b.sw_over_fl, cam_y = (0.5f - v) * b.sh_over_fl, cam_z = -1.0f;
    float d[3];
    for (int i = 0; i < 3; ++i) d[i] = cam_x * b.right[i] + cam_y * b.up[i] + cam_z * b.fwd[i];
    float dlen = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (dlen < 1e-6f) { d[0] = 0.0f; d[1] = 0.0f; d[2] = 1.0f; }
    float len = (float)std::sqrt((double)d[0] * d[0] + (double)d[1] * d[1] + (double)d[2] * d[2]);
    for (int i = 0; i < 3; ++i) { out6[i] = c->location[i]; out6[3 + i] = -(d[i] / len); }
}
