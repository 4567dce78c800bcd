// The binding a maintainer of the reference would add, compiled for real: the reference's OWN front end - Config::parseArgs,
// the five parse...FromJson calls, Image::readImage of the template, BVHNode::buildBVH - feeds the C ABI of
// include/rt_api.h in place of the pixel loop (raytracer.cpp:105-193), and the result goes back out through the
// reference's Image::modifyPixel / Image::writeImage.  This file is this repository's code; it is linked against the
// reference's unmodified objects (raytracer.cpp compiled with -Dmain=ref_main) by `make -C oracle bind`.
// TEST INFRASTRUCTURE: it proves the boundary against the reference's classes (bvh.h:13-37, raytracer.cpp:55-79).
#include "bvh.h"
#include "camera.h"
#include "config.h"
#include "cube.h"
#include "image.h"
#include "plane.h"
#include "pointlight.h"
#include "sphere.h"

#include "rt_api.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

using namespace std::chrono;

namespace {

rt_material to_mat(const Shape& s, std::vector<std::string>& tex_names) {
    rt_material m{};
    for (int i = 0; i < 3; ++i) { m.diffuse[i] = s.diffuse[i]; m.specular[i] = s.specular[i]; }
    m.shininess = s.shininess; m.transparency = s.transparency; m.ior = s.ior;
    m.tex_id = -1;
    if (s.hasTex) {                                          // shape.h:23-24: set only with -t and a texture name
        auto it = std::find(tex_names.begin(), tex_names.end(), s.texture);
        m.tex_id = (int)(it - tex_names.begin());
        if (it == tex_names.end()) tex_names.push_back(s.texture);
    }
    return m;
}

bool same_shape(const Shape& a, const Shape& b) {
    return a.startLocation == b.startLocation && a.endLocation == b.endLocation && a.diffuse == b.diffuse && a.specular == b.specular &&
           a.shininess == b.shininess && a.transparency == b.transparency && a.ior == b.ior && a.texture == b.texture && a.hasTex == b.hasTex;
}
bool same(const Plane& a, const Plane& b) { return same_shape(a, b) && a.vertices == b.vertices && a.normal == b.normal; }
bool same(const Cube& a, const Cube& b) { return same_shape(a, b) && a.rotation == b.rotation && a.scale == b.scale; }
bool same(const Sphere& a, const Sphere& b) { return same_shape(a, b) && a.radius == b.radius; }

// A leaf of the reference's tree holds COPIES of the shapes (bvh.cpp:113-115), in the order of the scene's vectors
// (every split keeps the order, bvh.cpp:139-160).  Their indices in the scene's vectors: the next equal object after
// the previous match; identical objects are interchangeable.
template <class T>
void leaf_ids(const std::vector<T>& leaf, const std::vector<T>& all, std::vector<int32_t>& out) {
    size_t from = 0;
    for (const T& s : leaf) {
        size_t k = from;
        while (k < all.size() && !same(s, all[k])) ++k;
        if (k == all.size()) {                               // equal objects met out of order: search from the start
            k = 0;
            while (k < all.size() && !same(s, all[k])) ++k;
            if (k == all.size()) throw std::runtime_error("a leaf holds a shape the scene does not");
        }
        out.push_back((int32_t)k);
        from = k + 1;
    }
}

// BVHNode pointer tree (bvh.h:13-37) -> rt_bvh_node array, root at index 0, children by index
int flatten(const BVHNode* n, const std::vector<Plane>& planes, const std::vector<Cube>& cubes, const std::vector<Sphere>& spheres,
            std::vector<rt_bvh_node>& nodes, std::vector<int32_t>& ids) {
    const int me = (int)nodes.size();
    nodes.emplace_back();
    rt_bvh_node o{};
    for (int i = 0; i < 3; ++i) { o.box_min[i] = n->aabb.min[i]; o.box_max[i] = n->aabb.max[i]; }
    o.left = o.right = -1; o.first_prim = 0;
    if (!n->left && !n->right) {
        o.first_prim = (int32_t)ids.size();
        o.n_planes = (int32_t)n->planes.size(); o.n_cubes = (int32_t)n->cubes.size(); o.n_spheres = (int32_t)n->spheres.size();
        leaf_ids(n->planes, planes, ids); leaf_ids(n->cubes, cubes, ids); leaf_ids(n->spheres, spheres, ids);
    }
    if (n->left) o.left = flatten(n->left, planes, cubes, spheres, nodes, ids);
    if (n->right) o.right = flatten(n->right, planes, cubes, spheres, nodes, ids);
    nodes[(size_t)me] = o;
    return me;
}

} // namespace

int main(int argc, char* argv[]) {
    try {
        Config config = Config::parseArgs(argc, argv);                                   // config.cpp:7-91
        std::vector<Camera> cams = Camera::parseCameraDataFromJson(config);             // raytracer.cpp:55-59
        std::vector<PointLight> lights = PointLight::parseLightDataFromJson();
        std::vector<Sphere> spheres = Sphere::parseSphereDataFromJson(config);
        std::vector<Cube> cubes = Cube::parseCubeDataFromJson(config);
        std::vector<Plane> planes = Plane::parsePlaneDataFromJson(config);
        Image img("blank_1920x1080.ppm");                                                // raytracer.cpp:63-64
        img.readImage();
        BVHNode* bvh = config.bvh ? BVHNode::buildBVH(planes, cubes, spheres, config) : nullptr;   // raytracer.cpp:72-79

        // ---- the reference's containers -> the PODs of rt_api.h --------------------------------------------------
        std::vector<std::string> tex_names;
        std::vector<rt_sphere> S; std::vector<rt_cube> C; std::vector<rt_plane> P; std::vector<float> PV; std::vector<rt_light> L;
        // planes first: material (and texture) ids follow the order planes, cubes, spheres only through tex_names, which
        // is an arbitrary but consistent numbering
        for (auto& p : planes) {
            rt_plane o{}; o.n_verts = (int)p.vertices.size(); o.first_vert = (int)PV.size() / 3;
            for (auto& v : p.vertices) PV.insert(PV.end(), v.begin(), v.end());          // file order: no intersect() has run yet
            std::copy_n(p.normal.begin(), 3, o.normal); o.mat = to_mat(p, tex_names); P.push_back(o);
        }
        for (auto& c : cubes) {
            rt_cube o{}; std::copy_n(c.startLocation.begin(), 3, o.start); std::copy_n(c.endLocation.begin(), 3, o.end);
            std::copy_n(c.rotation.begin(), 3, o.rotation); o.scale = c.scale; o.mat = to_mat(c, tex_names); C.push_back(o);
        }
        for (auto& s : spheres) {
            rt_sphere o{}; std::copy_n(s.startLocation.begin(), 3, o.start); std::copy_n(s.endLocation.begin(), 3, o.end);
            o.radius = s.radius; o.mat = to_mat(s, tex_names); S.push_back(o);
        }
        for (auto& l : lights) { rt_light o{}; std::copy_n(l.location.begin(), 3, o.location); o.rad_intensity = l.rad_intensity; L.push_back(o); }

        // what blinnPhong's texture cache would load lazily (raytracer.cpp:271-285), read with the reference's reader
        std::vector<std::vector<int32_t>> tex_flat; std::vector<rt_texture> T;
        for (auto& name : tex_names) {
            Image im(name); im.readImage();
            tex_flat.emplace_back();
            for (auto& row : im.pixelValues) for (auto& px : row) tex_flat.back().insert(tex_flat.back().end(), px.begin(), px.end());
            T.push_back(rt_texture{ im.width, im.height, nullptr });
        }
        for (size_t i = 0; i < T.size(); ++i) T[i].rgb = tex_flat[i].data();

        rt_scene sc{};
        const Camera& cam = cams.at(0);
        std::copy_n(cam.location.begin(), 3, sc.camera.location); std::copy_n(cam.gaze_vector.begin(), 3, sc.camera.gaze);
        sc.camera.aperture = cam.aperture; sc.camera.focal_length = cam.focal_length; sc.camera.focal_distance = cam.focal_distance;
        sc.camera.sensor_width = cam.sensor_width; sc.camera.sensor_height = cam.sensor_height; sc.camera.res_x = cam.res_x; sc.camera.res_y = cam.res_y;
        sc.n_lights = (int)L.size(); sc.lights = L.data(); sc.n_spheres = (int)S.size(); sc.spheres = S.data();
        sc.n_cubes = (int)C.size(); sc.cubes = C.data(); sc.n_planes = (int)P.size(); sc.planes = P.data();
        sc.n_plane_verts = (int)PV.size() / 3; sc.plane_verts = PV.data(); sc.n_textures = (int)T.size(); sc.textures = T.data();

        rt_config rc; rt_default_config(&rc);
        rc.soft_shadows = config.softShadows; rc.ss_samples = config.SSsamples; rc.glossy_reflect = config.glossyReflect;
        rc.gr_samples = config.GRsamples; rc.light_radius = config.lightRadius; rc.anti_aliasing = config.antiAliasing;
        rc.aa_samples = config.AAsamples; rc.textures = config.textures; rc.bvh = config.bvh;
        rc.reflect_refract_depth = config.reflectRefractDepth; rc.dof = config.dof; rc.dof_samples = config.dofSamples;
        rc.motion_blur = config.motionBlur; rc.mb_samples = config.mbSamples;

        rt_ctx* ctx = rt_create(0, 0);
        if (!ctx) throw std::runtime_error(rt_last_error(nullptr));
        if (rt_upload_scene(ctx, &sc) != RT_OK) throw std::runtime_error(rt_last_error(ctx));
        if (bvh) {
            std::vector<rt_bvh_node> nodes; std::vector<int32_t> ids;
            flatten(bvh, planes, cubes, spheres, nodes, ids);
            if (rt_upload_bvh(ctx, nodes.data(), (int32_t)nodes.size(), ids.data(), (int32_t)ids.size()) != RT_OK)
                throw std::runtime_error(rt_last_error(ctx));
            std::cout << "bound the reference's BVH: " << nodes.size() << " nodes, " << ids.size() << " leaf entries\n";
        }

        // ---- in place of the pixel loop, raytracer.cpp:109-193 ---------------------------------------------------
        uint64_t start = duration_cast<milliseconds>(system_clock::now().time_since_epoch()).count();
        const int W = (int)cam.res_x, H = (int)cam.res_y;
        std::vector<uint8_t> rgb((size_t)W * (size_t)H * 3);
        rt_stats st{};
        if (rt_render(ctx, &rc, rgb.data(), &st) != RT_OK) throw std::runtime_error(rt_last_error(ctx));
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const uint8_t* p = &rgb[((size_t)y * (size_t)W + (size_t)x) * 3];
                img.modifyPixel(x, y, p[0], p[1], p[2]);                                 // raytracer.cpp:178
            }
        uint64_t end = duration_cast<milliseconds>(system_clock::now().time_since_epoch()).count();
        std::cout << "Finished! Time taken: " << (end - start) / 1000.0f << "s\n";
        img.writeImage(config.outputFile);                                               // raytracer.cpp:195
        rt_destroy(ctx);
        delete bvh;
        return 0;
    } catch (const std::exception& e) {
        std::cerr << "ref_bind: " << e.what() << "\n";
        return 1;
    }
}
