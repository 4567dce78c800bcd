// raytracer: command-line front end with the reference's behaviour (raytracer.cpp:30-199): same flags, same
// banner and progress lines, scene from ../ASCII/scene.json, textures from ../Textures, image to ../Output.
// The pixel loop between the two timer reads is one rt_render() call into the HIP pipeline.
//
// Extensions (not in the reference): --scene PATH, --textures DIR, --outdir DIR, --gpu N, --gpus N (the frame split over the
// first N devices of the node, tile blocks gathered with RCCL), --seed N, --fallback N, --p6 (binary PPM instead of the
// reference's text form).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rt_api.h"
#include "rt_host.h"

int main(int argc, char* argv[]) {
    rt_config cfg;
    char out_name[512];
    rt_host_parse_args(argc, argv, 1, &cfg, out_name, (int)sizeof out_name);
    std::string scene_path = "../ASCII/scene.json", tex_dir = "../Textures", out_dir = "../Output";
    int gpu = 0, gpus = 1;
    bool p6 = false;
    for (int i = 1; i < argc; ++i) if (!std::strcmp(argv[i], "--p6")) p6 = true;
    for (int i = 1; i + 1 < argc; ++i) {
        if (!std::strcmp(argv[i], "--scene")) scene_path = argv[i + 1];
        else if (!std::strcmp(argv[i], "--textures")) tex_dir = argv[i + 1];
        else if (!std::strcmp(argv[i], "--outdir")) out_dir = argv[i + 1];
        else if (!std::strcmp(argv[i], "--gpu")) gpu = std::atoi(argv[i + 1]);
        else if (!std::strcmp(argv[i], "--gpus")) gpus = std::atoi(argv[i + 1]);
    }

    // raytracer.cpp:34-49
    std::printf("---Config---\nBVH: %d\nAnti-aliasing: %d\n  AA Samples: %d\nReflections: %d\nTexture Mapping: %d\n"
                "-Distributed effects-\n  Glossy reflections: %d\n  GR samples: %d\n  Soft shadows: %d\n  SS samples: %d\n"
                "-Lens effects-\n  Motion blur: %d\n  MB samples: %d\n  Depth of Field: %d\n  DoF samples: %d\n\n",
                cfg.bvh, cfg.anti_aliasing, cfg.aa_samples, cfg.reflect_refract_depth, cfg.textures, cfg.glossy_reflect,
                cfg.gr_samples, cfg.soft_shadows, cfg.ss_samples, cfg.motion_blur, cfg.mb_samples, cfg.dof, cfg.dof_samples);

    char err[512] = "";
    rt_host_scene* hs = rt_host_load_scene(scene_path.c_str(), tex_dir.c_str(), &cfg, err, (int)sizeof err);
    if (!hs) { std::fprintf(stderr, "%s\n", err); return 1; }
    const rt_scene& sc = hs->scene;
    const int W = (int)sc.camera.res_x, H = (int)sc.camera.res_y;
    std::printf("\n---Raytracing for %d object(s)---\n", sc.n_spheres + sc.n_cubes + sc.n_planes);

    if (gpus > 1) {
        // one process, several devices: rt_multi_* (tiles interleaved over the devices, blocks gathered to the first one)
        std::vector<int32_t> ids; for (int i = 0; i < gpus; ++i) ids.push_back(gpu + i);
        rt_multi* mm = rt_multi_create(gpus, ids.data(), 0);
        if (!mm) { std::fprintf(stderr, "%s\n", rt_multi_last_error(nullptr)); rt_host_free_scene(hs); return 1; }
        rt_host_bvh mb{};
        int mrc = cfg.bvh ? rt_host_build_bvh(&sc, 8, &mb) : RT_OK;
        std::printf(cfg.bvh ? "Running BVH accelerated RT...\n" : "Running unaccelerated RT...\n");
        if (mrc == RT_OK) mrc = rt_multi_upload(mm, &sc, cfg.bvh ? mb.nodes : nullptr, mb.n_nodes, mb.prim_ids, mb.n_prim_ids);
        std::vector<uint8_t> frame((size_t)W * H * 3);
        auto m0 = std::chrono::system_clock::now();
        if (mrc == RT_OK) mrc = rt_multi_render(mm, &cfg, frame.data());
        auto m1 = std::chrono::system_clock::now();
        if (mrc != RT_OK) { std::fprintf(stderr, "%s\n", rt_multi_last_error(mm)); rt_multi_destroy(mm); rt_host_free_scene(hs); return 1; }
        for (int p = 10; p <= 90; p += 10) std::printf("%d%% done\n", p);
        std::printf("Finished! Time taken: %gs\n", (double)((float)std::chrono::duration_cast<std::chrono::milliseconds>(m1 - m0).count() / 1000.0f));
        std::string mpath = out_dir + "/" + out_name;
        mrc = p6 ? rt_host_write_ppm_binary(mpath.c_str(), W, H, frame.data()) : rt_host_write_ppm(mpath.c_str(), W, H, 255, frame.data());
        rt_host_free_bvh(&mb); rt_multi_destroy(mm); rt_host_free_scene(hs);
        return mrc == RT_OK ? 0 : 1;
    }

    rt_ctx* ctx = rt_create(gpu, 0);
    if (!ctx) { std::fprintf(stderr, "%s\n", rt_last_error(nullptr)); rt_host_free_scene(hs); return 1; }
    int rc = rt_upload_scene(ctx, &sc);
    rt_host_bvh bvh{};
    if (rc == RT_OK && cfg.bvh) {                                   // raytracer.cpp:72-79
        rc = rt_host_build_bvh(&sc, 8, &bvh);
        if (rc == RT_OK) rc = rt_upload_bvh(ctx, bvh.nodes, bvh.n_nodes, bvh.prim_ids, bvh.n_prim_ids);
        std::printf("Running BVH accelerated RT...\n");
    } else if (rc == RT_OK) {
        std::printf("Running unaccelerated RT...\n");
    }
    if (rc != RT_OK) { std::fprintf(stderr, "%s\n", rt_last_error(ctx)); rt_destroy(ctx); rt_host_free_scene(hs); return 1; }

    std::vector<uint8_t> rgb((size_t)W * H * 3);
    rt_stats st{};
    auto t0 = std::chrono::system_clock::now();
    rc = rt_render(ctx, &cfg, rgb.data(), &st);                     // raytracer.cpp:109-193
    auto t1 = std::chrono::system_clock::now();
    if (rc != RT_OK) { std::fprintf(stderr, "%s\n", rt_last_error(ctx)); rt_destroy(ctx); rt_host_free_scene(hs); return 1; }
    for (int p = 10; p <= 90; p += 10) std::printf("%d%% done\n", p);   // the frame finishes in one call
    long long ms = std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count();
    std::printf("Finished! Time taken: %gs\n", (double)((float)ms / 1000.0f));
    unsigned long long rays = st.primary_rays + st.shadow_rays + st.reflection_rays + st.refraction_rays;
    std::printf("rays traced: %llu (primary %llu, shadow %llu, reflection %llu, refraction %llu), %.3f ms on device\n", rays,
                (unsigned long long)st.primary_rays, (unsigned long long)st.shadow_rays, (unsigned long long)st.reflection_rays,
                (unsigned long long)st.refraction_rays, st.ms_total);

    std::string out_path = out_dir + "/" + out_name;
    rc = p6 ? rt_host_write_ppm_binary(out_path.c_str(), W, H, rgb.data())
            : rt_host_write_ppm(out_path.c_str(), W, H, 255, rgb.data());   // raytracer.cpp:195; template maxval is 255
    rt_host_free_bvh(&bvh);
    rt_destroy(ctx);
    rt_host_free_scene(hs);
    return rc == RT_OK ? 0 : 1;
}
