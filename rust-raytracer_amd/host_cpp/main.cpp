// rtamd_render -- the reference's main.rs (main.rs:49-72) on top of the C ABI:
//   let world = select_scene(0); let result = world.cam.capture_image(integrator); result.save("output/test.png")
// and the same "Total / RT" timing print.  Also renders the reference's scene files.
//   rtamd_render [--scene cornell|FILE.json|FILE.yaml] [--cube data/mesh/cube.obj] [-w W] [-h H] [--spp N]
//                [--depth D] [--seed S] [--aspect A] [--integrator 0|1] [--sppm ITERATIONS PHOTONS_PER_ITER]
//                [--gpus N | --devices 0,1,...] [-o out.png] [--describe] [--vec3-selftest]
// --gpus N spreads the frame over N GPUs of this node inside ONE capture_image call (rt_render_multi: tiles dealt round-robin, RCCL
// gather; 0 = all visible); --devices names the HIP ordinal of every rank (an ordinal may repeat).
// `rtamd_render --cube data/mesh/cube.obj --sppm 50 500000` is the reference binary: SPPM pre-pass + 256 spp, output/test.png
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "rtamd.hpp"

using namespace rtamd_host;

// the reference's Vec3 unit tests (vec3.rs:425-564) against the mirrored Vec3
static int vec3_selftest() {
    int fails = 0;
#define EXPECT(c) if (!(c)) { std::printf("FAIL: %s\n", #c); fails++; }
    EXPECT(Vec3(1.0, 0.0, -1.0) + Vec3(2.0, 4.0, 6.0) == Vec3(3.0, 4.0, 5.0));
    EXPECT(Vec3(1.0, 0.0, -1.0) + 233.0 == Vec3(234.0, 233.0, 232.0));
    EXPECT(Vec3(1.0, 0.0, -1.0) - Vec3(2.0, 4.0, 6.0) == Vec3(-1.0, -4.0, -7.0));
    EXPECT(Vec3(1.0, 0.0, -1.0) - 1.0 == Vec3(0.0, -1.0, -2.0));
    EXPECT(Vec3(1.0, 0.0, -1.0) * Vec3::ones() == 0.0);
    EXPECT(Vec3(1.0, 0.0, -1.0) * 2.0 == Vec3(2.0, 0.0, -2.0));
    EXPECT(Vec3(1.0, -2.0, 0.0) / 2.0 == Vec3(0.5, -1.0, 0.0));
    EXPECT(Vec3::elemul(Vec3(1.0, 2.0, 3.0), Vec3(1.0, 2.0, 3.0)) == Vec3(1.0, 4.0, 9.0));
    EXPECT(Vec3::cross(Vec3(1.0, 2.0, 3.0), Vec3(2.0, 3.0, 4.0)) == Vec3(8.0 - 9.0, 6.0 - 4.0, 3.0 - 4.0));
    EXPECT(-Vec3(1.0, -2.0, 3.0) == Vec3(-1.0, 2.0, -3.0));
    EXPECT(Vec3(1.0, 2.0, 3.0).squared_length() == 14.0);
    EXPECT(Vec3(3.0, 4.0, 5.0).length() == std::sqrt(3.0 * 3.0 + 4.0 * 4.0 + 5.0 * 5.0));
    EXPECT(Vec3(233.0, 0.0, 0.0).unit() == Vec3(1.0, 0.0, 0.0));
    EXPECT(Vec3(-233.0, 0.0, 0.0).unit() == Vec3(-1.0, 0.0, 0.0));
    bool panicked = false;
    try { Vec3(0.0, 0.0, 0.0).unit(); } catch (const Error& e) { panicked = e.code == RT_ERR_UNIT_ZERO; }
    EXPECT(panicked);
    std::printf("vec3 selftest: %s\n", fails ? "FAILED" : "ok");
    return fails;
}

int main(int argc, char** argv) {
    std::string scene = "cornell", cube = "data/mesh/cube.obj", out = "output/test.png";
    Config cfg;
    double aspect = -1;
    bool describe = false;
    std::string checkpoint;  // --checkpoint FILE [--run-samples K]: trace the next K samples per pixel into the state in FILE; the image is written once all are in
    int run_samples = 64;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--scene") scene = next();
        else if (a == "--cube") cube = next();
        else if (a == "-w") cfg.width = std::atoi(next());
        else if (a == "-h") cfg.height = std::atoi(next());
        else if (a == "--spp") cfg.sample_per_pixel = std::atoi(next());
        else if (a == "--depth") cfg.max_depth = std::atoi(next());
        else if (a == "--seed") cfg.seed = std::strtoull(next(), nullptr, 10);
        else if (a == "--aspect") aspect = std::atof(next());
        else if (a == "--integrator") cfg.integrator = std::atoi(next());
        else if (a == "--sppm") { cfg.sppm_iterations = std::atoi(next()); cfg.sppm_photons_per_iter = std::atoi(next()); }
        else if (a == "--gpus") cfg.gpus = std::atoi(next());
        else if (a == "--devices") {
            for (const char* q = next(); *q;) {
                cfg.devices.push_back((int)std::strtol(q, const_cast<char**>(&q), 10));
                if (*q == ',') q++;
                else if (*q) { std::fprintf(stderr, "--devices wants a comma-separated list of ordinals\n"); return 2; }
            }
        }
        else if (a == "-o") out = next();
        else if (a == "--checkpoint") { checkpoint = next(); }
        else if (a == "--run-samples") run_samples = std::atoi(next());
        else if (a == "--describe") describe = true;
        else if (a == "--vec3-selftest") return vec3_selftest();
        else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    try {
        auto start_time = std::chrono::steady_clock::now();
        std::unique_ptr<World> world;
        if (scene == "cornell") {
            world = cornell_box_scene(cube, aspect > 0 ? aspect : (double)cfg.width / cfg.height);
        } else {
            world = std::make_unique<World>(scene);
            if (aspect > 0) world->cam.c.aspect = aspect;
        }
        rt_scene_info info;
        check(rt_scene_info_get(world->handle(), &info));
        std::printf("scene: %d nodes (%d boxes, %d spheres, %d rects, %d tris, %d xforms), %llu bytes flattened\n", info.n_nodes, info.n_boxes,
                    info.n_spheres, info.n_rects, info.n_tris, info.n_xforms, (unsigned long long)info.bytes);
        if (describe) return 0;
        if (!checkpoint.empty()) {
            RgbImage img;
            int done = 0;
            const bool complete = world->capture_image_resumable(cfg, checkpoint, run_samples, &img, &done);
            std::printf("checkpoint %s: %d of %d samples per pixel done%s\n", checkpoint.c_str(), done, cfg.sample_per_pixel, complete ? "" : " (run again to continue)");
            if (complete) img.save(out);
            return 0;
        }
        auto rt_start = std::chrono::steady_clock::now();
        rt_stats st{};
        std::vector<rt_stats> ranks;
        RgbImage result = world->capture_image(cfg, &st, nullptr, &ranks);
        result.save(out);
        auto end = std::chrono::steady_clock::now();
        double total = std::chrono::duration<double>(end - start_time).count(), rt = std::chrono::duration<double>(end - rt_start).count();
        double sppm = st.reserved[0] * 1e-6;
        std::printf("Total: %.3fs\n\tSPPM: %.3fs\n\tRT: %.3fs\n", total, sppm, rt - sppm);  // main.rs:57-71
        std::printf("%.2f Msamples/s (%llu samples, kernel %.1f ms in %d launches, scene %s)\n", st.samples / st.seconds / 1e6,
                    (unsigned long long)st.samples, st.kernel_ms, st.launches, st.scene_in_lds ? "in LDS" : "in L2/HBM");
        for (size_t i = 0; i < ranks.size(); i++)
            std::printf("\trank %zu: kernel %d, %.1f ms, %llu samples%s\n", i, ranks[i].kernel_used, ranks[i].kernel_ms, (unsigned long long)ranks[i].samples,
                        i == 0 ? (", exchange + stitch " + std::to_string(ranks[0].reserved[2] * 1e-3) + " ms, " + std::to_string(ranks[0].reserved[3]) + " rows through RCCL " +
                                  std::to_string(rt_rccl_version())).c_str() : "");
    } catch (const Error& e) {
        std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
