// TEST-ONLY driver for the ThreadSanitizer build of the host half of librtamd (run_host_tsan.sh): rt_render_multi's fan-out -- one host
// thread per rank -- on the device stub's fake devices, from two caller threads at once (two scenes), several device lists, and a tuning
// change between frames.  Exit code 0 and no TSan report = no data race between the ranks' threads, the tuning snapshot, the error slots,
// the stub's bookkeeping and the stitch.  Nothing here traces a ray (device_stub.cpp writes a pattern).
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "rtamd.h"

static int frames(const char* scene_file, int seed, std::vector<double>* first) {
    rt_scene* s = nullptr;
    rt_camera cam;
    if (rt_scene_load_file(scene_file, &s, &cam) < 0) { std::fprintf(stderr, "load: %s\n", rt_last_error()); return 1; }
    rt_params p;
    rt_default_params(&p);
    p.width = 100; p.height = 52; p.spp = 3; p.seed = (uint64_t)seed;
    const int lists[][5] = {{0, 1, 2, 3, -1}, {2, 0, 2, 1, 0}, {3, -1, -1, -1, -1}, {1, 1, 1, -1, -1}};
    int rc = 0;
    for (int rep = 0; rep < 3 && !rc; rep++)
        for (auto& l : lists) {
            int n = 0;
            while (n < 5 && l[n] >= 0) n++;
            std::vector<double> out((size_t)p.width * p.height * 3);
            std::vector<rt_stats> st((size_t)n);
            if (rt_render_multi(s, &cam, &p, n, l, out.data(), st.data()) < 0) { std::fprintf(stderr, "render: %s\n", rt_last_error()); rc = 1; break; }
            if (first->empty()) *first = out;
            else if (std::memcmp(first->data(), out.data(), out.size() * sizeof(double)) != 0) { std::fprintf(stderr, "frames differ between device lists\n"); rc = 1; break; }
            rt_tuning t;
            rt_tuning_default(&t);
            t.multi_force_rccl = (rep + n) & 1;  // every rank's rows through the (stub) exchange, or only the remote ones
            rt_tuning_set(&t);
        }
    rt_scene_destroy(s);
    return rc;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    int rc[2] = {0, 0};
    std::vector<double> f0, f1;
    std::thread a([&] { rc[0] = frames(argv[1], 1, &f0); }), b([&] { rc[1] = frames(argv[1], 2, &f1); });
    a.join();
    b.join();
    rt_release_workspaces();
    std::printf("tsan fan-out driver: %s\n", (rc[0] | rc[1]) ? "FAILED" : "ok");
    return rc[0] | rc[1];
}
