// TEST-ONLY stub of csrc/device/device.h so that the HOST half of librtamd (builders, loaders, flattener, accel builder,
// PNG writer, and the fan-out / partition / gather / stitch logic of rt_render_multi) can be linked with
// -fsanitize=address,undefined and exercised on a GPU-less box (GPU ASan is not available on this pool).  Nothing here is part
// of the product and nothing here traces a ray.
//
// By default every device entry point reports RT_ERR_NO_DEVICE.  With RTAMD_STUB_DEVICES=N in the environment (read HERE, by the
// test stub -- the product reads no environment variable) the stub models N "devices" whose memory is host memory:
//   render_tiles   fills the rank's tile-major rows with a PATTERN that depends only on (global tile index, pixel in tile, channel,
//                  seed) -- value(t, pix, c) below -- so a stitched frame is right iff every tile went through the right slot;
//   exchange_post  copies one row (what a grouped ncclSend / ncclRecv pair does), counts it and refuses overlapping calls;
//                  RTAMD_STUB_STAGGER_MS=k makes rank r's render take k * r milliseconds, so the ranks finish one after the other;
//   assemble_frame is the stitch of camera.rs:115-123 restated on the host (tile t of rank r sits at row r, slot t / world).
// Device memory is malloc'ed per allocation, so a slot computed wrongly is a heap overflow that ASan reports.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>

#include "device/device.h"
namespace rtamd {
static int stub_devices() {
    const char* e = std::getenv("RTAMD_STUB_DEVICES");
    return e ? std::atoi(e) : 0;
}
static void none() { throw RtError(RT_ERR_NO_DEVICE, "sanitizer build: host code only"); }
static thread_local int t_device = 0;
static std::mutex g_stub_mu;
struct Alloc {
    int device;
    size_t bytes;
};
static std::map<void*, Alloc> g_owner;  // allocation -> device it was made on, size
static long g_rows_moved = 0;

double stub_value(int64_t t, int pix, int c, uint64_t seed) { return (double)(t * 1000 + pix * 3 + c) + (double)(seed % 7) * 0.125; }

void render_tiles(const rt_scene&, const CameraDev&, const RenderPlan& pl, double* d_tiles_in, void*, rt_stats* st) {
    if (stub_devices() < 1) none();
    double* d_tiles = pl.ext_accum ? pl.ext_accum : d_tiles_in;  // (what the call writes: it must be memory of the current device)
    {
        std::lock_guard<std::mutex> g(g_stub_mu);
        // the rows must live on the device this thread has made current (rt_render_multi's per-rank threads)
        bool found = false;
        for (auto& kv : g_owner) {
            const char* b = (const char*)kv.first;
            if ((const char*)d_tiles >= b && (const char*)d_tiles < b + kv.second.bytes) {
                found = true;
                if (kv.second.device != t_device) throw RtError(RT_ERR_INTERNAL, "stub: rows are not on the current device");
            }
        }
        if (!found) throw RtError(RT_ERR_INTERNAL, "stub: rows are not device memory");
    }
    if (pl.ext_accum) {  // resumable rendering: every sample index of the range adds the pattern once (s_first == 0 initialises)
        const int n = (pl.s_last < 0 ? pl.spp : pl.s_last) - pl.s_first;
        for (int64_t lt = 0; lt < pl.tiles_owned; lt++) {
            const int64_t t = lt * pl.world + pl.rank;
            for (int pix = 0; pix < TILE_PIX; pix++)
                for (int c = 0; c < 3; c++) {
                    double& a = pl.ext_accum[((size_t)lt * TILE_PIX + pix) * 3 + c];
                    a = (pl.s_first == 0 ? 0. : a) + n * stub_value(t, pix, c, pl.seed);
                }
        }
    } else
    for (int64_t lt = 0; lt < pl.tiles_owned; lt++) {
        const int64_t t = lt * pl.world + pl.rank;
        for (int pix = 0; pix < TILE_PIX; pix++)
            for (int c = 0; c < 3; c++) d_tiles[((size_t)lt * TILE_PIX + pix) * 3 + c] = stub_value(t, pix, c, pl.seed);
    }
    if (const char* e = std::getenv("RTAMD_STUB_STAGGER_MS")) std::this_thread::sleep_for(std::chrono::milliseconds(std::atoi(e) * pl.rank));
    if (st) {
        st->kernel_ms = 1.0;
        st->launches = 1;
        st->kernel_used = 2;
    }
}
void render_sppm(const rt_scene& s, const CameraDev& cam, RenderPlan pl, const rt_sppm_config&, double* d_tiles, double*, void* stream, rt_stats* st, uint64_t*) {
    render_tiles(s, cam, pl, d_tiles, stream, st);
}
void assemble_frame(const RenderPlan& pl, const double* gathered, int64_t stride, double* frame, void*) {
    if (stub_devices() < 1) none();
    for (int y = 0; y < pl.height; y++)
        for (int x = 0; x < pl.width; x++) {
            const int64_t t = (int64_t)(y / TILE_H) * pl.tiles_x + x / TILE_W;
            const int64_t r = t % pl.world, lt = t / pl.world;
            const int pix = (y % TILE_H) * TILE_W + x % TILE_W;
            for (int c = 0; c < 3; c++) frame[((size_t)y * pl.width + x) * 3 + c] = gathered[((size_t)(r * stride + lt) * TILE_PIX + pix) * 3 + c];
        }
}
void finalize_tiles(const RenderPlan& pl, const double* acc, double* tiles, void*) {
    if (stub_devices() < 1) none();
    for (int64_t i = 0; i < pl.tiles_owned * TILE_PIX * 3; i++) tiles[i] = acc[i] / pl.spp;
}
void debug_rng_device(uint64_t, uint64_t, uint64_t, int, uint64_t*) { none(); }
void debug_rng_floats_device(uint64_t, uint64_t, uint64_t, int, double, double, double*, double*) { none(); }
void debug_math_device(int, size_t, const double*, const double*, double*) { none(); }
void debug_hit_device(const rt_scene&, int, size_t, const double*, double, double, double*) { none(); }
int device_count() { return stub_devices(); }
void* dev_alloc(size_t n) {
    if (stub_devices() < 1) none();
    void* p = std::malloc(n ? n : 16);
    std::lock_guard<std::mutex> g(g_stub_mu);
    g_owner[p] = Alloc{t_device, n ? n : 16};
    return p;
}
void dev_free(void* p) {
    {
        std::lock_guard<std::mutex> g(g_stub_mu);
        g_owner.erase(p);
    }
    std::free(p);
}
void dev_copy_to_host(void* dst, const void* src, size_t n) {
    if (stub_devices() < 1) none();
    std::memcpy(dst, src, n);
}
void dev_copy_to_device(void* dst, const void* src, size_t n) {
    if (stub_devices() < 1) none();
    std::memcpy(dst, src, n);
}
void dev_set_device(int d) {
    if (d < 0 || d >= stub_devices()) none();
    t_device = d;
}
int dev_get_device() {
    if (stub_devices() < 1) none();
    return t_device;
}
void dev_synchronize() {}
void free_device_copies(rt_scene&) {}
size_t release_workspaces() { return 0; }

struct Exchange {
    std::vector<int> devices;
    bool posting = false;  // exchange_post calls on ONE communicator set must not overlap (several rank threads drive it)
};
static int g_ex_open = 0, g_ex_created = 0, g_ex_destroyed = 0;
static std::vector<std::vector<int>> g_ex_cache;  // device lists whose communicators "exist" (the real cache keeps them between calls)
Exchange* exchange_open(const std::vector<int>& devices, bool* created) {
    for (size_t i = 0; i < devices.size(); i++)
        for (size_t j = 0; j < i; j++)
            if (devices[i] == devices[j]) throw RtError(RT_ERR_ARG, "stub: a communicator holds a device once (ncclCommInitAll would refuse)");
    std::lock_guard<std::mutex> g(g_stub_mu);
    bool found = false;
    for (auto& d : g_ex_cache) found = found || d == devices;
    if (!found) {
        g_ex_cache.push_back(devices);
        g_ex_created++;
    }
    if (created) *created = !found;
    g_ex_open++;
    Exchange* e = new Exchange();
    e->devices = devices;
    return e;
}
void exchange_close(Exchange* e, bool failed) {
    if (!e) return;
    {
        std::lock_guard<std::mutex> g(g_stub_mu);
        g_ex_open--;
        if (failed) {  // the product destroys the communicators of a failed exchange instead of caching them
            for (auto it = g_ex_cache.begin(); it != g_ex_cache.end(); ++it)
                if (*it == e->devices) {
                    g_ex_cache.erase(it);
                    break;
                }
            g_ex_destroyed++;
        }
    }
    delete e;
}
void exchange_post(Exchange* e, const RowMove& m) {
    {
        std::lock_guard<std::mutex> g(g_stub_mu);
        if (e->posting) throw RtError(RT_ERR_INTERNAL, "stub: two exchange_post calls on one communicator set at once");
        e->posting = true;
    }
    struct Done {
        Exchange* e;
        ~Done() {
            std::lock_guard<std::mutex> g(g_stub_mu);
            e->posting = false;
        }
    } done{e};
    std::this_thread::sleep_for(std::chrono::microseconds(200));  // (a window in which an unserialised second post would be caught)
    if (const char* f = std::getenv("RTAMD_STUB_FAIL_POST"))  // "fail the post of rows from this comm rank": the error path of a failed exchange
        if (std::atoi(f) == m.src_rank) throw RtError(RT_ERR_HIP, "stub: ncclSend failed (asked for by RTAMD_STUB_FAIL_POST)");
    if (m.src_rank < 0 || m.src_rank >= (int)e->devices.size() || m.dst_rank < 0 || m.dst_rank >= (int)e->devices.size())
        throw RtError(RT_ERR_ARG, "stub: bad comm rank");
    {   // a row leaves the device of its source rank and lands on the device of its destination rank
        std::lock_guard<std::mutex> g(g_stub_mu);
        auto owner = [&](const void* q) {
            for (auto& kv : g_owner)
                if ((const char*)q >= (const char*)kv.first && (const char*)q < (const char*)kv.first + kv.second.bytes) return kv.second.device;
            return -1;
        };
        if (owner(m.src) != e->devices[(size_t)m.src_rank] || owner(m.dst) != e->devices[(size_t)m.dst_rank])
            throw RtError(RT_ERR_INTERNAL, "stub: a row's buffer is not on its communicator rank's device");
    }
    std::memcpy(m.dst, m.src, m.count * sizeof(double));
    std::lock_guard<std::mutex> g(g_stub_mu);
    g_rows_moved++;
}
void exchange_wait(Exchange*) {}
size_t exchange_release_idle() { return 0; }
int exchange_library_version() { return 0; }
}  // namespace rtamd
