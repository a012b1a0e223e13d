// Host-callable launch interface of the HIP kernels (device/kernels.hip).
#pragma once
#include <cstdint>
#include <vector>

#include "../common/flat.h"
#include "../host/scene.h"

namespace rtamd {

static const int TILE_W = 8, TILE_H = 8, TILE_PIX = 64;  // one 8x8 pixel tile == one wave64 of primary rays

struct RenderPlan {
    int width, height, spp, max_depth;
    double t_min;
    uint64_t seed;
    int rank, world;
    int tiles_x, tiles_y;
    int64_t tiles_total, tiles_owned;
    int spp_chunk;  // samples per pixel per launch (default: all of them)
    int sub_spp;    // samples per pixel per work unit (a wave's pool = 64 * sub_spp paths; at most 8)
    int kernel;
    int integrator;  // 0 BSDF sampling, 1 light/cosine mixture pdf, 2 SPPM final gather (needs sppm_est)
    double time0 = 0., time1 = 0.;  // D9: the camera's shutter (time1 > time0: every sample draws a time)
    const double* sppm_est = nullptr;  // device: per pixel {caustic estimate[3], global estimate[3]}
    // resumable rendering (rt_render_accumulate_device): only the sample indices [s_first, s_last) are traced (s_last < 0: spp) and folded
    // into the CALLER's accumulator (device, [tiles_owned][64][3] f64 sums; s_first == 0 initialises it); no division by spp, d_tiles unused
    int s_first = 0, s_last = -1;
    double* ext_accum = nullptr;
};

// Renders plan.tiles_owned tiles into d_tiles (device, tile-major f64 RGB) on `stream`; blocks until done.
// Throws RtError.
void render_tiles(const rt_scene& s, const CameraDev& cam, const RenderPlan& plan, double* d_tiles, void* stream, rt_stats* st);
// SPPMIntegrator::new + capture_image: pre-pass statistics (optional host copy, 10 f64 per pixel) and the final render
void render_sppm(const rt_scene& s, const CameraDev& cam, RenderPlan plan, const rt_sppm_config& cfg, double* d_tiles, double* stats_host,
                 void* stream, rt_stats* st, uint64_t* totals2);
void assemble_frame(const RenderPlan& plan, const double* d_gathered, int64_t tiles_per_rank_stride, double* d_frame, void* stream);
// pixel_color /= spp (camera.rs:102) of an accumulator filled by render_tiles(plan.ext_accum): d_tiles = d_accum / plan.spp inside the image, 0 outside
void finalize_tiles(const RenderPlan& plan, const double* d_accum, double* d_tiles, void* stream);
void debug_rng_device(uint64_t seed, uint64_t pixel, uint64_t sample, int n, uint64_t* out_host);
void debug_rng_floats_device(uint64_t seed, uint64_t pixel, uint64_t sample, int n, double lo, double hi, double* out_gen, double* out_range);
void debug_math_device(int op, size_t n, const double* a, const double* b, double* out);
void debug_hit_device(const rt_scene& s, int kernel, size_t n, const double* rays, double t_min, double t_max, double* out);
int device_count();
// thin HIP wrappers so abi.cpp stays free of HIP headers
void* dev_alloc(size_t n);
void dev_free(void* p);
void dev_copy_to_host(void* dst, const void* src, size_t n);
void dev_copy_to_device(void* dst, const void* src, size_t n);
void dev_set_device(int d);
void free_device_copies(rt_scene& s);
size_t release_workspaces();
int dev_get_device();
void dev_synchronize();  // the current device

// ---- the exchange of the multi-device frame (device/exchange.hip: RCCL; rt_render_multi in abi.cpp drives it) ----
struct Exchange;  // communicators + one stream per device of a device list (cached per process, leased)
struct RowMove {  // `count` f64 from `src` on comm rank src_rank's device to `dst` on comm rank dst_rank's device
    int src_rank;
    const double* src;
    int dst_rank;
    double* dst;
    size_t count;
};
Exchange* exchange_open(const std::vector<int>& devices, bool* created = nullptr);  // comm rank r = devices[r] (distinct ordinals); ncclCommInitAll on first use (*created)
void exchange_close(Exchange* e, bool failed = false);     // returns the lease; the communicators stay cached -- unless an exchange on them failed: then they are destroyed
// One row: ncclSend on the source rank's communicator + the matching ncclRecv on the destination's, as one group, enqueued on the two ranks'
// exchange streams; returns without waiting.  Calls on one Exchange must not overlap (rt_render_multi's rank threads take a mutex).
void exchange_post(Exchange* e, const RowMove& move);
void exchange_wait(Exchange* e);                           // returns when every posted row has arrived
size_t exchange_release_idle();                            // destroys the idle cached communicators; returns how many sets
int exchange_library_version();                            // ncclGetVersion

}  // namespace rtamd
