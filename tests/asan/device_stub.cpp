// TEST-ONLY stub of csrc/device/device.h so that the HOST half of librtamd (builders, loaders, flattener, accel builder,
// PNG writer) can be linked with -fsanitize=address,undefined and exercised on a GPU-less box (GPU ASan is not
// available on this pool).  Every device entry point reports RT_ERR_NO_DEVICE; nothing here is part of the product.
#include "device/device.h"
namespace rtamd {
static void none() { throw RtError(RT_ERR_NO_DEVICE, "sanitizer build: host code only"); }
void render_tiles(const rt_scene&, const CameraDev&, const RenderPlan&, double*, void*, rt_stats*) { none(); }
void render_sppm(const rt_scene&, const CameraDev&, RenderPlan, const rt_sppm_config&, double*, double*, void*, rt_stats*, uint64_t*) { none(); }
void assemble_frame(const RenderPlan&, const double*, int64_t, double*, void*) { none(); }
void debug_rng_device(uint64_t, uint64_t, uint64_t, int, uint64_t*) { none(); }
void debug_rng_floats_device(uint64_t, uint64_t, uint64_t, int, double, double, double*, double*) { none(); }
void debug_math_device(int, size_t, const double*, const double*, double*) { none(); }
void debug_hit_device(const rt_scene&, int, size_t, const double*, double, double, double*) { none(); }
int device_count() { return 0; }
void* dev_alloc(size_t) { none(); return nullptr; }
void dev_free(void*) {}
void dev_copy_to_host(void*, const void*, size_t) { none(); }
void dev_set_device(int) { none(); }
void free_device_copies(rt_scene&) {}
size_t release_workspaces() { return 0; }
}  // namespace rtamd
