// rtamd-rng-1: the counter-based integer RNG that replaces rand::thread_rng()
// (vec3.rs:98,103,112,154; material.rs:37,172; camera.rs:90; bvh.rs:61-62).
//
// Spec (shared by host and device code of the product; restated independently
// by the test oracle and pinned against it by tests/golden/rng_kat.json):
//   mix(z):  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9
//            z = (z ^ (z >> 27)) * 0x94D049BB133111EB
//            z ^ (z >> 31)                                  (SplitMix64 finaliser)
//   stream(seed, pixel, sample):
//            h  = mix(seed + 0x9E3779B97F4A7C15 * (pixel + 1))
//            s0 = mix(h    + 0xD1B54A32D192ED03 * (sample + 1))
//   next_u64: s += 0x9E3779B97F4A7C15 ; return mix(s)
//   gen::<f64>()       = (next_u64 >> 11) * 2^-53            in [0,1)
//   gen_range(lo..hi)  = lo + (hi - lo) * gen::<f64>()
//   gen_range(0..3)    = ((next_u64 >> 32) * 3) >> 32
// pixel = y * width + x of the FULL frame, sample = index in 0..spp: the value of
// a sample never depends on how the image is tiled, chunked or spread over GPUs.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rtamd {

struct Rng {
    uint64_t s;
    static RT_HD uint64_t mix(uint64_t z) {
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    RT_HD void seed_stream(uint64_t seed, uint64_t pixel, uint64_t sample) {
        uint64_t h = mix(seed + 0x9E3779B97F4A7C15ULL * (pixel + 1));
        s = mix(h + 0xD1B54A32D192ED03ULL * (sample + 1));
    }
    RT_HD uint64_t next_u64() {
        s += 0x9E3779B97F4A7C15ULL;
        return mix(s);
    }
    RT_HD double gen_f64() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }
    RT_HD double gen_range(double lo, double hi) { return lo + (hi - lo) * gen_f64(); }
    RT_HD uint32_t gen_below3() { return (uint32_t)(((next_u64() >> 32) * 3ULL) >> 32); }
};

// key of the BVHNode::new split-axis stream: stream(bvh_seed, RT_BVH_STREAM_KEY, 0)
static const uint64_t RT_BVH_STREAM_KEY = 0xB7E151628AED2A6AULL;

}  // namespace rtamd
