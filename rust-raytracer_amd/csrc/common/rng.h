// rtamd-rng-3: the counter-based integer RNG that replaces rand::thread_rng()
// (vec3.rs:98,103,112,154; material.rs:37,172; camera.rs:90; bvh.rs:61-62).
//
// Spec (shared by host and device code of the product; restated independently
// by the test oracle and pinned against it by tests/golden/rng_kat.json):
//   mix(z):  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9
//            z = (z ^ (z >> 27)) * 0x94D049BB133111EB
//            z ^ (z >> 31)                                  (SplitMix64 finaliser)
//   stream(seed, pixel, sample):
//            h = mix(seed + 0x9E3779B97F4A7C15 * (pixel + 1))
//            s = mix(h    + 0xD1B54A32D192ED03 * (sample + 1)) ;  s == 0 -> 0x9E3779B97F4A7C15
//            state (s0, s1) = (low, high) 32 bits of s
//   one step (the xoroshiro64 engine of Blackman & Vigna 2018) yields 64 bits, a bijection of the 64-bit state:
//            hi = rotl(s0 * 0x9E3779BB, 5) * 5               (the `**` scrambler: xoroshiro64**'s output)
//            lo = s0 + s1                                    (the `+` scrambler; only its upper 20-21 bits are ever used)
//            s1 ^= s0 ; s0 = rotl(s0, 26) ^ s1 ^ (s1 << 9) ; s1 = rotl(s1, 13)
//   next_u64 = hi << 32 | lo  of ONE step ;  next_u32 = hi of one step
//            (for a fixed s0, i.e. a fixed hi, lo runs through all 2^32 values with s1: over the period every 64-bit value
//            but one appears exactly once.  The sum's LOW bits are linear and weak -- they are shifted out by every consumer.)
//   gen::<f64>()       = (next_u64 >> 11) * 2^-53            in [0,1), 53 bits: rand 0.8.4 `Standard` for f64
//   gen_range(lo..hi)  = v * (hi - lo) + lo,  v = f64::from_bits(0x3FF0.. | next_u64 >> 12) - 1.0   (52 bits, multiply then add):
//                        rand 0.8.4 `UniformFloat::sample_single` (also `Uniform::sample` of WeightedIndex's f64 weights).  Its retry
//                        (`res >= hi`: rounding reached the open end) cannot happen at the reference's call sites -- lo is -1 or 0
//                        (vec3.rs:118-119,156; light.rs:150,222): (1 - 2^-52) * 2 - 1 = 1 - 2^-51 exactly, and v * hi < hi for lo = 0 --
//                        so it is not restated.
//   gen_range(0..3)    = (next_u32 * 3) >> 32                (host only: BVHNode::new's axis, D3)
// (rtamd-rng-2, rounds 2-3, drew gen::<f64>() as next_u32 * 2^-32: 2^21 times coarser than the reference's uniforms.  Two
// xoroshiro64** steps per number cost the headline 4.4 % (2 736 against 2 858 Msamples/s): the second scrambler on the same step
// costs one instruction instead of nine.)
// pixel = y * width + x of the FULL frame, sample = index in 0..spp: the value of
// a sample never depends on how the image is tiled, chunked or spread over GPUs.
// (rtamd-rng-1, round 1, drew every number through the SplitMix64 finaliser: two 64-bit multiplies = eight quarter-rate
// 32-bit multiplies per draw on CDNA, 11 % of the path tracer's issue cycles; the stream seeding keeps it, once per path.)
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
        if (s == 0) s = 0x9E3779B97F4A7C15ULL;  // the all-zero state is xoroshiro's fixed point
    }
    static RT_HD uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
    RT_HD void step(uint32_t& hi, uint32_t& lo) {
        uint32_t s0 = (uint32_t)s, s1 = (uint32_t)(s >> 32);
        hi = rotl32(s0 * 0x9E3779BBu, 5) * 5u;
        lo = s0 + s1;
        s1 ^= s0;
        s0 = rotl32(s0, 26) ^ s1 ^ (s1 << 9);
        s1 = rotl32(s1, 13);
        s = ((uint64_t)s1 << 32) | (uint64_t)s0;
    }
    RT_HD uint32_t next_u32() {
        uint32_t hi, lo;
        step(hi, lo);
        return hi;
    }
    RT_HD uint64_t next_u64() {
        uint32_t hi, lo;
        step(hi, lo);
        return ((uint64_t)hi << 32) | (uint64_t)lo;
    }
    RT_HD double gen_f64() {  // (u64 >> 11) * 2^-53: both halves convert exactly, their sum has 53 bits
        uint32_t hi, lo;
        step(hi, lo);
        return (double)hi * (1.0 / 4294967296.0) + (double)(lo >> 11) * (1.0 / 9007199254740992.0);
    }
    RT_HD double gen_12() {  // f64::from_bits(exponent 0 | u64 >> 12) in [1, 2)
        uint32_t h, l;
        step(h, l);
        const uint64_t bits = ((uint64_t)(0x3FF00000u | (h >> 12)) << 32) | (uint64_t)((h << 20) | (l >> 12));
        double v12;
        __builtin_memcpy(&v12, &bits, sizeof(v12));
        return v12;
    }
    RT_HD double gen_range(double lo, double hi) { return (gen_12() - 1.0) * (hi - lo) + lo; }
    // gen_range(-1.0..1.0) (vec3.rs:118-119,156) in one instruction: (v - 1) * 2 + (-1) == 2 v - 3, and every operation of both
    // forms is exact (v in [1, 2) with 52 fraction bits), so the fused form returns the same number
    RT_HD double gen_range_pm1() { return __builtin_fma(gen_12(), 2.0, -3.0); }
    // gen_range(0.0..1.0) (light.rs:150): (v - 1) * 1 + 0 == v - 1
    RT_HD double gen_range_01() { return gen_12() - 1.0; }
    RT_HD uint32_t gen_below3() { return (uint32_t)(((uint64_t)next_u32() * 3ULL) >> 32); }
};

// key of the BVHNode::new split-axis stream: stream(bvh_seed, RT_BVH_STREAM_KEY, 0)
static const uint64_t RT_BVH_STREAM_KEY = 0xB7E151628AED2A6AULL;
// key of a noise texture's table stream: stream(texture seed, RT_PERLIN_STREAM_KEY, 0)  (D9)
static const uint64_t RT_PERLIN_STREAM_KEY = 0x243F6A8885A308D3ULL;

}  // namespace rtamd
