// ============================================================================
// ORACLE -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
//
// CPU restatement (C++17, f64, -ffp-contract=off) of the per-pixel radiance
// path of BlackCloud37/rust-raytracer, one function per row of SURVEY.md s8a.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this library; the product (rust-raytracer_amd/) never links, imports or
// calls anything in oracle/.
//
// PINNING STATUS
//   * Vec3 arithmetic: pinned by the reference's own 24 #[test]s
//     (raytracer/src/vec3.rs:425-564), replayed in tests/test_oracle_vec3.py.
//   * Sphere::bounding_box, AABB::surrounding_box, BVHNode::construct and the scene-file tree wiring: pinned by the
//     1 018 BVHNode boxes the upstream generator stored in data/scene_10.json / scene_500.json
//     (tests/test_reference_bbox_pins.py: reproduced bit for bit in f32 arithmetic).
//   * Everything else (intersection, BVH traversal, scatter, camera, tonemap, images):
//     PARITY UNPINNED.  The reference cannot be compiled here (no Rust
//     toolchain, 13 un-vendored crates), is not seedable (rand::thread_rng
//     everywhere) and its tests hold no golden vector for this path
//     (SURVEY.md s8c).  These functions follow the cited reference lines
//     operation by operation and are additionally checked by analytic
//     known-answer tests authored in this repo (tests/test_oracle_kat.py).
//
// DELIBERATE DIVERGENCES FROM THE REFERENCE (all documented in DESIGN.md)
//   D1. RNG: rand::thread_rng() -> counter-based stream keyed by (seed, pixel,
//       sample) (spec rtamd-rng-3: SplitMix64-hashed key, xoroshiro64** draws);
//       draws inside a sample are sequential in the reference's call order.
//   D2. sample_ray: on a Diffuse interaction the path CONTINUES
//       (throughput *= attenuation; ray = scattered), i.e. the two lines the
//       reference author commented out at photon_mapper.rs:346-347, instead
//       of the SPPM photon-map lookup at :349-351.
//   D3. BVHNode::new split axis comes from a seeded stream, not thread_rng.
//   D4. Rust panics become an error flag (ORC_ERR_*), never an abort.
//   D8. ConstantMedium::hit's `.ln()` is the deterministic algorithm rtamd-ln-1 (det_ln below; < 1 ulp from libm's log),
//       and the phase function is the reference's commented-out Isotropic (material.rs:213-231) classified as a
//       pass-through (Specular) interaction.
//
// Instrumentation: every AABB / primitive / transform test is counted so the
// "algorithmic bytes per sample" of SURVEY.md s8d can be computed in the
// reference's own traversal order.
// ============================================================================
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <vector>

namespace orc {

static const double PI = 3.14159265358979323846264338327950288;          // std::f64::consts::PI
static const double FRAC_1_PI = 0.318309886183790671537767526745028724;  // std::f64::consts::FRAC_1_PI
static const double INF = std::numeric_limits<double>::infinity();

struct UnitZero : std::runtime_error {
    UnitZero() : std::runtime_error("unitizing zero vector") {}
};

// ----------------------------------------------------------------------------
// Vec3 -- raytracer/src/vec3.rs:14-19 (struct), :21-185 (methods), :247-424 (ops)
// ----------------------------------------------------------------------------
struct Vec3 {
    double x, y, z;
    Vec3() : x(0), y(0), z(0) {}
    Vec3(double a, double b, double c) : x(a), y(b), z(c) {}
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }  // vec3.rs:211-221
};
static inline Vec3 v_add(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }  // :247-257
static inline Vec3 v_adds(Vec3 a, double s) { return Vec3(a.x + s, a.y + s, a.z + s); }     // :259-269
static inline Vec3 v_sub(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }  // :291-301
static inline Vec3 v_subs(Vec3 a, double s) { return Vec3(a.x - s, a.y - s, a.z - s); }     // :303-313
// Vec3 * Vec3 is the DOT product (quirk Q1), vec3.rs:335-341; left-assoc sum.
static inline double v_dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline Vec3 v_muls(Vec3 a, double s) { return Vec3(a.x * s, a.y * s, a.z * s); }  // :343-365 (both orders)
static inline Vec3 v_divs(Vec3 a, double s) { return Vec3(a.x / s, a.y / s, a.z / s); }  // :377-397 true divides
static inline Vec3 v_neg(Vec3 a) { return Vec3(-a.x, -a.y, -a.z); }                      // :410-418
static inline Vec3 v_elemul(Vec3 a, Vec3 b) { return Vec3(a.x * b.x, a.y * b.y, a.z * b.z); }  // :65-71
static inline Vec3 v_cross(Vec3 a, Vec3 b) {                                                    // :73-79
    return Vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline double v_sqlen(Vec3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }  // :61-63
static inline double v_len(Vec3 a) { return std::sqrt(v_sqlen(a)); }                // :81-83
static inline Vec3 v_unit(Vec3 a) {                                                 // :85-90 (panic -> throw, D4)
    double l = v_len(a);
    if (l == 0.) throw UnitZero();
    return v_divs(a, l);
}
static inline bool v_near_zero(Vec3 a) {  // :92-95
    const double S = 1e-8;
    return (std::fabs(a.x) < S) && (std::fabs(a.y) < S) && (std::fabs(a.z) < S);
}
static inline double v_max(Vec3 a) { return std::fmax(std::fmax(a.x, a.y), a.z); }  // :57-59

// ----------------------------------------------------------------------------
// RNG (divergence D1).  Spec "rtamd-rng-3" -- restated independently in the
// product (rust-raytracer_amd/csrc/common/rng.h); pinned against each other by
// tests/golden/rng_kat.json.
//   stream key (SplitMix64 finaliser; Steele, Lea, Flood 2014; public-domain reference by Vigna):
//                state = mix(mix(seed + G*(pixel+1)) + H*(sample+1)), G if that is 0; (s0, s1) = its (low, high) halves
//   generator:   the xoroshiro64 engine (Blackman & Vigna 2018; public-domain reference); ONE step yields 64 bits:
//                upper half = xoroshiro64**'s output rotl(s0 * 0x9E3779BB, 5) * 5, lower half = s0 + s1 (of the state before the
//                step; the map state -> 64 bits is a bijection; every consumer shifts the sum's weak low bits out)
//   next_u64     = that; next_u32 = its upper half
//   The float conversions are those of the reference's `rand 0.8.4` (Cargo.lock:836-837; the crate is not under /root/reference,
//   restated from its published algorithm):
//   gen::<f64>() = (next_u64 >> 11) as f64 * 2^-53                  `Standard`, 53 bits, [0,1)       (distributions/float.rs)
//   gen_range(lo..hi) = `UniformFloat::sample_single`: v = from_bits(exponent 0 | next_u64 >> 12) - 1.0  (52 bits);
//                  res = v * (hi - lo) + lo; return it if res < hi, otherwise (rounding reached the open end) draw again with
//                  `scale` one ulp smaller.  At the reference's call sites (lo = -1 or 0) the retry cannot happen.
//   gen_range(0..3)   = (u32 * 3) >> 32            (BVHNode::new's axis only, D3; rand's widening-multiply-with-rejection is not restated)
// ----------------------------------------------------------------------------
struct Rng {
    uint32_t s0, s1;
    uint64_t draws = 0;  // random NUMBERS drawn so far (a gen::<f64>() / gen_range is one; orc_hit_rng reports it)
    static inline uint64_t mix(uint64_t z) {
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    static inline uint32_t rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
    Rng() : s0(0x7F4A7C15u), s1(0x9E3779B9u) {}
    Rng(uint64_t seed, uint64_t pixel, uint64_t sample) {
        uint64_t h = mix(seed + 0x9E3779B97F4A7C15ULL * (pixel + 1));
        uint64_t s = mix(h + 0xD1B54A32D192ED03ULL * (sample + 1));
        if (s == 0) s = 0x9E3779B97F4A7C15ULL;
        s0 = (uint32_t)s;
        s1 = (uint32_t)(s >> 32);
    }
    inline uint64_t next_u64() {
        const uint64_t out = ((uint64_t)(rotl(s0 * 0x9E3779BBu, 5) * 5u) << 32) | (uint64_t)(uint32_t)(s0 + s1);
        s1 ^= s0;
        s0 = rotl(s0, 26) ^ s1 ^ (s1 << 9);
        s1 = rotl(s1, 13);
        return out;
    }
    inline uint32_t next_u32() { return (uint32_t)(next_u64() >> 32); }
    inline double gen_f64() {
        draws++;
        return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0);
    }
    inline double gen_range(double lo, double hi) {
        draws++;
        double scale = hi - lo;
        for (;;) {
            const uint64_t bits = (uint64_t(1023) << 52) | (next_u64() >> 12);  // into_float_with_exponent(0): [1, 2)
            double value1_2;
            std::memcpy(&value1_2, &bits, sizeof(value1_2));
            const double value0_1 = value1_2 - 1.0;
            const double res = value0_1 * scale + lo;
            if (res < hi) return res;
            scale = std::nextafter(scale, 0.0);  // decrease_masked
        }
    }
    inline uint32_t gen_below3() {  // gen_range(0..3)
        draws++;
        return (uint32_t)(((uint64_t)next_u32() * 3ULL) >> 32);
    }
};

// vec3.rs:111-129 -- Marsaglia; returns a point ON the unit sphere (quirk Q3).
static inline Vec3 random_in_unit_sphere(Rng& rng) {
    double u, v, r2;
    for (;;) {
        u = rng.gen_range(-1., 1.);
        v = rng.gen_range(-1., 1.);
        r2 = u * u + v * v;
        if (r2 <= 1.) break;
    }
    return Vec3(2. * u * std::sqrt(1. - r2), 2. * v * std::sqrt(1. - r2), 1. - 2. * r2);
}
static inline Vec3 random_unit_vector(Rng& rng) { return v_unit(random_in_unit_sphere(rng)); }  // :140-142
static inline Vec3 random_in_hemisphere(Rng& rng, Vec3 n) {                                    // :144-151
    Vec3 s = random_in_unit_sphere(rng);
    return (v_dot(s, n) > 0.0) ? s : v_neg(s);
}
static inline Vec3 random_in_unit_disk(Rng& rng) {  // :153-162
    for (;;) {
        double a = rng.gen_range(-1.0, 1.0);
        double b = rng.gen_range(-1.0, 1.0);
        Vec3 p(a, b, 0.);
        if (v_sqlen(p) >= 1.) continue;
        return p;
    }
}
static inline Vec3 reflect(Vec3 v_in, Vec3 n) {  // :163-165   v - (2*(v.n))*n
    return v_sub(v_in, v_muls(n, 2. * v_dot(v_in, n)));
}
static inline Vec3 refract(Vec3 uv, Vec3 n, double etai_over_etat) {  // :167-172
    double cos_theta = std::fmin(v_dot(v_neg(uv), n), 1.0);
    Vec3 r_out_perp = v_muls(v_add(uv, v_muls(n, cos_theta)), etai_over_etat);
    Vec3 r_out_parallel = v_muls(n, -std::sqrt(std::fabs(1.0 - v_sqlen(r_out_perp))));
    return v_add(r_out_perp, r_out_parallel);
}

// 4x4 matrix, row-major; stands in for nalgebra::Matrix4<f64> (transform.rs, vec3.rs:174-184)
struct Mat4 {
    double m[4][4];
};
static Mat4 mat_mul(const Mat4& a, const Mat4& b) {
    Mat4 c;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double acc = a.m[i][0] * b.m[0][j];
            for (int k = 1; k < 4; k++) acc = acc + a.m[i][k] * b.m[k][j];
            c.m[i][j] = acc;
        }
    return c;
}
// General 4x4 inverse by cofactors (the MESA/GLU formula nalgebra's 4x4
// try_inverse specialisation uses); returns false when det == 0.
static bool mat_inverse(const Mat4& a, Mat4& out) {
    // column-major flat view, as in the GLU routine
    double m[16], inv[16];
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) m[c * 4 + r] = a.m[r][c];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0.) return false;
    double inv_det = 1.0 / det;
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) out.m[r][c] = inv[c * 4 + r] * inv_det;
    return true;
}
static inline Vec3 transform_point(Vec3 p, const Mat4& t) {  // vec3.rs:174-178 (w = 1)
    return Vec3(t.m[0][0] * p.x + t.m[0][1] * p.y + t.m[0][2] * p.z + t.m[0][3] * 1.,
                t.m[1][0] * p.x + t.m[1][1] * p.y + t.m[1][2] * p.z + t.m[1][3] * 1.,
                t.m[2][0] * p.x + t.m[2][1] * p.y + t.m[2][2] * p.z + t.m[2][3] * 1.);
}
static inline Vec3 transform_dir(Vec3 p, const Mat4& t) {  // vec3.rs:180-184 (w = 0)
    return Vec3(t.m[0][0] * p.x + t.m[0][1] * p.y + t.m[0][2] * p.z + t.m[0][3] * 0.,
                t.m[1][0] * p.x + t.m[1][1] * p.y + t.m[1][2] * p.z + t.m[1][3] * 0.,
                t.m[2][0] * p.x + t.m[2][1] * p.y + t.m[2][2] * p.z + t.m[2][3] * 0.);
}

// ----------------------------------------------------------------------------
// Ray -- raytracer/src/ray.rs:3-15
// ----------------------------------------------------------------------------
struct Ray {
    Vec3 orig, dir;
    double time = 0.;  // D9 (book-2 extension; the reference's Ray has no time, ray.rs:3-6): when the sample was taken, within the camera's shutter
    Vec3 at(double t) const { return v_add(orig, v_muls(dir, t)); }
};

// Test counters (instrumentation only; feeds SURVEY s8d B_alg).
struct Counters {
    uint64_t n_aabb = 0, n_sphere = 0, n_rect = 0, n_tri = 0, n_xform = 0;
    uint64_t n_segments = 0;  // World::hit calls
    uint64_t n_samples = 0;
    void add(const Counters& o) {
        n_aabb += o.n_aabb; n_sphere += o.n_sphere; n_rect += o.n_rect; n_tri += o.n_tri;
        n_xform += o.n_xform; n_segments += o.n_segments; n_samples += o.n_samples;
    }
};
// Per-thread traversal context: RNG stream of the current sample + counters.
struct Ctx {
    Rng rng;
    Counters cnt;
};

struct Material;

// ----------------------------------------------------------------------------
// HitRecord -- raytracer/src/objects/hit.rs:7-48
// ----------------------------------------------------------------------------
struct HitRecord {
    Vec3 p, normal;
    double t = 0;
    bool front_face = false;
    const Material* mat = nullptr;
    double u = 0, v = 0;
    int prim_id = -1;  // instrumentation: id of the primitive object that produced the hit

    // hit.rs:16-39: p = r.at(t); face test on the un-normalised dir; normal re-normalised (Q6).
    static HitRecord make(double t, Vec3 outward_normal, const Ray& r, const Material* mat, double u, double v, int prim) {
        HitRecord h;
        h.p = r.at(t);
        h.front_face = v_dot(r.dir, outward_normal) < 0.;
        Vec3 n = h.front_face ? outward_normal : v_neg(outward_normal);
        h.normal = v_unit(n);
        h.t = t;
        h.mat = mat;
        h.u = u;
        h.v = v;
        h.prim_id = prim;
        return h;
    }
    // hit.rs:41-48
    void set_face_normal(const Ray& r, Vec3 outward_normal) {
        front_face = v_dot(r.dir, outward_normal) < 0.;
        normal = front_face ? v_unit(outward_normal) : v_neg(v_unit(outward_normal));
    }
};

// ----------------------------------------------------------------------------
// Textures -- raytracer/src/material.rs:18-20, 48-84
// ----------------------------------------------------------------------------
struct Texture {
    virtual ~Texture() {}
    virtual Vec3 get_color(const HitRecord& rec) const = 0;
};
struct ConstantTexture : Texture {  // material.rs:48,52-56
    Vec3 c;
    explicit ConstantTexture(Vec3 c_) : c(c_) {}
    Vec3 get_color(const HitRecord&) const override { return c; }
};
struct CheckerTexture : Texture {  // material.rs:49,58-68 ; .0 when sines < 0
    const Texture *t0, *t1;
    CheckerTexture(const Texture* a, const Texture* b) : t0(a), t1(b) {}
    Vec3 get_color(const HitRecord& rec) const override {
        Vec3 p = rec.p;
        double sines = std::sin(10. * p.x) * std::sin(10. * p.y) * std::sin(10. * p.z);
        return (sines < 0.) ? t0->get_color(rec) : t1->get_color(rec);
    }
};
struct ImageTexture : Texture {  // material.rs:50,70-84 ; Q11: x==width at u==1 panics in Rust -> clamped here
    int w, h;
    std::vector<uint8_t> rgb;
    ImageTexture(int w_, int h_, const uint8_t* d) : w(w_), h(h_), rgb(d, d + (size_t)w_ * h_ * 3) {}
    Vec3 get_color(const HitRecord& rec) const override {
        double u = rec.u, v = rec.v;
        u = std::fmin(std::fmax(u, 0.), 1.);
        v = 1. - std::fmin(std::fmax(v, 0.), 1.);
        long x = (long)std::floor((double)w * u);
        long y = (long)std::floor((double)h * v);
        if (x > w - 1) x = w - 1;  // Q11 clamp (documented divergence: Rust would panic)
        if (y > h - 1) y = h - 1;
        const uint8_t* px = &rgb[((size_t)y * w + x) * 3];
        return Vec3(px[0] / 255., px[1] / 255., px[2] / 255.);  // vec3.rs:233-238
    }
};

// ----------------------------------------------------------------------------
// D9: the two book-2 ("Ray Tracing: The Next Week") features BASELINE's config C5 names and the reference has no code for --
// motion blur (a time on every ray, a sphere whose centre moves linearly during the shutter) and a Perlin-noise texture.  Parity
// exists only against this restatement.  Restated from the book's published algorithms, with two choices of
// this build: the tables come from the seeded stream (seed, PERLIN_KEY, 0), and the marble texture's sin() is the deterministic
// rtamd-sin-1 below (libm's last bit differs between glibc and the device library; the value feeds a colour, so oracle and
// kernel must agree to the bit).
// ----------------------------------------------------------------------------
// rtamd-sin-1: sin(x) for |x| < 2^19 * pi/2 by Cody-Waite reduction x = n * pi/2 + (y0 + y1) with the two-step constants published
// for fdlibm's e_rem_pio2.c (always two steps: good to 118 bits) and the degree-13 / degree-14 kernels of k_sin.c / k_cos.c in their
// plain forms; IEEE + - * only, no fused operations, so every implementation returns the same bits.  |x| beyond the range -> 0.
static double det_sin(double x) {
    static const double INVPIO2 = 6.36619772367581382433e-01, PIO2_1 = 1.57079632673412561417e+00, PIO2_2 = 6.07710050630396597660e-11,
                        PIO2_2T = 2.02226624879595063154e-21;
    static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                        S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                        C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    if (!(std::fabs(x) < 823549.0)) return 0.0;  // (also NaN)
    const double t = std::fabs(x);
    const int n = (int)(t * INVPIO2 + 0.5);
    const double fn = (double)n;
    const double r1 = t - fn * PIO2_1;
    const double w2 = fn * PIO2_2;
    const double r2 = r1 - w2;
    const double w = fn * PIO2_2T - ((r1 - r2) - w2);
    const double y0 = r2 - w, y1 = (r2 - y0) - w;
    const double z = y0 * y0;
    double res;
    if ((n & 1) == 0) {  // +- sin(y0 + y1)
        const double v = z * y0;
        const double rr = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
        res = y0 - ((z * (0.5 * y1 - v * rr) - y1) - v * S1);
    } else {  // +- cos(y0 + y1)
        const double rr = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
        res = 1.0 - (0.5 * z - (z * rr - y0 * y1));
    }
    if (n & 2) res = -res;
    return (x < 0.) ? -res : res;
}

static const uint64_t PERLIN_KEY = 0x243F6A8885A308D3ULL;  // stream (seed, PERLIN_KEY, 0) fills a noise texture's tables
// perlin / noise_texture of the book: 256 random unit vectors, three permutations, trilinear interpolation with Hermite smoothing,
// 7 octaves of turbulence, marble = 0.5 (1 + sin(scale z + 10 turb(p)))
struct NoiseTexture : Texture {
    double scale;
    Vec3 ranvec[256];
    int perm[3][256];
    NoiseTexture(double scale_, uint64_t seed) : scale(scale_) {
        Rng rng(seed, PERLIN_KEY, 0);
        for (int i = 0; i < 256; i++) {
            const double x = rng.gen_range(-1., 1.), y = rng.gen_range(-1., 1.), z = rng.gen_range(-1., 1.);
            ranvec[i] = v_unit(Vec3(x, y, z));  // unit_vector(vec3::random(-1, 1))
        }
        for (int a = 0; a < 3; a++) {  // perlin_generate_perm: identity, then permute from the top
            for (int i = 0; i < 256; i++) perm[a][i] = i;
            for (int i = 255; i > 0; i--) {
                const int target = (int)(((uint64_t)rng.next_u32() * (uint64_t)(i + 1)) >> 32);  // random_int(0, i)
                std::swap(perm[a][i], perm[a][target]);
            }
        }
    }
    double noise(Vec3 p) const {
        const double fx = std::floor(p.x), fy = std::floor(p.y), fz = std::floor(p.z);
        const double u = p.x - fx, v = p.y - fy, w = p.z - fz;
        const int i = (int)fx, j = (int)fy, k = (int)fz;
        const double uu = u * u * (3. - 2. * u), vv = v * v * (3. - 2. * v), ww = w * w * (3. - 2. * w);
        double accum = 0.;
        for (int di = 0; di < 2; di++)
            for (int dj = 0; dj < 2; dj++)
                for (int dk = 0; dk < 2; dk++) {
                    const Vec3 c = ranvec[perm[0][(i + di) & 255] ^ perm[1][(j + dj) & 255] ^ perm[2][(k + dk) & 255]];
                    const Vec3 weight_v(u - di, v - dj, w - dk);
                    accum += (di * uu + (1 - di) * (1. - uu)) * (dj * vv + (1 - dj) * (1. - vv)) * (dk * ww + (1 - dk) * (1. - ww)) * v_dot(c, weight_v);
                }
        return accum;
    }
    double turb(Vec3 p) const {
        double accum = 0., weight = 1.;
        Vec3 temp_p = p;
        for (int i = 0; i < 7; i++) {
            accum += weight * noise(temp_p);
            weight *= 0.5;
            temp_p = v_muls(temp_p, 2.);
        }
        return std::fabs(accum);
    }
    double value(Vec3 p) const { return 0.5 * (1. + det_sin(scale * p.z + 10. * turb(p))); }
    Vec3 get_color(const HitRecord& rec) const override {
        const double m = value(rec.p);
        return Vec3(1. * m, 1. * m, 1. * m);  // color(1,1,1) * 0.5 * (1 + sin(..))
    }
};

// ----------------------------------------------------------------------------
// Materials -- raytracer/src/material.rs:10-46 (trait), :86-212 (impls)
// ----------------------------------------------------------------------------
enum Interaction { Diffuse, Specular, Absorb, Reflect, Refract };  // material.rs:10-16
struct ScatterResult {
    Interaction kind;
    bool has_ray, has_att;
    Ray ray;
    Vec3 att;
};
struct Material {
    virtual ~Material() {}
    virtual Vec3 bsdf(Vec3 r_dir, const HitRecord& rec) const = 0;
    virtual ScatterResult scatter(const Ray& r, const HitRecord& rec, Ctx& cx) const = 0;
    virtual Vec3 emitted(const HitRecord&) const { return Vec3(0, 0, 0); }  // material.rs:24-26
};
// material.rs:92-98
static inline Vec3 scattered_direction(Vec3 n, Ctx& cx) {
    Vec3 d = v_add(n, random_unit_vector(cx.rng));
    if (v_near_zero(d)) d = n;
    return d;
}
struct Lambertian : Material {  // material.rs:88-113
    const Texture* albedo;
    explicit Lambertian(const Texture* a) : albedo(a) {}
    Vec3 bsdf(Vec3, const HitRecord& rec) const override { return albedo->get_color(rec); }
    ScatterResult scatter(const Ray& r, const HitRecord& rec, Ctx& cx) const override {
        Ray s{rec.p, scattered_direction(rec.normal, cx)};
        return ScatterResult{Diffuse, true, true, s, bsdf(r.dir, rec)};
    }
};
struct Metal : Material {  // material.rs:115-139 ; RNG drawn even for fuzz == 0 (Q4); Absorb when pointing inward (Q15)
    const Texture* albedo;
    double fuzz;
    Metal(const Texture* a, double f) : albedo(a), fuzz(f) {}
    Vec3 bsdf(Vec3, const HitRecord& rec) const override { return albedo->get_color(rec); }
    ScatterResult scatter(const Ray& r, const HitRecord& rec, Ctx& cx) const override {
        Vec3 reflected = reflect(v_unit(r.dir), rec.normal);
        Ray s{rec.p, v_add(reflected, v_muls(random_in_unit_sphere(cx.rng), fuzz))};
        if (v_dot(s.dir, rec.normal) > 0.) return ScatterResult{Specular, true, true, s, bsdf(r.dir, rec)};
        return ScatterResult{Absorb, false, false, Ray{}, Vec3()};
    }
};
struct Dielectric : Material {  // material.rs:141-188
    double ir;
    const Texture* albedo;
    Dielectric(double i, const Texture* a) : ir(i), albedo(a) {}
    static double reflectance(double cosine, double ref_idx) {  // :150-154 Schlick; powi(2), powi(5)
        double q = (1. - ref_idx) / (1. + ref_idx);
        double r0 = q * q;
        double b = 1. - cosine;
        double b2 = b * b;
        double b4 = b2 * b2;
        double b5 = b * b4;  // powi(5) = a * (a^2)^2 (compiler-rt __powidf2 / LLVM powi expansion order)
        return r0 + (1. - r0) * b5;
    }
    Vec3 bsdf(Vec3, const HitRecord& rec) const override { return albedo->get_color(rec); }
    ScatterResult scatter(const Ray& r, const HitRecord& rec, Ctx& cx) const override {
        Vec3 attenuation = bsdf(r.dir, rec);
        double refraction_ratio = rec.front_face ? (1.0 / ir) : ir;
        Vec3 unit_direction = v_unit(r.dir);
        double cos_theta = std::fmin(v_dot(v_neg(unit_direction), rec.normal), 1.0);
        double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
        bool cannot_refract = refraction_ratio * sin_theta > 1.0;
        Vec3 direction;
        Interaction kind;
        // short-circuit: the random number is consumed only when refraction is possible (Q16)
        if (cannot_refract || reflectance(cos_theta, refraction_ratio) > cx.rng.gen_f64()) {
            direction = reflect(unit_direction, rec.normal);
            kind = Reflect;
        } else {
            direction = refract(unit_direction, rec.normal, refraction_ratio);
            kind = Refract;
        }
        return ScatterResult{kind, true, true, Ray{rec.p, direction}, attenuation};
    }
};
struct DiffuseLight : Material {  // material.rs:190-212 ; Q10
    const Texture* emit;
    explicit DiffuseLight(const Texture* e) : emit(e) {}
    Vec3 bsdf(Vec3, const HitRecord&) const override { return v_muls(Vec3(1, 1, 1), FRAC_1_PI); }
    ScatterResult scatter(const Ray& r, const HitRecord& rec, Ctx& cx) const override {
        Ray s{rec.p, scattered_direction(rec.normal, cx)};
        return ScatterResult{Diffuse, true, true, s, bsdf(r.dir, rec)};
    }
    Vec3 emitted(const HitRecord& rec) const override { return emit->get_color(rec); }
};

// material.rs:213-231 (commented out in the reference): Isotropic phase function of a ConstantMedium.
// scatter = (albedo colour, Ray(rec.p, random_in_unit_sphere())).  The old Option<(Vec3, Ray)> API has no Interaction;
// it is classified Specular here (the path simply continues; no photon store, no light sampling at a volume event).
struct Isotropic : Material {
    const Texture* albedo;
    explicit Isotropic(const Texture* a) : albedo(a) {}
    Vec3 bsdf(Vec3, const HitRecord& rec) const override { return albedo->get_color(rec); }
    ScatterResult scatter(const Ray& r, const HitRecord& rec, Ctx& cx) const override {
        Ray s{rec.p, random_in_unit_sphere(cx.rng)};
        return ScatterResult{Specular, true, true, s, bsdf(r.dir, rec)};
    }
};

// rtamd-ln-1 (D8): natural logarithm by argument reduction x = 2^k (1 + f), sqrt(2)/2 < 1 + f < sqrt(2), s = f / (2 + f),
// log(1 + f) = f - (f^2/2 - s (f^2/2 + R(s^2))) with the degree-14 polynomial R published for fdlibm's e_log.c.
// Restated here independently of the product's csrc/common/detlog.h; tests/test_medium.py compares the two and numpy.
static double det_ln(double x) {
    static const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    static const double LG[7] = {6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01, 2.222219843214978396e-01,
                                 1.818357216161805012e-01, 1.531383769920937332e-01, 1.479819860511658591e-01};
    if (x == 0.0) return -INF;
    if (!(x > 0.0)) return std::numeric_limits<double>::quiet_NaN();
    if (x == INF) return x;
    int e = 0;
    if (x < 2.2250738585072014e-308) {  // subnormal
        x *= 18014398509481984.0;  // 2^54
        e = -54;
    }
    int ex;
    double m = 2.0 * std::frexp(x, &ex);  // m in [1, 2), x = m * 2^(ex - 1)
    e += ex - 1;
    if (m > 1.41421356237309504880) {
        m = m * 0.5;
        e += 1;
    }
    const double f = m - 1.0, dk = (double)e;
    const double s = f / (2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * (LG[1] + w * (LG[3] + w * LG[5]));
    const double t2 = z * (LG[0] + w * (LG[2] + w * (LG[4] + w * LG[6])));
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    return dk * LN2_HI - ((hfsq - (s * (hfsq + R) + dk * LN2_LO)) - f);
}

// ----------------------------------------------------------------------------
// AABB -- raytracer/src/objects/aabb.rs:5-45
// ----------------------------------------------------------------------------
struct AABB {
    Vec3 minimum, maximum;
    // aabb.rs:15-32 ; f64::max/min ignore NaN -> fmax/fmin (NOT std::max)
    bool hit(const Ray& r, double t_min, double t_max, Ctx& cx) const {
        cx.cnt.n_aabb++;
        double mn = t_min, mx = t_max;
        for (int a = 0; a < 3; a++) {
            double inv_d = 1.0 / r.dir[a];
            double t0 = (minimum[a] - r.orig[a]) * inv_d;
            double t1 = (maximum[a] - r.orig[a]) * inv_d;
            if (inv_d < 0.0) std::swap(t0, t1);
            mn = std::fmax(mn, t0);
            mx = std::fmin(mx, t1);
            if (mx <= mn) return false;
        }
        return true;
    }
    static AABB surrounding_box(const AABB& a, const AABB& b) {  // aabb.rs:33-45
        Vec3 small(std::fmin(a.minimum.x, b.minimum.x), std::fmin(a.minimum.y, b.minimum.y), std::fmin(a.minimum.z, b.minimum.z));
        Vec3 big(std::fmax(a.maximum.x, b.maximum.x), std::fmax(a.maximum.y, b.maximum.y), std::fmax(a.maximum.z, b.maximum.z));
        return AABB{small, big};
    }
};

// ----------------------------------------------------------------------------
// Hitable -- raytracer/src/objects/hit.rs:51-54
// ----------------------------------------------------------------------------
struct Hitable {
    int id = -1;
    virtual ~Hitable() {}
    virtual bool hit(const Ray& r, double t_min, double t_max, HitRecord& out, Ctx& cx) const = 0;
    virtual bool bounding_box(AABB& out) const = 0;
};

// impl Hitable for Vec<Arc<dyn Hitable>> -- hit.rs:56-93
struct HitableList : Hitable {
    std::vector<const Hitable*> items;
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& out, Ctx& cx) const override {
        double closest = t_max;
        bool any = false;
        HitRecord tmp;
        for (const Hitable* o : items) {
            if (o->hit(r, t_min, closest, tmp, cx)) {
                closest = tmp.t;
                out = tmp;
                any = true;
            }
        }
        return any;
    }
    bool bounding_box(AABB& out) const override {
        if (items.empty()) return false;
        bool first = true;
        AABB acc{Vec3(), Vec3()};
        for (const Hitable* o : items) {
            AABB b;
            if (!o->bounding_box(b)) return false;
            acc = first ? b : AABB::surrounding_box(acc, b);
            first = false;
        }
        out = acc;
        return true;
    }
};

// Sphere -- raytracer/src/objects/sphere.rs:8-62
struct Sphere : Hitable {
    Vec3 center;
    double radius;
    const Material* material;
    static void get_uv(Vec3 p, double& u, double& v) {  // :16-20
        double theta = std::acos(-p.y);
        double phi = std::atan2(-p.z, p.x) + PI;
        u = phi * FRAC_1_PI * 0.5;
        v = theta * FRAC_1_PI;
    }
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& out, Ctx& cx) const override {  // :24-55
        cx.cnt.n_sphere++;
        Vec3 oc = v_sub(r.orig, center);
        double a = v_sqlen(r.dir);
        double half_b = v_dot(oc, r.dir);
        double c = v_sqlen(oc) - radius * radius;
        double discriminant = half_b * half_b - a * c;  // half_b.powf(2.0)
        if (discriminant < 0.) return false;
        double sqrt_d = std::sqrt(discriminant);
        double root = (-half_b - sqrt_d) / a;
        if (!(root >= t_min && root <= t_max)) root = (-half_b + sqrt_d) / a;
        if (!(root >= t_min && root <= t_max)) return false;
        Vec3 p = r.at(root);
        Vec3 outward = v_divs(v_sub(p, center), radius);
        double u, v;
        get_uv(outward, u, v);  // computed for every candidate, as the reference does (:52)
        out = HitRecord::make(root, outward, r, material, u, v, id);
        return true;
    }
    bool bounding_box(AABB& out) const override {  // :56-61
        out = AABB{v_sub(center, Vec3(radius, radius, radius)), v_add(center, Vec3(radius, radius, radius))};
        return true;
    }
};

// D9: moving_sphere of the book: the centre moves linearly from center0 at time0 to center1 at time1; everything else is Sphere::hit
struct MovingSphere : Hitable {
    Vec3 center0, center1;
    double time0, time1, radius;
    const Material* material;
    Vec3 center(double time) const { return v_add(center0, v_muls(v_sub(center1, center0), (time - time0) / (time1 - time0))); }
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& out, Ctx& cx) const override {
        cx.cnt.n_sphere++;
        const Vec3 c = center(r.time);
        Vec3 oc = v_sub(r.orig, c);
        double a = v_sqlen(r.dir);
        double half_b = v_dot(oc, r.dir);
        double cc = v_sqlen(oc) - radius * radius;
        double discriminant = half_b * half_b - a * cc;
        if (discriminant < 0.) return false;
        double sqrt_d = std::sqrt(discriminant);
        double root = (-half_b - sqrt_d) / a;
        if (!(root >= t_min && root <= t_max)) root = (-half_b + sqrt_d) / a;
        if (!(root >= t_min && root <= t_max)) return false;
        Vec3 p = r.at(root);
        Vec3 outward = v_divs(v_sub(p, c), radius);
        double u, v;
        Sphere::get_uv(outward, u, v);
        out = HitRecord::make(root, outward, r, material, u, v, id);
        return true;
    }
    bool bounding_box(AABB& out) const override {  // surrounding_box of the boxes at time0 and time1
        const Vec3 rr(radius, radius, radius);
        out = AABB::surrounding_box(AABB{v_sub(center0, rr), v_add(center0, rr)}, AABB{v_sub(center1, rr), v_add(center1, rr)});
        return true;
    }
};

// XY/XZ/YZ rectangles -- raytracer/src/objects/rectangle.rs:7-117.
// axis = the constant axis (2: XY rect, 1: XZ rect, 0: YZ rect).
// No guard on a zero direction component: t may be NaN/inf and every reject
// comparison is then false -> Some(t = NaN); restated literally.
struct Rect : Hitable {
    int axis;
    double a0, b0, a1, b1, k;
    const Material* material;
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& out, Ctx& cx) const override {
        cx.cnt.n_rect++;
        double t = (k - r.orig[axis]) / r.dir[axis];
        if (t < t_min || t > t_max) return false;
        Vec3 p = r.at(t);
        double a, b;
        Vec3 n;
        if (axis == 2) { a = p.x; b = p.y; n = Vec3(0., 0., 1.); }       // :15-34
        else if (axis == 1) { a = p.x; b = p.z; n = Vec3(0., 1., 0.); }  // :53-72
        else { a = p.y; b = p.z; n = Vec3(1., 0., 0.); }                 // :90-109
        if (a < a0 || a > a1 || b < b0 || b > b1) return false;
        out = HitRecord::make(t, n, r, material, (a - a0) / (a1 - a0), (b - b0) / (b1 - b0), id);
        return true;
    }
    bool bounding_box(AABB& out) const override {  // :35-41, :73-79, :110-116
        const double BIAS = 0.0001;
        if (axis == 2) out = AABB{Vec3(a0, b0, k - BIAS), Vec3(a1, b1, k + BIAS)};
        else if (axis == 1) out = AABB{Vec3(a0, k - BIAS, b0), Vec3(a1, k + BIAS, b1)};
        else out = AABB{Vec3(k - BIAS, a0, b0), Vec3(k + BIAS, a1, b1)};
        return true;
    }
};

// Cube -- raytracer/src/objects/cube.rs:9-70 : 6 rects scanned as a list; bbox = (min,max)
struct Cube : Hitable {
    Vec3 box_min, box_max;
    HitableList sides;
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& out, Ctx& cx) const override {
        return sides.hit(r, t_min, t_max, out, cx);
    }
    bool bounding_box(AABB& out) const override {
        out = AABB{box_min, box_max};
        return true;
    }
};

// BVHNode -- raytracer/src/objects/bvh.rs:29-106
struct BVHNode : Hitable {
    const Hitable *left = nullptr, *right = nullptr;
    AABB box;
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& out, Ctx& cx) const override {  // :86-102
        if (!box.hit(r, t_min, t_max, cx)) return false;
        HitRecord hl, hr;
        bool l = left->hit(r, t_min, t_max, hl, cx);
        double left_t_max = l ? hl.t : t_max;
        bool rr = right->hit(r, t_min, left_t_max, hr, cx);
        if (rr) { out = hr; return true; }
        if (l) { out = hl; return true; }
        return false;
    }
    bool bounding_box(AABB& out) const override {
        out = box;
        return true;
    }
};

// Triangle -- raytracer/src/objects/mesh.rs:8-142
struct MeshData {
    std::vector<Vec3> positions, normals;
};
struct Triangle : Hitable {
    size_t a, b, c;
    const MeshData* md;
    AABB box;
    const Material* material;
    void init_box() {  // mesh.rs:29-42 : +-0.1 padding in object space (Q9)
        const Vec3 &pa = md->positions[a], &pb = md->positions[b], &pc = md->positions[c];
        Vec3 mx(std::fmax(std::fmax(pa.x, pb.x), pc.x) + 0.1, std::fmax(std::fmax(pa.y, pb.y), pc.y) + 0.1, std::fmax(std::fmax(pa.z, pb.z), pc.z) + 0.1);
        Vec3 mn(std::fmin(std::fmin(pa.x, pb.x), pc.x) - 0.1, std::fmin(std::fmin(pa.y, pb.y), pc.y) - 0.1, std::fmin(std::fmin(pa.z, pb.z), pc.z) - 0.1);
        box = AABB{mn, mx};
    }
    bool hit(const Ray& ray, double t_min, double t_max, HitRecord& out, Ctx& cx) const override {  // :57-137
        cx.cnt.n_tri++;
        const Vec3 &pa = md->positions[a], &pb = md->positions[b], &pc = md->positions[c];
        const Vec3 &na = md->normals[a], &nb = md->normals[b], &nc = md->normals[c];
        Vec3 e0 = v_sub(pb, pa), e1 = v_sub(pc, pa);
        Vec3 s0 = v_cross(ray.dir, e1);
        double dd = v_dot(s0, e0);
        if (dd == 0.0) return false;
        double div = 1.0 / dd;
        Vec3 d = v_sub(ray.orig, pa);
        double b1 = v_dot(d, s0) * div;
        if (b1 < 0.0 || b1 > 1.0) return false;
        Vec3 s1 = v_cross(d, e0);
        double b2 = v_dot(ray.dir, s1) * div;
        if (b2 < 0.0 || b1 + b2 > 1.0) return false;
        double t = v_dot(e1, s1) * div;
        if (t < t_min || t > t_max) return false;
        double b0 = 1.0 - b1 - b2;
        Vec3 n = v_unit(v_add(v_add(v_muls(na, b0), v_muls(nb, b1)), v_muls(nc, b2)));
        out = HitRecord::make(t, n, ray, material, 0.0, 0.0, id);
        return true;
    }
    bool bounding_box(AABB& out) const override {
        out = box;
        return true;
    }
};

// Mesh -- raytracer/src/objects/mesh.rs:144-208 : per-mesh inner BVHNode
struct Mesh : Hitable {
    std::unique_ptr<MeshData> md;
    const Hitable* bvh = nullptr;
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& out, Ctx& cx) const override {
        return bvh->hit(r, t_min, t_max, out, cx);
    }
    bool bounding_box(AABB& out) const override { return bvh->bounding_box(out); }
};

// Transform -- raytracer/src/objects/transform.rs:9-169
struct Transform : Hitable {
    const Hitable* obj;
    Mat4 trans, inverse_trans;
    bool has_box = false;
    AABB box;
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& out, Ctx& cx) const override {  // :152-165
        cx.cnt.n_xform++;
        Ray trans_r{transform_point(r.orig, inverse_trans), transform_dir(r.dir, inverse_trans), r.time};
        HitRecord rec;
        if (!obj->hit(trans_r, t_min, t_max, rec, cx)) return false;
        Vec3 outward_normal = transform_dir(rec.normal, trans);  // M, not inverse-transpose (Q7)
        rec.p = transform_point(rec.p, trans);
        rec.set_face_normal(trans_r, outward_normal);  // object-space ray vs world normal (Q8)
        out = rec;
        return true;
    }
    bool bounding_box(AABB& out) const override {
        if (!has_box) return false;
        out = box;
        return true;
    }
};

// ----------------------------------------------------------------------------
// Camera -- raytracer/src/camera.rs:11-64
// ----------------------------------------------------------------------------
// ConstantMedium -- raytracer/src/objects/medium.rs:9-57 (dead code in the reference)
struct ConstantMedium : Hitable {
    const Hitable* boundary = nullptr;
    const Material* phase_function = nullptr;
    double neg_inv_density = 0.;  // -1 / d   (medium.rs:19)
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& out, Ctx& cx) const override {  // :25-53
        HitRecord rec1, rec2;
        if (!boundary->hit(r, -INF, INF, rec1, cx)) return false;
        if (!boundary->hit(r, rec1.t + 0.0001, INF, rec2, cx)) return false;
        rec1.t = std::fmax(rec1.t, t_min);
        rec2.t = std::fmin(rec2.t, t_max);
        if (rec1.t >= rec2.t) return false;
        rec1.t = std::fmax(rec1.t, 0.);
        const double ray_length = v_len(r.dir);
        const double distance_inside_boundary = (rec2.t - rec1.t) * ray_length;
        const double hit_distance = neg_inv_density * det_ln(cx.rng.gen_f64());  // the draw happens only here (:37-38)
        if (hit_distance > distance_inside_boundary) return false;
        const double t = rec1.t + hit_distance / ray_length;
        out = HitRecord::make(t, Vec3(1., 0., 0.), r, phase_function, 0.0, 0.0, id);  // arbitrary normal (1,0,0), uv (0,0)
        return true;
    }
    bool bounding_box(AABB& out) const override { return boundary->bounding_box(out); }  // :54-56
};

struct Camera {
    Vec3 origin, lower_left_corner, horizontal, vertical, u, v, w;
    double lens_radius = 0;
    double time0 = 0., time1 = 0.;  // D9: shutter open / close; time1 > time0 makes get_ray draw a time
    void init(Vec3 look_from, Vec3 look_at, Vec3 vup, double vfov, double aspect_ratio, double aperture, double focus_dist) {
        double theta = vfov * PI / 180.;  // degrees_to_radians, vec3.rs:10-12
        double h = std::tan(theta / 2.);
        double viewport_height = 2.0 * h;
        double viewport_width = aspect_ratio * viewport_height;
        w = v_unit(v_sub(look_from, look_at));
        u = v_unit(v_cross(vup, w));
        v = v_cross(w, u);
        origin = look_from;
        horizontal = v_muls(u, focus_dist * viewport_width);
        vertical = v_muls(v, focus_dist * viewport_height);
        lower_left_corner = v_sub(v_sub(v_sub(origin, v_divs(horizontal, 2.)), v_divs(vertical, 2.)), v_muls(w, focus_dist));
        lens_radius = aperture / 2.;
    }
    // camera.rs:57-64 ; the lens sample is drawn even when lens_radius == 0 (Q4)
    Ray get_ray(double s, double t, Rng& rng) const {
        Vec3 rd = v_muls(random_in_unit_disk(rng), lens_radius);
        Vec3 offset = v_add(v_muls(u, rd.x), v_muls(v, rd.y));
        Vec3 o = v_add(origin, offset);
        Vec3 d = v_sub(v_sub(v_add(v_add(lower_left_corner, v_muls(horizontal, s)), v_muls(vertical, t)), origin), offset);
        // D9 (book 2: ray(origin + offset, .., random_double(time0, time1))): one more draw, AFTER the lens sample, and only when the
        // shutter is open for a while -- scenes without motion keep their streams
        const double time = (time1 > time0) ? rng.gen_range(time0, time1) : time0;
        return Ray{o, d, time};
    }
};

// ----------------------------------------------------------------------------
// World + Integrator::sample_ray
//   world.rs:27-29 ; integrator/photon_mapper.rs:327-365 with divergence D2.
// ----------------------------------------------------------------------------
struct Scene {
    std::vector<std::unique_ptr<Texture>> textures;
    std::vector<std::unique_ptr<Material>> materials;
    std::vector<std::unique_ptr<Hitable>> objects;
    const Hitable* root = nullptr;
    std::vector<const Hitable*> lights;  // World::new's `lights` (world.rs:18): Sphere / XZ Rect hitables
    std::vector<Vec3> light_flux;        // XZRectLight / SphereDiffuseLight `flux` (light.rs:69-72,129-132)
    std::vector<double> light_scale;     // ... and `scale` (photon power = flux * scale)
    Camera cam;
    bool cam_set = false;
    std::string err;
};

static Vec3 sample_ray(const Scene& sc, Ray ray, int max_depth, double t_min, Ctx& cx) {
    Vec3 throughput(1, 1, 1);
    Vec3 radiance(0, 0, 0);
    Ray curr = ray;
    int depth = max_depth;
    HitRecord rec;
    for (;;) {
        cx.cnt.n_segments++;
        if (!sc.root->hit(curr, t_min, INF, rec, cx)) break;  // miss => black background
        if (depth <= 0) break;                                // Q12: test after the hit
        depth -= 1;
        radiance = v_add(radiance, v_elemul(throughput, rec.mat->emitted(rec)));  // Le, no face test
        ScatterResult sr = rec.mat->scatter(curr, rec, cx);
        if (sr.has_ray && sr.has_att) {
            // D2: Diffuse continues exactly like Specular/Reflect/Refract
            throughput = v_elemul(throughput, sr.att);
            const double time = curr.time;  // D9: scattered = ray(rec.p, direction, r_in.time())
            curr = sr.ray;
            curr.time = time;
        } else {
            break;  // Absorb
        }
    }
    return radiance;
}

// ----------------------------------------------------------------------------
// Light importance sampling (SURVEY.md s8f "next" #2; north_star's "importance-sampled PDF").
// The reference has NO pdf code: its only sketch is the dead Light::sample_li shadow-ray loop
// (light.rs:107-124,170-183) and XZRectLight::random_point_on_area (light.rs:148-154).  This
// integrator mode follows book 3 ("The Rest of Your Life") MixturePDF semantics instead:
// on a Diffuse interaction the next direction is drawn from  0.5 * p_lights + 0.5 * p_cosine
// and the throughput is weighted by  attenuation * scattering_pdf / pdf_mix .
// Everything is trig-free (Marsaglia / rejection sampling only), so CPU and GPU agree bit for bit.
//   p_cosine(dir)   = max(0, n . unit(dir)) / pi     (the density of the reference's own
//                     Lambertian sample n + random_unit_vector(), material.rs:92-98)
//   p_lights(dir)   = mean over lights of pdf_value(light, p, dir)   (book 3 hittable_list)
//   rect  light: pdf_value = t^2 |v|^2 / (|v.y|/|v| * area) if the ray (p, v) hits it, else 0;
//                random(p)  = (x0 + (x1-x0) u1, y, z0 + (z1-z0) u2) - p      (light.rs:148-154)
//   sphere light: pdf_value = 1 / (2 pi (1 - cos_max)), cos_max = sqrt(1 - r^2/|c-p|^2), if hit;
//                random(p)  = uniform direction in the cone, built from a rejection-sampled disk point
// RNG order per Diffuse interaction: scatter()'s own draws, coin, light index, light sample.
// PARITY UNPINNED (no reference counterpart); validated by equality in expectation with the
// brute-force tracer (tests/test_mixture.py).
// ----------------------------------------------------------------------------
static double light_pdf_value(const Hitable* l, Vec3 o, Vec3 v, Ctx& cx) {
    HitRecord rec;
    Ctx scratch;  // the pdf's own hit test is not part of the traversal counters
    (void)cx;
    if (!l->hit(Ray{o, v}, 0.001, INF, rec, scratch)) return 0.;
    if (const Rect* r = dynamic_cast<const Rect*>(l)) {
        double area = (r->a1 - r->a0) * (r->b1 - r->b0);
        double distance_squared = rec.t * rec.t * v_sqlen(v);
        double cosine = std::fabs(v.y / v_len(v));
        return distance_squared / (cosine * area);
    }
    const Sphere* s = static_cast<const Sphere*>(l);
    double cos_theta_max = std::sqrt(1. - s->radius * s->radius / v_sqlen(v_sub(s->center, o)));
    double solid_angle = 2. * PI * (1. - cos_theta_max);
    return 1. / solid_angle;
}
static Vec3 light_random(const Hitable* l, Vec3 o, Rng& rng) {
    if (const Rect* r = dynamic_cast<const Rect*>(l)) {
        double u = rng.gen_range(0., 1.), v = rng.gen_range(0., 1.);
        Vec3 p(r->a0 + (r->a1 - r->a0) * u, r->k, r->b0 + (r->b1 - r->b0) * v);
        return v_sub(p, o);
    }
    const Sphere* s = static_cast<const Sphere*>(l);
    Vec3 direction = v_sub(s->center, o);
    double distance_squared = v_sqlen(direction);
    // orthonormal basis around w = unit(direction) (book 3 onb::build_from_w)
    Vec3 w = v_unit(direction);
    Vec3 a = (std::fabs(w.x) > 0.9) ? Vec3(0, 1, 0) : Vec3(1, 0, 0);
    Vec3 vv = v_unit(v_cross(w, a));
    Vec3 uu = v_cross(w, vv);
    // uniform direction in the cone of half-angle acos(cos_max): z uniform in [cos_max, 1], azimuth from a disk sample
    Vec3 dsk = random_in_unit_disk(rng);
    double s2 = dsk.x * dsk.x + dsk.y * dsk.y;
    double cos_theta_max = std::sqrt(1. - s->radius * s->radius / distance_squared);
    double z = 1. + s2 * (cos_theta_max - 1.);
    double rr = std::sqrt(std::fmax(0., 1. - z * z));
    double inv_s = (s2 > 0.) ? 1. / std::sqrt(s2) : 0.;
    double x = dsk.x * inv_s * rr, y = dsk.y * inv_s * rr;
    return v_add(v_add(v_muls(uu, x), v_muls(vv, y)), v_muls(w, z));
}

static Vec3 sample_ray_mixture(const Scene& sc, Ray ray, int max_depth, double t_min, Ctx& cx) {
    Vec3 throughput(1, 1, 1);
    Vec3 radiance(0, 0, 0);
    Ray curr = ray;
    int depth = max_depth;
    HitRecord rec;
    const size_t n_lights = sc.lights.size();
    for (;;) {
        cx.cnt.n_segments++;
        if (!sc.root->hit(curr, t_min, INF, rec, cx)) break;
        if (depth <= 0) break;
        depth -= 1;
        radiance = v_add(radiance, v_elemul(throughput, rec.mat->emitted(rec)));
        ScatterResult sr = rec.mat->scatter(curr, rec, cx);
        if (!(sr.has_ray && sr.has_att)) break;
        if (sr.kind == Diffuse) {
            Vec3 dir = sr.ray.dir;  // the cosine-distributed sample the reference's scatter drew
            if (cx.rng.gen_f64() < 0.5) {
                size_t li = (size_t)(cx.rng.gen_f64() * (double)n_lights);
                if (li >= n_lights) li = n_lights - 1;
                dir = light_random(sc.lights[li], rec.p, cx.rng);
            }
            double cosine = v_dot(rec.normal, v_unit(dir));
            double scattering_pdf = (cosine < 0.) ? 0. : cosine / PI;
            double lp = 0.;
            for (size_t i = 0; i < n_lights; i++) lp = lp + light_pdf_value(sc.lights[i], rec.p, dir, cx);
            double pdf_val = 0.5 * (lp / (double)n_lights) + 0.5 * scattering_pdf;
            double wgt = scattering_pdf / pdf_val;
            if (!(wgt > 0.)) break;  // direction below the surface (or a NaN): the path carries nothing further
            throughput = v_muls(v_elemul(throughput, sr.att), wgt);
            curr = Ray{rec.p, dir, curr.time};
        } else {
            throughput = v_elemul(throughput, sr.att);
            const double time = curr.time;
            curr = sr.ray;
            curr.time = time;
        }
    }
    return radiance;
}

// ----------------------------------------------------------------------------
// SPPM -- the reference's actual integrator (integrator/photon_mapper.rs:17-324): per iteration, trace
// photons from the lights into a global and a caustic photon map, shoot one eye ray per pixel to its first
// diffuse hit and update that pixel's progressive statistics; the final render adds the per-pixel radiance
// estimate on the first Diffuse hit and stops (photon_mapper.rs:345-352).  SURVEY.md s8f "next" #3.
// Restated with seeded streams.  The photon maps are queried by BRUTE FORCE here (the reference uses the
// un-vendored kd-tree 0.4.1 crate; only the query RESULTS matter):
//   nearests(p, k)        -> the k photons of smallest squared distance      (photon_mapper.rs:85)
//   within_radius(p, r)   -> photons with  d^2 < r*r                         (photon_mapper.rs:105)
// DIVERGENCE D6: the flux of a query is accumulated in fixed point (terms rounded to 2^-S, S from the
// brightest light) so that the sum does not depend on the order in which a spatial index returns photons --
// the reference's order is that of the kd-tree crate and cannot be reproduced; the difference is ~1e-12
// relative.  D7: photon and eye paths are capped at `max_bounces` (the reference loops until absorbed).
// PARITY UNPINNED (no reference test touches this code).
// ----------------------------------------------------------------------------
struct Photon {  // light.rs:19-25
    Vec3 position, power, direction, norm;
};
struct SPPM {  // photon_mapper.rs:33-40
    Vec3 flux;
    double radius2 = 0.;
    uint64_t photons = 0;
};
struct SPPMPixel {  // photon_mapper.rs:66-70
    SPPM global, caustic;
};
struct SppmConfig {
    int iterations = 50;          // max_iter_cnt, photon_mapper.rs:148
    int photons_per_iter = 500000;  // photon_mapper.rs:149
    double alpha = 0.7;           // ALPHA :17
    int k_global = 100;           // GLOBAL_INIT_PHOTONS :18
    int k_caustic = 50;           // CAUSTIC_INIT_PHOTONS :19
    int max_bounces = 4096;       // D7
};
static const uint64_t SALT_PHOTON = 0x50484F544F4E5F31ULL, SALT_EYE = 0x5350504D4559455FULL;

static int sppm_fixed_shift(const Scene& sc) {
    double maxp = 0.;
    for (size_t i = 0; i < sc.lights.size(); i++) {
        Vec3 p = v_muls(sc.light_flux[i], sc.light_scale[i]);
        maxp = std::fmax(maxp, std::fmax(std::fabs(p.x), std::fmax(std::fabs(p.y), std::fabs(p.z))));
    }
    if (!(maxp > 0.) || !std::isfinite(maxp)) return 40;
    int e;
    std::frexp(maxp, &e);  // maxp = m * 2^e, m in [0.5, 1)
    int S = 40 - e;
    return S < 0 ? 0 : (S > 60 ? 60 : S);
}
struct FixedFlux {
    int64_t x = 0, y = 0, z = 0;
    void add(Vec3 t, int S) {
        x += std::llrint(std::ldexp(t.x, S));
        y += std::llrint(std::ldexp(t.y, S));
        z += std::llrint(std::ldexp(t.z, S));
    }
    Vec3 get(int S) const { return Vec3(std::ldexp((double)x, -S), std::ldexp((double)y, -S), std::ldexp((double)z, -S)); }
};

// AllLights::emit (light.rs:220-225): WeightedIndex over |power| ; then Light::emit
static void lights_emit(const Scene& sc, const std::vector<double>& cum, double total, Rng& rng, Ray& ray, Vec3& power) {
    double chosen = rng.gen_range(0., total);  // WeightedIndex: Uniform::new(0, total_weight)
    size_t idx = 0;
    while (idx + 1 < sc.lights.size() && cum[idx] <= chosen) idx++;  // partition_point(|w| w <= chosen) over the first n-1 sums
    const Hitable* l = sc.lights[idx];
    Vec3 flux = sc.light_flux[idx];
    double scale = sc.light_scale[idx];
    if (const Rect* r = dynamic_cast<const Rect*>(l)) {  // XZRectLight::emit, light.rs:158-166
        double u = rng.gen_range(0., 1.), v = rng.gen_range(0., 1.);
        Vec3 orig(r->a0 + (r->a1 - r->a0) * u, r->k, r->b0 + (r->b1 - r->b0) * v);
        Vec3 w = random_in_hemisphere(rng, Vec3(0., -1., 0.));
        ray = Ray{orig, w};
        power = v_muls(v_muls(flux, scale), std::fmax(v_dot(Vec3(0., -1., 0.), w), 0.));
    } else {  // SphereDiffuseLight::emit, light.rs:98-103
        const Sphere* s = static_cast<const Sphere*>(l);
        Vec3 norm = random_in_unit_sphere(rng);
        Vec3 point = v_add(s->center, v_muls(norm, s->radius + 0.0001));
        Vec3 dir = random_in_hemisphere(rng, norm);
        ray = Ray{point, dir};
        power = v_muls(flux, scale);
    }
}

// SPPMIntegrator::generate_photon_map (photon_mapper.rs:234-276) for photon paths [g0, g1) of the global numbering
static void trace_photons(const Scene& sc, const SppmConfig& cfg, uint64_t seed, uint64_t g0, uint64_t g1, std::vector<Photon>& all,
                          std::vector<Photon>& caustic) {
    std::vector<double> cum;
    double total = 0.;
    {
        std::vector<double> lp;
        double tot = 0.;
        for (size_t i = 0; i < sc.lights.size(); i++) {  // AllLights::new, light.rs:202-217
            lp.push_back(v_len(v_muls(sc.light_flux[i], sc.light_scale[i])));
            tot = tot + lp.back();
        }
        for (size_t i = 0; i < lp.size(); i++) {
            total = total + lp[i] / tot;
            cum.push_back(total);
        }
    }
    Ctx cx;
    for (uint64_t g = g0; g < g1; g++) {
        cx.rng = Rng(seed ^ SALT_PHOTON, g, 0);
        Ray ray;
        Vec3 power;
        lights_emit(sc, cum, total, cx.rng, ray, power);
        bool has_specular = false, has_diffuse = false;
        HitRecord rec;
        for (int bounce = 0; bounce < cfg.max_bounces; bounce++) {
            if (!sc.root->hit(ray, 0.0001, INF, rec, cx)) break;
            // Material::scatter_photon, material.rs:27-45 (Russian roulette on max(f))
            ScatterResult sr = rec.mat->scatter(ray, rec, cx);
            Interaction kind = Absorb;
            bool cont = false;
            Vec3 new_power;
            if (sr.has_att) {
                double hmax = v_max(sr.att);
                if (cx.rng.gen_f64() > hmax) {
                    kind = Absorb;
                } else {
                    kind = sr.kind;
                    new_power = v_elemul(power, v_divs(sr.att, v_max(sr.att)));
                    cont = sr.has_ray;
                }
            }
            if (kind == Diffuse) {
                Photon ph{rec.p, power, ray.dir, rec.normal};
                all.push_back(ph);
                if (!has_diffuse && has_specular) caustic.push_back(ph);
                has_diffuse = true;
            } else if (kind == Absorb) {
                break;
            } else {
                has_specular = true;
            }
            if (cont) {
                ray = sr.ray;
                power = new_power;
            }
        }
    }
}

struct GatherPoint {  // what update_sppm needs from the eye path's first Diffuse hit
    bool valid = false;
    Vec3 p, bsdf;  // rec.p and rec.mat.bsdf(., rec) (constant over photons for every material of the reference)
};
static double disk_factor(Vec3 p, const Photon& ph) {  // photon_mapper.rs:77-79
    return std::fabs(v_dot(ph.norm, v_unit(v_sub(ph.position, p))));
}
// PhotonMap::estimate_flux_by_count, photon_mapper.rs:82-100
static void estimate_by_count(const std::vector<Photon>& pm, const GatherPoint& gp, size_t k, int S, Vec3& flux, double& radius2) {
    std::vector<double> d2(pm.size());
    for (size_t i = 0; i < pm.size(); i++) d2[i] = v_sqlen(v_sub(pm[i].position, gp.p));
    radius2 = 0.;
    FixedFlux acc;
    if (!pm.empty()) {
        size_t kk = std::min(k, pm.size());
        std::vector<double> tmp = d2;
        std::nth_element(tmp.begin(), tmp.begin() + (kk - 1), tmp.end());
        double r2k = tmp[kk - 1];  // squared distance of the k-th nearest
        for (size_t i = 0; i < pm.size(); i++)
            if (d2[i] <= r2k) {  // exact ties at the k-th distance are all taken (measure zero)
                radius2 = std::fmax(radius2, d2[i]);
                acc.add(v_muls(v_elemul(gp.bsdf, pm[i].power), 1. - disk_factor(gp.p, pm[i])), S);
            }
    }
    flux = acc.get(S);
}
// PhotonMap::estimate_flux_within_radius, photon_mapper.rs:101-114
static void estimate_within_radius(const std::vector<Photon>& pm, const GatherPoint& gp, double radius, int S, Vec3& flux, uint64_t& count) {
    FixedFlux acc;
    count = 0;
    for (size_t i = 0; i < pm.size(); i++) {
        double d2 = v_sqlen(v_sub(pm[i].position, gp.p));
        if (d2 < radius * radius) {
            count++;
            acc.add(v_muls(v_elemul(gp.bsdf, pm[i].power), 1. - disk_factor(gp.p, pm[i])), S);
        }
    }
    flux = acc.get(S);
}
// SPPM::update, photon_mapper.rs:49-63
static void sppm_update(SPPM& s, const GatherPoint& gp, const std::vector<Photon>& pm, size_t photon_init, double alpha, int S) {
    if (s.photons == 0) {
        s.photons = photon_init;
        estimate_by_count(pm, gp, photon_init, S, s.flux, s.radius2);
    } else {
        Vec3 flux;
        uint64_t found;
        estimate_within_radius(pm, gp, std::sqrt(s.radius2), S, flux, found);
        uint64_t prev = s.photons;
        s.photons += (uint64_t)(alpha * (double)found);
        double frac = (double)s.photons / (double)(prev + found);
        s.radius2 *= frac;
        s.flux = v_muls(v_add(s.flux, flux), frac);
    }
}
// SPPMIntegrator::update_sppm's eye path, photon_mapper.rs:277-296
static GatherPoint eye_gather_point(const Scene& sc, Ray ray, const SppmConfig& cfg, Ctx& cx) {
    GatherPoint gp;
    Ray curr = ray;
    HitRecord rec;
    for (int bounce = 0; bounce < cfg.max_bounces; bounce++) {
        if (!sc.root->hit(curr, 0.001, INF, rec, cx)) break;
        ScatterResult sr = rec.mat->scatter(curr, rec, cx);
        if (sr.kind == Diffuse) {
            gp.valid = true;
            gp.p = rec.p;
            gp.bsdf = rec.mat->bsdf(curr.dir, rec);
            break;
        }
        if (sr.has_ray && sr.has_att) curr = sr.ray;
        else break;
    }
    return gp;
}

// SPPMIntegrator::new (photon_mapper.rs:139-233): returns per-pixel stats, pixel-major [y*W + x]
static int sppm_prepass(const Scene& sc, const SppmConfig& cfg, int width, int height, uint64_t seed, int n_workers,
                        std::vector<SPPMPixel>& px, uint64_t* n_global_total, uint64_t* n_caustic_total) {
    if (!sc.root || !sc.cam_set || sc.lights.empty()) return -1;
    px.assign((size_t)width * height, SPPMPixel());
    const int S = sppm_fixed_shift(sc);
    if (n_workers < 1) n_workers = 1;
    std::atomic<int> err{0};
    uint64_t tg = 0, tc = 0;
    for (int it = 0; it < cfg.iterations; it++) {
        // photon tracing pass (parallel over photon paths; the maps are sets, their order is irrelevant by D6)
        std::vector<std::vector<Photon>> all_t(n_workers), cau_t(n_workers);
        {
            std::vector<std::thread> th;
            for (int w = 0; w < n_workers; w++)
                th.emplace_back([&, w]() {
                    uint64_t P = (uint64_t)cfg.photons_per_iter;
                    uint64_t a = P * w / n_workers, b = P * (w + 1) / n_workers;
                    try {
                        trace_photons(sc, cfg, seed, (uint64_t)it * P + a, (uint64_t)it * P + b, all_t[w], cau_t[w]);
                    } catch (const UnitZero&) {
                        err.store(-2);
                    }
                });
            for (auto& t : th) t.join();
        }
        std::vector<Photon> all, cau;
        for (int w = 0; w < n_workers; w++) {
            all.insert(all.end(), all_t[w].begin(), all_t[w].end());
            cau.insert(cau.end(), cau_t[w].begin(), cau_t[w].end());
        }
        tg += all.size();
        tc += cau.size();
        // eye pass: one jittered ray per pixel (photon_mapper.rs:186-201)
        std::atomic<int> next_row{0};
        std::vector<std::thread> th;
        for (int w = 0; w < n_workers; w++)
            th.emplace_back([&]() {
                Ctx cx;
                for (;;) {
                    int y = next_row.fetch_add(1);
                    if (y >= height) break;
                    for (int x = 0; x < width; x++) {
                        try {
                            uint64_t pix = (uint64_t)y * width + x;
                            cx.rng = Rng(seed ^ SALT_EYE, pix, (uint64_t)it);
                            double u = ((double)x + cx.rng.gen_f64()) / (double)(width - 1);
                            double v = ((double)y + cx.rng.gen_f64()) / (double)(height - 1);
                            Ray r = sc.cam.get_ray(u, 1.0 - v, cx.rng);
                            GatherPoint gp = eye_gather_point(sc, r, cfg, cx);
                            if (gp.valid) {
                                sppm_update(px[pix].caustic, gp, cau, (size_t)cfg.k_caustic, cfg.alpha, S);  // caustic first (:284)
                                sppm_update(px[pix].global, gp, all, (size_t)cfg.k_global, cfg.alpha, S);
                            }
                        } catch (const UnitZero&) {
                            err.store(-2);
                        }
                    }
                }
            });
        for (auto& t : th) t.join();
    }
    if (n_global_total) *n_global_total = tg;
    if (n_caustic_total) *n_caustic_total = tc;
    return err.load();
}
// adjust_flux, photon_mapper.rs:117-119 ; N = max_iter_cnt * photon_per_iter for both maps (:227-228)
static Vec3 sppm_estimate(const SPPM& s, double n_emitted) { return v_divs(s.flux, PI * s.radius2 * n_emitted); }

// SPPMIntegrator::sample_ray LITERALLY (photon_mapper.rs:327-365): Diffuse adds the pixel's estimates and stops
static Vec3 sample_ray_sppm(const Scene& sc, Ray ray, int max_depth, double t_min, Vec3 est_caustic, Vec3 est_global, Ctx& cx) {
    Vec3 throughput(1, 1, 1);
    Vec3 radiance(0, 0, 0);
    Ray curr = ray;
    int depth = max_depth;
    HitRecord rec;
    for (;;) {
        cx.cnt.n_segments++;
        if (!sc.root->hit(curr, t_min, INF, rec, cx)) break;
        if (depth <= 0) break;
        depth -= 1;
        radiance = v_add(radiance, v_elemul(throughput, rec.mat->emitted(rec)));
        ScatterResult sr = rec.mat->scatter(curr, rec, cx);
        if (sr.kind == Diffuse && sr.has_ray && sr.has_att) {
            radiance = v_add(radiance, v_elemul(throughput, est_caustic));
            radiance = v_add(radiance, v_elemul(throughput, est_global));
            break;
        } else if (sr.has_ray && sr.has_att) {
            throughput = v_elemul(throughput, sr.att);
            curr = sr.ray;
        } else {
            break;
        }
    }
    return radiance;
}

struct RenderArgs {
    int width, height, spp, max_depth;
    double t_min;
    uint64_t seed;
    int x0, y0, x1, y1;  // pixel window [x0,x1) x [y0,y1) rendered with the full-frame camera mapping
    int integrator = 0;  // 0: sample_ray (BSDF sampling only); 1: sample_ray_mixture; 2: sample_ray_sppm (needs `sppm`)
    const SPPMPixel* sppm = nullptr;  // per-pixel statistics of the pre-pass, [y*W + x]
    double sppm_emitted = 0.;         // iterations * photons_per_iter
};

// Camera::capture_image -- camera.rs:66-128.  n_jobs row bands executed FIFO by
// n_workers threads; per-pixel sample loop :96-102.  Output: linear radiance
// (sum / spp) f64 RGB, row-major, y down, window-local indexing.
static int render(const Scene& sc, const RenderArgs& a, int n_jobs, int n_workers, double* out_rgb, Counters* total) {
    if (!sc.root || !sc.cam_set) return -1;
    if (a.integrator == 1 && sc.lights.empty()) return -1;
    if (a.integrator == 2 && !a.sppm) return -1;
    const int wh = a.y1 - a.y0, ww = a.x1 - a.x0;
    if (n_jobs < 1) n_jobs = 1;
    if (n_workers < 1) n_workers = 1;
    std::atomic<int> next_job{0};
    std::atomic<int> err{0};
    std::mutex mu;
    Counters sum;
    auto worker = [&]() {
        Ctx cx;
        for (;;) {
            int i = next_job.fetch_add(1);
            if (i >= n_jobs) break;
            int row_begin = a.y0 + (int)((long)wh * i / n_jobs);      // camera.rs:85
            int row_end = a.y0 + (int)((long)wh * (i + 1) / n_jobs);  // camera.rs:86
            try {
                for (int x = a.x0; x < a.x1; x++) {  // x is the OUTER loop, camera.rs:91
                    for (int y = row_begin; y < row_end; y++) {
                        Vec3 pixel(0, 0, 0);
                        Vec3 est_c, est_g;
                        if (a.integrator == 2) {
                            const SPPMPixel& sp = a.sppm[(size_t)y * a.width + x];
                            est_c = sppm_estimate(sp.caustic, a.sppm_emitted);
                            est_g = sppm_estimate(sp.global, a.sppm_emitted);
                        }
                        for (int s = 0; s < a.spp; s++) {
                            cx.rng = Rng(a.seed, (uint64_t)y * (uint64_t)a.width + (uint64_t)x, (uint64_t)s);
                            double u = ((double)x + cx.rng.gen_f64()) / (double)(a.width - 1);
                            double v = ((double)y + cx.rng.gen_f64()) / (double)(a.height - 1);
                            Ray r = sc.cam.get_ray(u, 1.0 - v, cx.rng);
                            pixel = v_add(pixel, a.integrator == 2   ? sample_ray_sppm(sc, r, a.max_depth, a.t_min, est_c, est_g, cx)
                                                 : a.integrator == 1 ? sample_ray_mixture(sc, r, a.max_depth, a.t_min, cx)
                                                                     : sample_ray(sc, r, a.max_depth, a.t_min, cx));
                            cx.cnt.n_samples++;
                        }
                        pixel = v_divs(pixel, (double)a.spp);
                        double* o = out_rgb + ((size_t)(y - a.y0) * ww + (x - a.x0)) * 3;
                        o[0] = pixel.x; o[1] = pixel.y; o[2] = pixel.z;
                    }
                }
            } catch (const UnitZero&) {
                err.store(-2);
            }
        }
        std::lock_guard<std::mutex> g(mu);
        sum.add(cx.cnt);
    };
    std::vector<std::thread> th;
    for (int i = 0; i < n_workers; i++) th.emplace_back(worker);
    for (auto& t : th) t.join();
    if (total) *total = sum;
    return err.load();
}

// From<Vec3> for Rgb<u8> -- vec3.rs:223-231 : floor(clamp(sqrt(c),0,1)*255); NaN -> 0 (saturating cast)
static inline uint8_t tonemap_channel(double c) {
    double s = std::sqrt(c);
    if (s < 0.) s = 0.;        // f64::clamp: NaN stays NaN
    else if (s > 1.) s = 1.;
    double f = std::floor(s * 255.);
    if (!(f == f)) return 0;   // NaN as u8 == 0
    if (f <= 0.) return 0;
    if (f >= 255.) return 255;
    return (uint8_t)f;
}

// BVHNode::construct / ::new / box_compare -- bvh.rs:34-83 (axis from a seeded stream, D3)
static const Hitable* bvh_construct(Scene& sc, const Hitable* l, const Hitable* r) {
    AABB bl, br;
    if (!l->bounding_box(bl) || !r->bounding_box(br)) throw std::runtime_error("No bounding box in bvh_node constructor.");
    auto n = std::make_unique<BVHNode>();
    n->left = l;
    n->right = r;
    n->box = AABB::surrounding_box(bl, br);
    n->id = (int)sc.objects.size();
    const Hitable* p = n.get();
    sc.objects.push_back(std::move(n));
    return p;
}
static const Hitable* bvh_new(Scene& sc, std::vector<const Hitable*> objs, Rng& rng) {
    int axis = (int)rng.gen_below3();  // bvh.rs:61-62
    auto less = [axis](const Hitable* a, const Hitable* b) {  // box_compare: only "Less" is observable to a stable sort
        AABB ba, bb;
        if (!a->bounding_box(ba) || !b->bounding_box(bb)) throw std::runtime_error("No bounding box in bvh_node constructor.");
        return ba.minimum[axis] < bb.minimum[axis];
    };
    size_t n = objs.size();
    if (n == 0) throw std::runtime_error("BVHNode::new on an empty list");
    if (n == 1) return bvh_construct(sc, objs[0], objs[0]);  // Q14: same object in both children
    if (n == 2) {
        if (less(objs[0], objs[1])) return bvh_construct(sc, objs[0], objs[1]);  // .is_le() is true only for Less
        return bvh_construct(sc, objs[1], objs[0]);
    }
    std::stable_sort(objs.begin(), objs.end(), less);  // slice::sort_by is a stable merge sort
    size_t mid = n / 2;
    std::vector<const Hitable*> lo(objs.begin(), objs.begin() + mid), hi(objs.begin() + mid, objs.end());
    const Hitable* l = bvh_new(sc, lo, rng);
    const Hitable* r = bvh_new(sc, hi, rng);
    return bvh_construct(sc, l, r);
}

}  // namespace orc

// ============================================================================
// C API for the ctypes harness (tests / smoke / cpu_baseline only)
// ============================================================================
using namespace orc;

#define ORC_OK 0
#define ORC_ERR_ARG -1
#define ORC_ERR_UNIT_ZERO -2
#define ORC_ERR_BUILD -3

template <class T>
static int push_obj(Scene& sc, std::unique_ptr<T> o) {
    o->id = (int)sc.objects.size();
    sc.objects.push_back(std::move(o));
    return (int)sc.objects.size() - 1;
}

extern "C" {

void* orc_scene_new() { return new Scene(); }
void orc_scene_free(void* s) { delete (Scene*)s; }
const char* orc_last_error(void* s) { return ((Scene*)s)->err.c_str(); }

int orc_tex_constant(void* s, double r, double g, double b) {
    Scene& sc = *(Scene*)s;
    sc.textures.emplace_back(new ConstantTexture(Vec3(r, g, b)));
    return (int)sc.textures.size() - 1;
}
int orc_tex_checker(void* s, int t0, int t1) {
    Scene& sc = *(Scene*)s;
    if (t0 < 0 || t1 < 0 || t0 >= (int)sc.textures.size() || t1 >= (int)sc.textures.size()) return ORC_ERR_ARG;
    sc.textures.emplace_back(new CheckerTexture(sc.textures[t0].get(), sc.textures[t1].get()));
    return (int)sc.textures.size() - 1;
}
int orc_tex_image(void* s, int w, int h, const uint8_t* rgb) {
    Scene& sc = *(Scene*)s;
    if (w <= 0 || h <= 0 || !rgb) return ORC_ERR_ARG;
    sc.textures.emplace_back(new ImageTexture(w, h, rgb));
    return (int)sc.textures.size() - 1;
}
static const Texture* tex(Scene& sc, int t) { return (t >= 0 && t < (int)sc.textures.size()) ? sc.textures[t].get() : nullptr; }
static const Material* mat(Scene& sc, int m) { return (m >= 0 && m < (int)sc.materials.size()) ? sc.materials[m].get() : nullptr; }
static const Hitable* obj(Scene& sc, int o) { return (o >= 0 && o < (int)sc.objects.size()) ? sc.objects[o].get() : nullptr; }

int orc_mat_lambertian(void* s, int t) {
    Scene& sc = *(Scene*)s;
    if (!tex(sc, t)) return ORC_ERR_ARG;
    sc.materials.emplace_back(new Lambertian(tex(sc, t)));
    return (int)sc.materials.size() - 1;
}
int orc_mat_metal(void* s, int t, double fuzz) {
    Scene& sc = *(Scene*)s;
    if (!tex(sc, t)) return ORC_ERR_ARG;
    sc.materials.emplace_back(new Metal(tex(sc, t), fuzz));
    return (int)sc.materials.size() - 1;
}
int orc_mat_dielectric(void* s, double ir, int t) {
    Scene& sc = *(Scene*)s;
    if (!tex(sc, t)) return ORC_ERR_ARG;
    sc.materials.emplace_back(new Dielectric(ir, tex(sc, t)));
    return (int)sc.materials.size() - 1;
}
int orc_mat_diffuse_light(void* s, int t) {
    Scene& sc = *(Scene*)s;
    if (!tex(sc, t)) return ORC_ERR_ARG;
    sc.materials.emplace_back(new DiffuseLight(tex(sc, t)));
    return (int)sc.materials.size() - 1;
}

int orc_mat_isotropic(void* s, int t) {
    Scene& sc = *(Scene*)s;
    if (!tex(sc, t)) return ORC_ERR_ARG;
    sc.materials.emplace_back(new Isotropic(tex(sc, t)));
    return (int)sc.materials.size() - 1;
}
double orc_det_ln(double x) { return det_ln(x); }
double orc_det_sin(double x) { return det_sin(x); }
// D9
int orc_tex_noise(void* s, double scale, uint64_t seed) {
    Scene& sc = *(Scene*)s;
    try {
        sc.textures.emplace_back(new NoiseTexture(scale, seed));
    } catch (const UnitZero&) {
        return ORC_ERR_UNIT_ZERO;
    }
    return (int)sc.textures.size() - 1;
}
int orc_noise_value(void* s, int t, const double* p3, double* out3) {  // {noise(p), turb(p), the marble value}
    Scene& sc = *(Scene*)s;
    const NoiseTexture* n = dynamic_cast<const NoiseTexture*>(tex(sc, t));
    if (!n) return ORC_ERR_ARG;
    const Vec3 p(p3[0], p3[1], p3[2]);
    out3[0] = n->noise(p);
    out3[1] = n->turb(p);
    out3[2] = n->value(p);
    return ORC_OK;
}
int orc_moving_sphere(void* s, const double* c0, const double* c1, double time0, double time1, double r, int m) {
    Scene& sc = *(Scene*)s;
    if (!mat(sc, m) || !(time1 > time0)) return ORC_ERR_ARG;
    auto o = std::make_unique<MovingSphere>();
    o->center0 = Vec3(c0[0], c0[1], c0[2]);
    o->center1 = Vec3(c1[0], c1[1], c1[2]);
    o->time0 = time0;
    o->time1 = time1;
    o->radius = r;
    o->material = mat(sc, m);
    return push_obj(sc, std::move(o));
}
int orc_set_shutter(void* s, double time0, double time1) {
    Scene& sc = *(Scene*)s;
    if (!(time1 >= time0)) return ORC_ERR_ARG;
    sc.cam.time0 = time0;
    sc.cam.time1 = time1;
    return ORC_OK;
}

int orc_sphere(void* s, double cx, double cy, double cz, double r, int m) {
    Scene& sc = *(Scene*)s;
    if (!mat(sc, m)) return ORC_ERR_ARG;
    auto o = std::make_unique<Sphere>();
    o->center = Vec3(cx, cy, cz);
    o->radius = r;
    o->material = mat(sc, m);
    return push_obj(sc, std::move(o));
}
// axis: 2 = XYRectangle{xy0,xy1,z}, 1 = XZRectangle{xz0,xz1,y}, 0 = YZRectangle{yz0,yz1,x}
int orc_rect(void* s, int axis, double a0, double b0, double a1, double b1, double k, int m) {
    Scene& sc = *(Scene*)s;
    if (!mat(sc, m) || axis < 0 || axis > 2) return ORC_ERR_ARG;
    auto o = std::make_unique<Rect>();
    o->axis = axis; o->a0 = a0; o->b0 = b0; o->a1 = a1; o->b1 = b1; o->k = k;
    o->material = mat(sc, m);
    return push_obj(sc, std::move(o));
}
static std::unique_ptr<Rect> mk_rect(int axis, double a0, double b0, double a1, double b1, double k, const Material* m) {
    auto o = std::make_unique<Rect>();
    o->axis = axis; o->a0 = a0; o->b0 = b0; o->a1 = a1; o->b1 = b1; o->k = k; o->material = m;
    return o;
}
int orc_cube(void* s, const double* mn, const double* mx, int m) {  // cube.rs:16-61 side order
    Scene& sc = *(Scene*)s;
    if (!mat(sc, m)) return ORC_ERR_ARG;
    const Material* mm = mat(sc, m);
    auto c = std::make_unique<Cube>();
    c->box_min = Vec3(mn[0], mn[1], mn[2]);
    c->box_max = Vec3(mx[0], mx[1], mx[2]);
    int ids[6];
    ids[0] = push_obj(sc, mk_rect(2, mn[0], mn[1], mx[0], mx[1], mn[2], mm));
    ids[1] = push_obj(sc, mk_rect(2, mn[0], mn[1], mx[0], mx[1], mx[2], mm));
    ids[2] = push_obj(sc, mk_rect(1, mn[0], mn[2], mx[0], mx[2], mn[1], mm));
    ids[3] = push_obj(sc, mk_rect(1, mn[0], mn[2], mx[0], mx[2], mx[1], mm));
    ids[4] = push_obj(sc, mk_rect(0, mn[1], mn[2], mx[1], mx[2], mn[0], mm));
    ids[5] = push_obj(sc, mk_rect(0, mn[1], mn[2], mx[1], mx[2], mx[0], mm));
    for (int i = 0; i < 6; i++) c->sides.items.push_back(sc.objects[ids[i]].get());
    return push_obj(sc, std::move(c));
}
// ConstantMedium::new(d, boundary, phase_function) -- medium.rs:16-22
int orc_constant_medium(void* s, double density, int boundary, int phase) {
    Scene& sc = *(Scene*)s;
    if (!obj(sc, boundary) || !mat(sc, phase)) return ORC_ERR_ARG;
    auto o = std::make_unique<ConstantMedium>();
    o->boundary = obj(sc, boundary);
    o->phase_function = mat(sc, phase);
    o->neg_inv_density = -1. / density;
    return push_obj(sc, std::move(o));
}
int orc_list(void* s, int n, const int* ids) {
    Scene& sc = *(Scene*)s;
    auto l = std::make_unique<HitableList>();
    for (int i = 0; i < n; i++) {
        if (!obj(sc, ids[i])) return ORC_ERR_ARG;
        l->items.push_back(obj(sc, ids[i]));
    }
    return push_obj(sc, std::move(l));
}
int orc_bvh_construct(void* s, int l, int r) {
    Scene& sc = *(Scene*)s;
    if (!obj(sc, l) || !obj(sc, r)) return ORC_ERR_ARG;
    try {
        return bvh_construct(sc, obj(sc, l), obj(sc, r))->id;
    } catch (const std::exception& e) {
        sc.err = e.what();
        return ORC_ERR_BUILD;
    }
}
int orc_bvh_new(void* s, int n, const int* ids, uint64_t seed) {
    Scene& sc = *(Scene*)s;
    std::vector<const Hitable*> v;
    for (int i = 0; i < n; i++) {
        if (!obj(sc, ids[i])) return ORC_ERR_ARG;
        v.push_back(obj(sc, ids[i]));
    }
    try {
        Rng rng(seed, 0xB7E151628AED2A6AULL, 0);  // BVH-build stream (spec rtamd-rng-3, "bvh" key)
        return bvh_new(sc, v, rng)->id;
    } catch (const std::exception& e) {
        sc.err = e.what();
        return ORC_ERR_BUILD;
    }
}
// Mesh::load_obj's result given already-parsed arrays (positions/normals: n_vert*3 f64; idx: n_tri*3)
int orc_mesh(void* s, int n_vert, const double* pos, const double* nrm, int n_tri, const uint32_t* idx, int m, uint64_t seed) {
    Scene& sc = *(Scene*)s;
    if (!mat(sc, m) || n_vert <= 0 || n_tri <= 0 || !pos || !nrm || !idx) return ORC_ERR_ARG;
    auto me = std::make_unique<Mesh>();
    me->md = std::make_unique<MeshData>();
    for (int i = 0; i < n_vert; i++) {
        me->md->positions.emplace_back(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
        me->md->normals.emplace_back(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]);
    }
    std::vector<const Hitable*> tris;
    for (int i = 0; i < n_tri; i++) {
        for (int k = 0; k < 3; k++)
            if (idx[3 * i + k] >= (uint32_t)n_vert) return ORC_ERR_ARG;
        auto t = std::make_unique<Triangle>();
        t->a = idx[3 * i]; t->b = idx[3 * i + 1]; t->c = idx[3 * i + 2];
        t->md = me->md.get();
        t->material = mat(sc, m);
        t->init_box();
        int id = push_obj(sc, std::move(t));
        tris.push_back(sc.objects[id].get());
    }
    try {
        Rng rng(seed, 0xB7E151628AED2A6AULL, 0);
        me->bvh = bvh_new(sc, tris, rng);
    } catch (const std::exception& e) {
        sc.err = e.what();
        return ORC_ERR_BUILD;
    }
    return push_obj(sc, std::move(me));
}
// Transform::new -- transform.rs:17-148 : M = T * S * Rx * Ry * Rz
int orc_transform(void* s, const double* rot_deg, const double* scale, const double* translate, int o) {
    Scene& sc = *(Scene*)s;
    if (!obj(sc, o)) return ORC_ERR_ARG;
    double rx = rot_deg[0] * PI / 180., ry = rot_deg[1] * PI / 180., rz = rot_deg[2] * PI / 180.;
    Mat4 T = {{{1., 0., 0., translate[0]}, {0., 1., 0., translate[1]}, {0., 0., 1., translate[2]}, {0., 0., 0., 1.}}};
    Mat4 S = {{{scale[0], 0., 0., 0.}, {0., scale[1], 0., 0.}, {0., 0., scale[2], 0.}, {0., 0., 0., 1.}}};
    Mat4 RX = {{{1., 0., 0., 0.}, {0., std::cos(rx), -std::sin(rx), 0.}, {0., std::sin(rx), std::cos(rx), 0.}, {0., 0., 0., 1.}}};
    Mat4 RY = {{{std::cos(ry), 0., std::sin(ry), 0.}, {0., 1., 0., 0.}, {-std::sin(ry), 0., std::cos(ry), 0.}, {0., 0., 0., 1.}}};
    Mat4 RZ = {{{std::cos(rz), -std::sin(rz), 0., 0.}, {std::sin(rz), std::cos(rz), 0., 0.}, {0., 0., 1., 0.}, {0., 0., 0., 1.}}};
    Mat4 M = mat_mul(mat_mul(mat_mul(mat_mul(T, S), RX), RY), RZ);
    auto t = std::make_unique<Transform>();
    t->obj = obj(sc, o);
    t->trans = M;
    AABB bb;
    if (t->obj->bounding_box(bb)) {  // :110-136 : 8 transformed corners
        double mn[3] = {INF, INF, INF}, mx[3] = {-INF, -INF, -INF};
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++)
                for (int k = 0; k < 2; k++) {
                    double fi = i, fj = j, fk = k;
                    Vec3 tester(fi * bb.maximum.x + (1. - fi) * bb.minimum.x, fj * bb.maximum.y + (1. - fj) * bb.minimum.y, fk * bb.maximum.z + (1. - fk) * bb.minimum.z);
                    tester = transform_point(tester, M);
                    for (int c = 0; c < 3; c++) {
                        mn[c] = std::fmin(mn[c], tester[c]);
                        mx[c] = std::fmax(mx[c], tester[c]);
                    }
                }
        t->box = AABB{Vec3(mn[0], mn[1], mn[2]), Vec3(mx[0], mx[1], mx[2])};
        t->has_box = true;
    }
    if (!mat_inverse(M, t->inverse_trans)) {
        sc.err = "Invalid transform matrix";  // transform.rs:146
        return ORC_ERR_BUILD;
    }
    return push_obj(sc, std::move(t));
}
int orc_set_camera(void* s, const double* from, const double* at, const double* vup, double vfov, double aspect, double aperture, double focus) {
    Scene& sc = *(Scene*)s;
    try {
        sc.cam.init(Vec3(from[0], from[1], from[2]), Vec3(at[0], at[1], at[2]), Vec3(vup[0], vup[1], vup[2]), vfov, aspect, aperture, focus);
    } catch (const UnitZero&) {
        return ORC_ERR_UNIT_ZERO;
    }
    sc.cam_set = true;
    return ORC_OK;
}
// camera basis readback (origin, llc, horizontal, vertical, u, v, w, lens_radius) = 22 doubles
int orc_get_camera(void* s, double* out) {
    Scene& sc = *(Scene*)s;
    if (!sc.cam_set) return ORC_ERR_ARG;
    const Vec3* vs[7] = {&sc.cam.origin, &sc.cam.lower_left_corner, &sc.cam.horizontal, &sc.cam.vertical, &sc.cam.u, &sc.cam.v, &sc.cam.w};
    for (int i = 0; i < 7; i++) { out[3 * i] = vs[i]->x; out[3 * i + 1] = vs[i]->y; out[3 * i + 2] = vs[i]->z; }
    out[21] = sc.cam.lens_radius;
    return ORC_OK;
}
int orc_set_root(void* s, int o) {
    Scene& sc = *(Scene*)s;
    if (!obj(sc, o)) return ORC_ERR_ARG;
    sc.root = obj(sc, o);
    return ORC_OK;
}
// World::new's lights (world.rs:18): each must be a Sphere or an XZ rectangle (the reference's two Light impls)
int orc_set_lights(void* s, int n, const int* ids, const double* flux3, const double* scales) {
    Scene& sc = *(Scene*)s;
    std::vector<const Hitable*> v;
    sc.light_flux.clear();
    sc.light_scale.clear();
    for (int i = 0; i < n; i++) {
        sc.light_flux.push_back(flux3 ? Vec3(flux3[3 * i], flux3[3 * i + 1], flux3[3 * i + 2]) : Vec3(1, 1, 1));
        sc.light_scale.push_back(scales ? scales[i] : 1.);
    }
    for (int i = 0; i < n; i++) {
        const Hitable* h = obj(sc, ids[i]);
        if (!h) return ORC_ERR_ARG;
        const Rect* r = dynamic_cast<const Rect*>(h);
        if (!(dynamic_cast<const Sphere*>(h) || (r && r->axis == 1))) return ORC_ERR_ARG;
        v.push_back(h);
    }
    sc.lights = v;
    return ORC_OK;
}
int orc_bounding_box(void* s, int o, double* out6) {
    Scene& sc = *(Scene*)s;
    AABB b;
    if (!obj(sc, o) || !obj(sc, o)->bounding_box(b)) return ORC_ERR_ARG;
    out6[0] = b.minimum.x; out6[1] = b.minimum.y; out6[2] = b.minimum.z;
    out6[3] = b.maximum.x; out6[4] = b.maximum.y; out6[5] = b.maximum.z;
    return ORC_OK;
}

// counters out: n_aabb, n_sphere, n_rect, n_tri, n_xform, n_segments, n_samples
int orc_render(void* s, int width, int height, int spp, int max_depth, double t_min, uint64_t seed, int x0, int y0, int x1, int y1,
               int n_jobs, int n_workers, double* out_rgb, uint64_t* counters7, int integrator) {
    Scene& sc = *(Scene*)s;
    if (width <= 0 || height <= 0 || spp <= 0 || x0 < 0 || y0 < 0 || x1 > width || y1 > height || x1 <= x0 || y1 <= y0 || !out_rgb) return ORC_ERR_ARG;
    RenderArgs a{width, height, spp, max_depth, t_min, seed, x0, y0, x1, y1, integrator, nullptr, 0.};
    if (integrator == 2) return ORC_ERR_ARG;  // use orc_render_sppm
    Counters c;
    int rc = render(sc, a, n_jobs, n_workers, out_rgb, &c);
    if (counters7) {
        counters7[0] = c.n_aabb; counters7[1] = c.n_sphere; counters7[2] = c.n_rect; counters7[3] = c.n_tri;
        counters7[4] = c.n_xform; counters7[5] = c.n_segments; counters7[6] = c.n_samples;
    }
    return rc;
}

// One World::hit for an explicit ray: out = {hit(0/1), t, p[3], normal[3], front_face, u, v, prim_id}
int orc_hit(void* s, int o, const double* orig, const double* dir, double t_min, double t_max, double* out12) {
    Scene& sc = *(Scene*)s;
    const Hitable* h = (o < 0) ? sc.root : obj(sc, o);
    if (!h) return ORC_ERR_ARG;
    Ctx cx;
    HitRecord rec;
    Ray r{Vec3(orig[0], orig[1], orig[2]), Vec3(dir[0], dir[1], dir[2])};
    bool ok;
    try {
        ok = h->hit(r, t_min, t_max, rec, cx);
    } catch (const UnitZero&) {
        return ORC_ERR_UNIT_ZERO;
    }
    out12[0] = ok ? 1. : 0.;
    if (ok) {
        out12[1] = rec.t;
        out12[2] = rec.p.x; out12[3] = rec.p.y; out12[4] = rec.p.z;
        out12[5] = rec.normal.x; out12[6] = rec.normal.y; out12[7] = rec.normal.z;
        out12[8] = rec.front_face ? 1. : 0.;
        out12[9] = rec.u; out12[10] = rec.v;
        out12[11] = (double)rec.prim_id;
    }
    return ORC_OK;
}
// The same with the RNG stream (seed, pixel, sample) in the context (ConstantMedium::hit draws from it); *draws = numbers consumed
int orc_hit_rng(void* s, int o, const double* orig, const double* dir, double t_min, double t_max, uint64_t seed, uint64_t pixel, uint64_t sample,
                double* out12, int* draws) {
    Scene& sc = *(Scene*)s;
    const Hitable* h = (o < 0) ? sc.root : obj(sc, o);
    if (!h) return ORC_ERR_ARG;
    Ctx cx;
    cx.rng = Rng(seed, pixel, sample);
    HitRecord rec;
    Ray r{Vec3(orig[0], orig[1], orig[2]), Vec3(dir[0], dir[1], dir[2])};
    bool ok;
    try {
        ok = h->hit(r, t_min, t_max, rec, cx);
    } catch (const UnitZero&) {
        return ORC_ERR_UNIT_ZERO;
    }
    if (draws) *draws = (int)cx.rng.draws;
    out12[0] = ok ? 1. : 0.;
    if (ok) {
        out12[1] = rec.t;
        out12[2] = rec.p.x; out12[3] = rec.p.y; out12[4] = rec.p.z;
        out12[5] = rec.normal.x; out12[6] = rec.normal.y; out12[7] = rec.normal.z;
        out12[8] = rec.front_face ? 1. : 0.;
        out12[9] = rec.u; out12[10] = rec.v;
        out12[11] = (double)rec.prim_id;
    }
    return ORC_OK;
}
// AABB::hit on an explicit box (KAT helper)
int orc_aabb_hit(const double* box6, const double* orig, const double* dir, double t_min, double t_max) {
    AABB b{Vec3(box6[0], box6[1], box6[2]), Vec3(box6[3], box6[4], box6[5])};
    Ctx cx;
    Ray r{Vec3(orig[0], orig[1], orig[2]), Vec3(dir[0], dir[1], dir[2])};
    return b.hit(r, t_min, t_max, cx) ? 1 : 0;
}
// Material::scatter + emitted on an explicit hit (KAT helper).
// in: ray o/d, p, normal, front_face, u, v; rng key; out: kind, has, ray o/d, att, emitted, draws
int orc_scatter(void* s, int m, const double* ray6, const double* p3, const double* n3, int front_face, double u, double v,
                uint64_t seed, uint64_t pixel, uint64_t sample, double* out14) {
    Scene& sc = *(Scene*)s;
    if (!mat(sc, m)) return ORC_ERR_ARG;
    Ctx cx;
    cx.rng = Rng(seed, pixel, sample);
    HitRecord rec;
    rec.p = Vec3(p3[0], p3[1], p3[2]);
    rec.normal = Vec3(n3[0], n3[1], n3[2]);
    rec.front_face = front_face != 0;
    rec.u = u; rec.v = v;
    rec.mat = mat(sc, m);
    Ray r{Vec3(ray6[0], ray6[1], ray6[2]), Vec3(ray6[3], ray6[4], ray6[5])};
    try {
        Vec3 e = rec.mat->emitted(rec);
        ScatterResult sr = rec.mat->scatter(r, rec, cx);
        out14[0] = (double)sr.kind;
        out14[1] = (sr.has_ray && sr.has_att) ? 1. : 0.;
        out14[2] = sr.ray.orig.x; out14[3] = sr.ray.orig.y; out14[4] = sr.ray.orig.z;
        out14[5] = sr.ray.dir.x; out14[6] = sr.ray.dir.y; out14[7] = sr.ray.dir.z;
        out14[8] = sr.att.x; out14[9] = sr.att.y; out14[10] = sr.att.z;
        out14[11] = e.x; out14[12] = e.y; out14[13] = e.z;
    } catch (const UnitZero&) {
        return ORC_ERR_UNIT_ZERO;
    }
    return ORC_OK;
}
// Camera ray for (pixel x,y; sample s): out = orig[3], dir[3]
int orc_camera_ray(void* s, int width, int height, int x, int y, uint64_t seed, uint64_t sample, double* out6) {
    Scene& sc = *(Scene*)s;
    if (!sc.cam_set) return ORC_ERR_ARG;
    Rng rng(seed, (uint64_t)y * (uint64_t)width + (uint64_t)x, sample);
    double u = ((double)x + rng.gen_f64()) / (double)(width - 1);
    double v = ((double)y + rng.gen_f64()) / (double)(height - 1);
    Ray r = sc.cam.get_ray(u, 1.0 - v, rng);
    out6[0] = r.orig.x; out6[1] = r.orig.y; out6[2] = r.orig.z;
    out6[3] = r.dir.x; out6[4] = r.dir.y; out6[5] = r.dir.z;
    return ORC_OK;
}

// SPPMIntegrator::new + capture_image (main.rs:52-54).  cfg5 = {iterations, photons_per_iter, k_global, k_caustic, max_bounces};
// stats out (optional): per pixel 10 doubles {g.flux[3], g.radius2, g.photons, c.flux[3], c.radius2, c.photons}; totals2 = photons stored.
int orc_render_sppm(void* s, int width, int height, int spp, int max_depth, double t_min, uint64_t seed, const int* cfg5, double alpha,
                    int n_workers, double* out_rgb, double* stats_out, uint64_t* totals2) {
    Scene& sc = *(Scene*)s;
    if (width <= 0 || height <= 0 || spp < 0 || !cfg5) return ORC_ERR_ARG;
    SppmConfig cfg;
    cfg.iterations = cfg5[0]; cfg.photons_per_iter = cfg5[1]; cfg.k_global = cfg5[2]; cfg.k_caustic = cfg5[3]; cfg.max_bounces = cfg5[4];
    cfg.alpha = alpha;
    std::vector<SPPMPixel> px;
    uint64_t tg = 0, tc = 0;
    int rc = sppm_prepass(sc, cfg, width, height, seed, n_workers, px, &tg, &tc);
    if (rc < 0) return rc;
    if (totals2) { totals2[0] = tg; totals2[1] = tc; }
    if (stats_out)
        for (size_t i = 0; i < px.size(); i++) {
            double* o = stats_out + 10 * i;
            const SPPM* two[2] = {&px[i].global, &px[i].caustic};
            for (int k = 0; k < 2; k++) {
                o[5 * k] = two[k]->flux.x; o[5 * k + 1] = two[k]->flux.y; o[5 * k + 2] = two[k]->flux.z;
                o[5 * k + 3] = two[k]->radius2; o[5 * k + 4] = (double)two[k]->photons;
            }
        }
    if (spp == 0 || !out_rgb) return ORC_OK;
    RenderArgs a{width, height, spp, max_depth, t_min, seed, 0, 0, width, height, 2, px.data(),
                 (double)cfg.iterations * (double)cfg.photons_per_iter};
    return render(sc, a, 64, n_workers, out_rgb, nullptr);
}

void orc_tonemap_u8(const double* rgb, size_t n_channels, uint8_t* out) {
    for (size_t i = 0; i < n_channels; i++) out[i] = tonemap_channel(rgb[i]);
}
// RNG KAT: first n u64 and f64 draws of stream (seed, pixel, sample)
void orc_rng_stream(uint64_t seed, uint64_t pixel, uint64_t sample, int n, uint64_t* out_u64) {
    Rng r(seed, pixel, sample);
    for (int i = 0; i < n; i++) out_u64[i] = r.next_u64();
}
void orc_rng_f64(uint64_t seed, uint64_t pixel, uint64_t sample, int n, double* out) {
    Rng r(seed, pixel, sample);
    for (int i = 0; i < n; i++) out[i] = r.gen_f64();
}
void orc_rng_range(uint64_t seed, uint64_t pixel, uint64_t sample, int n, double lo, double hi, double* out) {
    Rng r(seed, pixel, sample);
    for (int i = 0; i < n; i++) out[i] = r.gen_range(lo, hi);
}
// Sampling helpers KAT: which = 0 in_unit_sphere, 1 unit_vector, 2 in_unit_disk, 3 in_hemisphere(n)
int orc_sample_helper(int which, uint64_t seed, uint64_t pixel, uint64_t sample, const double* n3, double* out3) {
    Rng r(seed, pixel, sample);
    Vec3 v;
    try {
        switch (which) {
            case 0: v = random_in_unit_sphere(r); break;
            case 1: v = random_unit_vector(r); break;
            case 2: v = random_in_unit_disk(r); break;
            case 3: v = random_in_hemisphere(r, Vec3(n3[0], n3[1], n3[2])); break;
            default: return ORC_ERR_ARG;
        }
    } catch (const UnitZero&) {
        return ORC_ERR_UNIT_ZERO;
    }
    out3[0] = v.x; out3[1] = v.y; out3[2] = v.z;
    return ORC_OK;
}

// Vec3 KAT entry (replays raytracer/src/vec3.rs:425-564).
// op: 0 add 1 add_f64 2 sub 3 sub_f64 4 dot 5 mul_f64 6 div 7 elemul 8 cross 9 neg
//     10 squared_length 11 length 12 unit 13 reflect 14 refract(s = eta)
// returns ORC_ERR_UNIT_ZERO where Rust would panic.
int orc_vec3_op(int op, const double* a3, const double* b3, double s, double* out3) {
    Vec3 a(a3[0], a3[1], a3[2]), b(0, 0, 0), r(0, 0, 0);
    if (b3) b = Vec3(b3[0], b3[1], b3[2]);
    try {
        switch (op) {
            case 0: r = v_add(a, b); break;
            case 1: r = v_adds(a, s); break;
            case 2: r = v_sub(a, b); break;
            case 3: r = v_subs(a, s); break;
            case 4: r = Vec3(v_dot(a, b), 0, 0); break;
            case 5: r = v_muls(a, s); break;
            case 6: r = v_divs(a, s); break;
            case 7: r = v_elemul(a, b); break;
            case 8: r = v_cross(a, b); break;
            case 9: r = v_neg(a); break;
            case 10: r = Vec3(v_sqlen(a), 0, 0); break;
            case 11: r = Vec3(v_len(a), 0, 0); break;
            case 12: r = v_unit(a); break;
            case 13: r = reflect(a, b); break;
            case 14: r = refract(a, b, s); break;
            default: return ORC_ERR_ARG;
        }
    } catch (const UnitZero&) {
        return ORC_ERR_UNIT_ZERO;
    }
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
    return ORC_OK;
}
double orc_schlick(double cosine, double ref_idx) { return Dielectric::reflectance(cosine, ref_idx); }

}  // extern "C"
