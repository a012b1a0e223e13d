// HIP kernels of the radiance path for gfx950 (MI355X, wave64).
//
// pt_kernel  -- persistent path tracer.  256 workgroups x 1024 threads (4 waves per SIMD) stay resident for the whole
//   launch; a wave takes work units = (8x8 pixel tile, <= 8 sample indices) from a global counter and keeps its lanes
//   busy with in-wave path regeneration: lanes whose paths ended (once 8 of them are free) pull the next (pixel, sample)
//   of the unit's pool (wave64 ballot + prefix popcount), and the next unit is fetched as soon as the pool is empty, so
//   path lengths of 1 to 50 segments never drain a wave.  Per segment: closest hit (kernel 2: SAH BVH2 with conservative
//   f32 boxes, per-lane stack in LDS; kernel 1: the reference-order program), hit record, emission + scatter.
//   The flattened scene (common/flat.h) is staged once per workgroup into LDS when it fits (its traversal tables;
//   scene_500: 72 KB + 56 KB of stacks of the CU's 160 KB), otherwise read through L2 with the top of the BVHs cached.
//   Each finished path stores its radiance to its unit's buffer in the wave's ring; a complete unit is folded into the f64
//   accumulator IN SAMPLE ORDER (the reference's `pixel_color += sample` loop, camera.rs:96-101; per-tile tickets order the
//   units across waves), which makes the image independent of how work was scheduled.
// finalize / assemble -- `pixel_color /= spp` (camera.rs:102) and the tile stitch
//   (camera.rs:115-123).
//
// Arithmetic: f64 throughout, compiled with -ffp-contract=off, IEEE divide and
// sqrt; every expression keeps the reference's operation order (file:line cited
// per function) so results are bit-identical to the CPU oracle.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "../common/flat.h"
#include "../common/detlog.h"
#include "../common/detsin.h"
#include "../common/rng.h"
#include "../common/schedule.h"
#include "device.h"

namespace rtamd {

#ifndef PT_BLOCK
#define PT_BLOCK 1024
#endif

#define HIP_CHECK(expr)                                                                                  \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess)                                                                            \
            throw RtError((_e == hipErrorNoDevice || _e == hipErrorInvalidDevice) ? RT_ERR_NO_DEVICE : RT_ERR_HIP, \
                          std::string(#expr) + ": " + hipGetErrorString(_e));                            \
    } while (0)

// ---- phase statistics (tools-only build: tools/build_variant.sh phase -DRTAMD_PHASE_STATS; the product carries none of it) ----
// Per workgroup in LDS, flushed to g_phase at the end of pt_kernel: [0..15] wave clocks (s_memtime) spent in a phase, [16..31] how
// often a WAVE executed the phase's block, [32..47] how many LANES took part.  A phase is timed by its own clock pair, read by the
// lanes that execute it (the clock is scalar: one value per wave), and booked by the first active lane.
// phases: 0 regeneration, 1 traverse (all of it), 2 materialize, 3 shade (all of it), 4 Lambertian / DiffuseLight branch, 5 Isotropic,
// 6 Metal, 7 Dielectric, 8 what follows traverse in a segment (materialize + shade + radiance + mixture / store), 9 leaf sections of
// the traversal; events only: 10 inner-node steps, 11 leaf items, 12 cube tests, 13 segments (alive lanes), 14 mixture step, 15 inner-node steps inside an instance
#ifdef RTAMD_PHASE_STATS
__shared__ unsigned long long s_ph[48];
__device__ unsigned long long g_phase[48];
__device__ __forceinline__ bool ph_first() { return (int)(__ffsll((long long)__ballot(1)) - 1) == (int)(threadIdx.x & 63); }
#define PH_CLK() __builtin_amdgcn_s_memtime()
#define PH_ADD(i, dt)                                                         \
    do {                                                                      \
        if (ph_first()) atomicAdd(&s_ph[(i)], (unsigned long long)(dt));      \
    } while (0)
#define PH_EV(i)                                                              \
    do {                                                                      \
        const unsigned long long _m = __ballot(1);                            \
        if (ph_first()) {                                                     \
            atomicAdd(&s_ph[16 + (i)], 1ull);                                 \
            atomicAdd(&s_ph[32 + (i)], (unsigned long long)__popcll(_m));     \
        }                                                                     \
    } while (0)
#define PH_BEGIN(v) const unsigned long long v = PH_CLK()
#define PH_END(i, v)              \
    do {                          \
        PH_ADD(i, PH_CLK() - v);  \
        PH_EV(i);                 \
    } while (0)
#else
#define PH_ADD(i, dt) \
    do {              \
    } while (0)
#define PH_EV(i) \
    do {         \
    } while (0)
#define PH_BEGIN(v) \
    do {            \
    } while (0)
#define PH_END(i, v) \
    do {             \
    } while (0)
#endif

// ---------------------------------------------------------------- math ----
struct D3 {
    double x, y, z;
};
#define DEV __device__ __forceinline__
DEV D3 mk(double x, double y, double z) { D3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV D3 add(D3 a, D3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV D3 sub(D3 a, D3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV D3 muls(D3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
DEV D3 divs(D3 a, double s) { return mk(a.x / s, a.y / s, a.z / s); }
DEV D3 neg(D3 a) { return mk(-a.x, -a.y, -a.z); }
DEV D3 elemul(D3 a, D3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
DEV double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // vec3.rs:335-341
DEV double sqlen(D3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }     // vec3.rs:61-63
DEV D3 cross(D3 a, D3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DEV double comp(D3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
// Measured and dropped in round 3 (both forms exact in the whole suite): (a) the three divisions of unit() / of (p - c) / radius
// sharing ONE reciprocal refinement of the hardware's division pipeline (the three v_div_scale of the denominator, taken as the
// compiler's expansion takes them, must return the denominator itself; else plain divisions): 33 -> 29 VALU and two v_rcp_f64 less
// per triple, 2838 against 2836 Msamples/s; (b) division by a correctly rounded reciprocal with one fma correction behind an exponent
// guard: +-0 as in round 2.  Only (b) WITHOUT its guard gains (+1.8 %), and that is not exact at the ends of the exponent range.
// Vec3::unit, vec3.rs:85-90 ; the panic becomes a sticky error flag
DEV D3 unit(D3 a, int* err) {
    double l = sqrt(sqlen(a));
    if (l == 0.) atomicOr(err, 1);
    return divs(a, l);
}
DEV bool near_zero(D3 a) {  // vec3.rs:92-95
    const double S = 1e-8;
    return (fabs(a.x) < S) && (fabs(a.y) < S) && (fabs(a.z) < S);
}
DEV D3 reflect(D3 v, D3 n) { return sub(v, muls(n, 2. * dot(v, n))); }  // vec3.rs:163-165
DEV D3 refract(D3 uv, D3 n, double eta) {                              // vec3.rs:167-172
    double cos_theta = fmin(dot(neg(uv), n), 1.0);
    D3 perp = muls(add(uv, muls(n, cos_theta)), eta);
    D3 par = muls(n, -sqrt(fabs(1.0 - sqlen(perp))));
    return add(perp, par);
}
DEV D3 random_in_unit_sphere(Rng& rng) {  // vec3.rs:111-129 (Marsaglia; a point ON the sphere, Q3)
    double u, v, r2;
    for (;;) {
        u = rng.gen_range_pm1();
        v = rng.gen_range_pm1();
        r2 = u * u + v * v;
        if (r2 <= 1.) break;
    }
    double q = sqrt(1. - r2);
    return mk(2. * u * q, 2. * v * q, 1. - 2. * r2);
}
DEV D3 random_in_unit_disk(Rng& rng) {  // vec3.rs:153-162
    for (;;) {
        double a = rng.gen_range_pm1();
        double b = rng.gen_range_pm1();
        if (a * a + b * b + 0. * 0. >= 1.) continue;
        return mk(a, b, 0.);
    }
}
DEV D3 xf_point(const double* t, D3 p) {  // vec3.rs:174-178 (w = 1)
    return mk(t[0] * p.x + t[1] * p.y + t[2] * p.z + t[3] * 1., t[4] * p.x + t[5] * p.y + t[6] * p.z + t[7] * 1.,
              t[8] * p.x + t[9] * p.y + t[10] * p.z + t[11] * 1.);
}
DEV D3 xf_dir(const double* t, D3 p) {  // vec3.rs:180-184 (w = 0)
    return mk(t[0] * p.x + t[1] * p.y + t[2] * p.z + t[3] * 0., t[4] * p.x + t[5] * p.y + t[6] * p.z + t[7] * 0.,
              t[8] * p.x + t[9] * p.y + t[10] * p.z + t[11] * 0.);
}

// ------------------------------------------------------------- scene ------
#define AS_G __attribute__((address_space(1)))
#define AS_L __attribute__((address_space(3)))
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct Acc {  // typed views into the blob (LDS or global; the address space is inferred per instantiation)
    const uint2* meta;
    const double2* boxes;    // 3 x double2 per box: (minx,miny) (minz,maxx) (maxy,maxz)
    const double2* spheres;  // 2 x double2: (cx,cy) (cz,r)
    const int* sphere_mat;
    const double2* rects;    // 3 x double2: (a0,b0) (a1,b1) (k,-)
    const int* rect_mat;
    const uint4* tris;       // a,b,c,mat
    const double2* tripre;   // 5 x double2 per triangle: pa, e0 = pb-pa, e1 = pc-pa (mesh.rs:69, hoisted to commit time)
    const double2* tripre2;  // the same records in accel ITEM order (a leaf's triangles are contiguous)
    const AS_L f32x4* n2_top;  // LDS copy of the first n2_top_count Node2 (the shallowest levels) when the scene itself is not in LDS;
                               // typed with its address space: a generic pointer here lets the compiler fold the cached / uncached
                               // node loads into one FLAT load of a selected address
    uint32_t n2_top_count;
    uint32_t n2w_lds;        // LDS byte address of the NodeW table (kernel 2 with the scene in LDS; see box32w)
    const double* xforms;    // 32 per transform: M^-1 then M, row-major
    const MatDev* mats;
    const TexDev* texs;
    const double* vpos;
    const double* vnrm;
    const uint8_t* texels;   // always global
    uint32_t n_nodes;
    // accel (kernel 2)
    const float4* n2;        // 4 x float4 per Node2
    const uint2* items2;     // {kind | payload << 4, order}
    const uint2* inst2;      // {xform, root ref}
    uint32_t root2;
    const uint2* lights;     // {NK_SPHERE | NK_RECT_XZ, payload}
    uint32_t n_lights;
    const MediumDev* media;  // always global
    const double* msph;      // moving spheres (D9), always global: 10 f64 each
    const uint32_t* tie_view;  // FlatView::off_tie_view, always global: read on exact ties only (tie_resolve)
    uint32_t time_lds;       // LDS byte address of the per-lane ray times (D9; 0: the scene has no moving sphere, every time is 0)
    // kernel 5's serving waves: compact object-space data (flat.h "Compact instance data")
    const uint4* n2q;        // 2 x uint4 per NodeQ
    const uint4* n2q_top;    // LDS copy of the first n2q_top_count NodeQ
    uint32_t n2q_top_count;
    const uint4* tri32;      // 3 x uint4 per Tri32
    const double* qgrid;     // 8 per instance
};
template <class P>
DEV Acc make_acc(P hot, const char* gbase, const FlatView& v) {  // hot: LDS copy or the global blob; cold part always global
    Acc a;
    a.meta = (const uint2*)(hot + v.off_meta);
    a.boxes = (const double2*)(hot + v.off_boxes);
    a.spheres = (const double2*)(hot + v.off_spheres);
    a.rects = (const double2*)(hot + v.off_rects);
    a.tris = (const uint4*)(hot + v.off_tris);
    a.tripre = (const double2*)(hot + v.off_tripre);
    a.tripre2 = (const double2*)(hot + v.off_tripre2);
    a.n2_top = nullptr;
    a.n2_top_count = 0;
    a.n2w_lds = 0;
    a.xforms = (const double*)(hot + v.off_xforms);
    a.vpos = (const double*)(gbase + v.off_vpos);
    a.sphere_mat = (const int*)(gbase + v.off_sphere_mat);
    a.rect_mat = (const int*)(gbase + v.off_rect_mat);
    a.mats = (const MatDev*)(gbase + v.off_mats);
    a.texs = (const TexDev*)(gbase + v.off_texs);
    a.vnrm = (const double*)(gbase + v.off_vnrm);
    a.texels = (const uint8_t*)(gbase + v.off_texels);
    a.n_nodes = v.n_nodes;
    a.n2 = (const float4*)(hot + v.off_n2);
    a.items2 = (const uint2*)(hot + v.off_items2);
    a.inst2 = (const uint2*)(hot + v.off_inst2);
    a.root2 = v.root2;
    a.lights = (const uint2*)(gbase + v.off_lights);
    a.n_lights = v.n_lights;
    a.media = (const MediumDev*)(gbase + v.off_media);
    a.msph = (const double*)(gbase + v.off_msph);
    a.tie_view = (const uint32_t*)(gbase + v.off_tie_view);
    a.time_lds = 0u;
    return a;
}

struct CamK {
    D3 origin, llc, horizontal, vertical, u, v;
    double lens_radius;
};
struct RenderK {
    int width, height, max_depth;
    double t_min;
    uint64_t seed;
    int s_begin, s_end;  // sample indices of this launch
    int sub_spp, subs_per_tile, n_units;
    int tiles_x, rank, world;
    int tiles_owned;  // jobs are dealt sample-major: job j = (tile j % tiles_owned, sample blocks [j / tiles_owned * job_units, + job_units))
    int job_units;    // consecutive units of one tile a wave takes at a time (one ticket hand-off per job); n_units counts JOBS
    // The schedule of one launch (kernels 1, 2, 5; see make_schedule, host/schedule.cpp): every tile's samples [s_begin, s_end) are cut into the same sequence
    // of units, in up to SCHED_LEVELS levels of decreasing unit / job size.  Level l: rounds (jobs per tile) from lvl[l][0], units from
    // lvl[l][1], samples from lvl[l][2], lvl[l][3] samples per unit, lvl[l][4] units per job; lvl[n_levels] = {rounds, units, s_end, 0, 0}.
    int lvl[5][5];
    int pool_mode;  // the wave's unit buffers are a pool folded out of order (next_unit_pool) instead of a FIFO: launches of single-unit jobs
    const double* sppm_est;  // INTEG 2: per pixel {caustic estimate[3], global estimate[3]}, index y*width + x
    int n_top;               // kernel 2 with the scene in L2/HBM: number of (depth-sorted) Node2 cached in LDS
    int n_topq;              // kernel 5: number of NodeQ cached in LDS for the serving waves
    int coop_pool;           // kernel 5: parked-path slots in use (<= COOP_POOL)
    int coop_stack;          // kernel 5: stack entries per lane
    double time0, time1;     // D9: the camera's shutter; time1 > time0: every sample draws its time after the lens sample
    int time_slots;          // D9: 1 = the scene has moving spheres: 8 bytes of LDS per lane (behind the launch constants) hold the paths' times
};

// ------------------------------------------------------ intersection ------
// AABB::hit, aabb.rs:15-32, with 1/dir hoisted out of the node loop (same values).
// min only grows and max only shrinks across the three axes, so testing `max <= min`
// once at the end decides exactly as the reference's per-axis early returns do.
DEV bool aabb_hit(const double2* b, D3 o, D3 inv, double t_min, double t_max) {
    double2 v0 = b[0], v1 = b[1], v2 = b[2];
    double mn = t_min, mx = t_max;
    {
        double t0 = (v0.x - o.x) * inv.x, t1 = (v1.y - o.x) * inv.x;
        if (inv.x < 0.0) { double s = t0; t0 = t1; t1 = s; }
        mn = fmax(mn, t0);
        mx = fmin(mx, t1);
    }
    {
        double t0 = (v0.y - o.y) * inv.y, t1 = (v2.x - o.y) * inv.y;
        if (inv.y < 0.0) { double s = t0; t0 = t1; t1 = s; }
        mn = fmax(mn, t0);
        mx = fmin(mx, t1);
    }
    {
        double t0 = (v1.x - o.z) * inv.z, t1 = (v2.y - o.z) * inv.z;
        if (inv.z < 0.0) { double s = t0; t0 = t1; t1 = s; }
        mn = fmax(mn, t0);
        mx = fmin(mx, t1);
    }
    return !(mx <= mn);
}
// Sphere::hit's root selection, sphere.rs:24-43 ; a = |dir|^2 hoisted per ray
// RT_SPHERE_F32_REJECT (A/B switch, off; round 5's experiment for the headline scene): the sign of the discriminant decided in f32 where it is
// certain.  oc comes from the f64 subtraction the reference makes anyway; with oc32, d32 = its and the direction's f32 roundings (relative
// error e = 2^-24 per component), M = max |oc_i|, D = max |d_i|, r32 the radius rounded:  hb32 is within 9 e M D of half_b, c32 within 9 e M^2
// + 3 e r^2 of c, a32 within 9 e D^2 of a, hence disc32 = hb32^2 - a32 c32 within 64 e D^2 (M + r)^2 of disc; the test uses 2^-17 D^2 (M + r)^2.
// Bit-exact (parity tests green with it) and SLOWER: headline 2 704 against 2 812 Msamples/s, C2 2 803 / 2 911, Cornell 2 441 / 2 490 (two runs
// each): a miss saves the 15 f64 operations between oc and the discriminant (60 issue cycles) and pays 7 conversions + 19 f32 operations for them,
// a hit pays them on top.
#ifndef RT_SPHERE_F32_REJECT
#define RT_SPHERE_F32_REJECT 0
#endif
DEV bool sphere_hit(const double2* s, D3 o, D3 d, double a, double t_min, double t_max, double& t_out) {
    double2 c0 = s[0], c1 = s[1];
    D3 oc = mk(o.x - c0.x, o.y - c0.y, o.z - c1.x);
    double radius = c1.y;
    if (RT_SPHERE_F32_REJECT) {
        const float x = (float)oc.x, y = (float)oc.y, z = (float)oc.z, dx = (float)d.x, dy = (float)d.y, dz = (float)d.z, r32 = (float)radius;
        const float hb = __builtin_fmaf(z, dz, __builtin_fmaf(y, dy, x * dx));
        const float a32 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
        const float c32 = __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x)) - r32 * r32;
        const float disc32 = __builtin_fmaf(hb, hb, -(a32 * c32));
        const float m = fmaxf(fmaxf(fabsf(x), fabsf(y)), fabsf(z)) + fabsf(r32), dm = fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz));
        const float bound = 7.62939453125e-06f * (dm * m) * (dm * m);  // 2^-17 D^2 (M + r)^2
        if (disc32 < -bound) return false;  // disc < 0 for certain (NaN / inf: the comparison fails, the f64 test decides)
    }
    double half_b = dot(oc, d);
    double c = sqlen(oc) - radius * radius;
    double disc = half_b * half_b - a * c;
    if (disc < 0.) return false;
    double sq = sqrt(disc);
    double root = (-half_b - sq) / a;
    if (!(root >= t_min && root <= t_max)) root = (-half_b + sq) / a;
    if (!(root >= t_min && root <= t_max)) return false;
    t_out = root;
    return true;
}
// XY/XZ/YZRectangle::hit, rectangle.rs:15-34,53-72,90-109 (axis = constant axis).
// No zero-direction guard: NaN/inf t falls through the rejects exactly as in the reference.
DEV bool rect_hit(const double2* r, int axis, D3 o, D3 d, double t_min, double t_max, double& t_out) {
    double2 r0 = r[0], r1 = r[1], r2 = r[2];
    double t = (r2.x - comp(o, axis)) / comp(d, axis);
    if (t < t_min || t > t_max) return false;
    D3 p = add(o, muls(d, t));
    double a, b;
    if (axis == 2) { a = p.x; b = p.y; }
    else if (axis == 1) { a = p.x; b = p.z; }
    else { a = p.y; b = p.z; }
    if (a < r0.x || a > r1.x || b < r0.y || b > r1.y) return false;
    t_out = t;
    return true;
}
// D9 (book-2 extension, no reference code).  A path's time lives in LDS (one f64 per lane, written when the path is generated) so that
// the one primitive that reads it costs the others no register; the scene's kinds_mask says whether the slots exist.
DEV double ray_time(const Acc& A) { return A.time_lds ? *(const AS_L double*)(uintptr_t)(A.time_lds + 8u * threadIdx.x) : 0.; }
// moving_sphere::center(time) = center0 + ((time - time0) / (time1 - time0)) (center1 - center0)
DEV D3 msphere_center(const double* q, double time) {
    const D3 c0 = mk(q[0], q[1], q[2]), c1 = mk(q[3], q[4], q[5]);
    return add(c0, muls(sub(c1, c0), (time - q[6]) / (q[7] - q[6])));
}
// moving_sphere::hit = Sphere::hit (sphere.rs:24-43) around center(r.time); out of line: rare, and the sphere test's twin in the loops
__device__ __attribute__((noinline)) bool msphere_hit(const double* q, double time, double ox, double oy, double oz, double dx, double dy, double dz, double a,
                                                       double t_min, double t_max, double* t_out) {
    const D3 c = msphere_center(q, time), o = mk(ox, oy, oz), d = mk(dx, dy, dz);
    const D3 oc = sub(o, c);
    const double radius = q[8];
    const double half_b = dot(oc, d);
    const double cc = sqlen(oc) - radius * radius;
    const double disc = half_b * half_b - a * cc;
    if (disc < 0.) return false;
    const double sq = sqrt(disc);
    double root = (-half_b - sq) / a;
    if (!(root >= t_min && root <= t_max)) root = (-half_b + sq) / a;
    if (!(root >= t_min && root <= t_max)) return false;
    *t_out = root;
    return true;
}
// perlin::noise / turb and noise_texture::value of the book over the texture's tables (768 f64 gradient vectors, 768 permutation bytes)
__device__ __attribute__((noinline)) double noise_marble(const uint8_t* tab, double scale, double px, double py, double pz) {
    const double* ranvec = (const double*)tab;
    const uint8_t* perm = tab + 768 * sizeof(double);
    double accum_t = 0., weight = 1.;
    double x = px, y = py, z = pz;
    for (int oct = 0; oct < 7; oct++) {
        const double fx = floor(x), fy = floor(y), fz = floor(z);
        const double u = x - fx, v = y - fy, w = z - fz;
        const int i = (int)fx, j = (int)fy, k = (int)fz;
        const double uu = u * u * (3. - 2. * u), vv = v * v * (3. - 2. * v), ww = w * w * (3. - 2. * w);
        double accum = 0.;
        for (int di = 0; di < 2; di++)
            for (int dj = 0; dj < 2; dj++)
                for (int dk = 0; dk < 2; dk++) {
                    const int h = perm[(i + di) & 255] ^ perm[256 + ((j + dj) & 255)] ^ perm[512 + ((k + dk) & 255)];
                    const double cx = ranvec[3 * h], cy = ranvec[3 * h + 1], cz = ranvec[3 * h + 2];
                    const double wx = u - di, wy = v - dj, wz = w - dk;
                    accum += (di * uu + (1 - di) * (1. - uu)) * (dj * vv + (1 - dj) * (1. - vv)) * (dk * ww + (1 - dk) * (1. - ww)) * (cx * wx + cy * wy + cz * wz);
                }
        accum_t += weight * accum;
        weight *= 0.5;
        x = x * 2.;
        y = y * 2.;
        z = z * 2.;
    }
    const double turb = fabs(accum_t);
    return 0.5 * (1. + det_sin(scale * pz + 10. * turb));
}

// Cube::hit = self.sides.hit(r, t_min, t_max) (cube.rs:64-66): the list scan of hit.rs:56-67 over the six rectangles of Cube::new
// (cube.rs:17-54) in their order -- XY z=min.z, XY z=max.z, XZ y=min.y, XZ y=max.y, YZ x=min.x, YZ x=max.x -- each one rectangle.rs's
// test (:20-25, :58-63, :95-100: t = (k - o) / d; reject t < t_min || t > closest_so_far; then the two bounds) with the closest hit so
// far shrinking from side to side, so a later side wins an exact tie and a NaN t (a ray in a side's plane) propagates as
// in the reference.  c: (min.x, min.y) (min.z, max.x) (max.y, max.z).  Returns the winning side.
// Kept OUT OF LINE: inlined into the leaf loops it raised the spill count of every GENERAL kernel by a quarter (pt_kernel<true, true, 2, 0>:
// 100 -> 126 spilled VGPRs) and cost the Cornell box 17 % although one ray in a dozen meets its cube.
struct CubeHit {
    double t;
    int side;  // -1: no side was hit
};
__device__ __attribute__((noinline)) CubeHit cube_hit_sides(double mnx, double mny, double mnz, double mxx, double mxy, double mxz, double ox, double oy,
                                                            double oz, double dx, double dy, double dz, double t_min, double t_max) {
    double best = t_max;
    int side = -1;
#define RT_CUBE_SIDE(S, K, OK, DK, OA, DA, A0, A1, OB, DB, B0, B1)     \
    {                                                                  \
        const double t = ((K) - (OK)) / (DK);                          \
        if (!(t < t_min || t > best)) {                                \
            const double pa = (OA) + (DA) * t, pb = (OB) + (DB) * t;   \
            if (!(pa < (A0) || pa > (A1) || pb < (B0) || pb > (B1))) { \
                best = t;                                              \
                side = (S);                                            \
            }                                                          \
        }                                                              \
    }
    RT_CUBE_SIDE(0, mnz, oz, dz, ox, dx, mnx, mxx, oy, dy, mny, mxy)
    RT_CUBE_SIDE(1, mxz, oz, dz, ox, dx, mnx, mxx, oy, dy, mny, mxy)
    RT_CUBE_SIDE(2, mny, oy, dy, ox, dx, mnx, mxx, oz, dz, mnz, mxz)
    RT_CUBE_SIDE(3, mxy, oy, dy, ox, dx, mnx, mxx, oz, dz, mnz, mxz)
    RT_CUBE_SIDE(4, mnx, ox, dx, oy, dy, mny, mxy, oz, dz, mnz, mxz)
    RT_CUBE_SIDE(5, mxx, ox, dx, oy, dy, mny, mxy, oz, dz, mnz, mxz)
#undef RT_CUBE_SIDE
    CubeHit r;
    r.t = best;
    r.side = side;
    return r;
}
DEV bool cube_hit(const double2* c, D3 o, D3 d, double t_min, double t_max, double& t_out, uint32_t& side_out) {
    const double2 c0 = c[0], c1 = c[1], c2 = c[2];
    const CubeHit r = cube_hit_sides(c0.x, c0.y, c1.x, c1.y, c2.x, c2.y, o.x, o.y, o.z, d.x, d.y, d.z, t_min, t_max);
    t_out = r.t;
    side_out = r.side < 0 ? 0u : (uint32_t)r.side;
    return r.side >= 0;
}
DEV D3 ld3(const double* p, uint32_t i) { return mk(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }
// Triangle::hit, mesh.rs:57-102 ; returns t and the barycentrics b1,b2
DEV bool tri_hit_v(D3 pa, D3 e0, D3 e1, D3 o, D3 dir, double t_min, double t_max, double& t_out, double& b1o, double& b2o);
DEV bool tri_hit(const double2* q, D3 o, D3 dir, double t_min, double t_max, double& t_out, double& b1o, double& b2o) {
    double2 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4];
    return tri_hit_v(mk(q0.x, q0.y, q1.x), mk(q1.y, q2.x, q2.y), mk(q3.x, q3.y, q4.x), o, dir, t_min, t_max, t_out, b1o, b2o);
}
DEV bool tri_hit_v(D3 pa, D3 e0, D3 e1, D3 o, D3 dir, double t_min, double t_max, double& t_out, double& b1o, double& b2o) {
    D3 s0 = cross(dir, e1);
    double dd = dot(s0, e0);
    if (dd == 0.0) return false;
    double div = 1.0 / dd;
    D3 d = sub(o, pa);
    double b1 = dot(d, s0) * div;
    if (b1 < 0.0 || b1 > 1.0) return false;
    D3 s1 = cross(d, e0);
    double b2 = dot(dir, s1) * div;
    if (b2 < 0.0 || b1 + b2 > 1.0) return false;
    double t = dot(e1, s1) * div;
    if (t < t_min || t > t_max) return false;
    t_out = t;
    b1o = b1;
    b2o = b2;
    return true;
}

struct Hit {
    double t;
    int node;     // DFS index of the winning leaf in the reference-order program, -1 = miss
    int xf;       // enclosing Transform (xform index) or -1
    uint32_t kp;  // kind | payload << 4 of the winning leaf
};

// EXACT ties in the accel walks.  The reference gives a tie to the object it visits LATER (inclusive ranges, Q5) -- if it visits it:
// BVHNode::hit first tests the enclosing node's box with t_max = closest so far = the tied t (bvh.rs:88), and AABB::hit rejects an
// interval that has shrunk to a point (aabb.rs:28-30).  A later object whose innermost BVHNode box BEGINS at t (a Cube's exact box,
// cube.rs:67-69, entered through a face that is coplanar with what was hit before: a rectangle lying on the face, the face it shares
// with the cube the ray is leaving, a floor under a box lit from below) is therefore never visited and the earlier object keeps the
// hit.  The accel walk itself keeps the round-3 rule (the later object wins) and only FLAGS the tie in the winner's Hit::xf (TIE_FLAG;
// any closer hit overwrites it): one compare and a select in the leaf loops.  When the walk is over and the flag is still there -- the
// best hit is shared by two objects -- the ray is walked once more, through the reference-order program with the reference's own
// box tests (tie_resolve = traverse<>, out of line), and that walk's answer is the hit.  (Cheaper forms of the rule were measured:
// a call in the leaf loop cost every kernel 6-18 %, a note of the earlier party kept in private memory 6-12 % of the GENERAL
// kernels -- the register allocator pays for a call site or a store sequence in the loop whether it is executed or not.)
// Kernels 5 and 6 (round 5) follow the same rule in their TIE variants, picked by the host for every scene that holds a rectangle or a
// cube (sphere / triangle boxes never begin exactly where another surface lies: spheres touch their box in six points, triangle boxes
// are padded by 0.1): the world-space walk notes ties among the world-level items (TIEDEF), the object-space walks of the instance
// service note a tie between a triangle and the best hit they were posted with (or an earlier triangle) in a per-lane bit that travels
// with the answer (bit 30 of the order word, same encoding as TIE_FLAG in Hit::xf), and whoever holds the path when a flag turns up
// -- the walking lane, or the lane that adopts the answer -- lets tie_resolve walk the WHOLE ray through the reference-order
// program: that answer is final for the segment, instances included, so the path's remaining deferred instances are dropped.
#ifdef RT_NO_TIE_RULE  // A/B build: the round-3 rule alone (the later object always wins a tie)
#define TIE_RULE 0
#else
#define TIE_RULE 1
#endif
#ifndef RT_TIE_NOTRACK
#define RT_TIE_NOTRACK 0
#endif
#define TIE_FLAG 0x40000000  // in Hit::xf (-1 or a small index): bits 30 and 31 differ <=> the hit is an exact tie of two objects
DEV bool tie_flagged(int xf) { return (((uint32_t)xf >> 30) & 1u) != ((uint32_t)xf >> 31); }
// NEAR ties (round 5).  "Exactly" above is too narrow: the reference's box test computes a box's entry as (min - o) * (1 / d) (aabb.rs:20-21)
// while a rectangle's own t is (k - o) / d and a triangle's comes out of Moeller-Trumbore -- three roundings of the same real number.  When
// the later object Y (exact box, entered through the face in question) and the earlier object X lie in ONE plane, the reference visits Y iff
// fl((k - o) * fl(1 / d)) < t_X, and that differs from "t_Y <= t_X" (what a closest-hit walk decides) whenever t_X falls into the one or two
// ulps between t_Y and Y's box entry: 1 ray in 10 000 on a mesh face lying on a cube face (tools/tie_soak.py inst, round 5: 0.26382042861594929
// for the triangle, ...923 for the cube side; the reference keeps the triangle).  Under a Transform the window is wider: the cube's t comes out
// of the object-space ray M^-1 o, whose rounding is an ulp of |o|, not of the distance travelled -- t is off by up to 2^-53 |o| / (t |d_axis|)
// relatively (cubes translated by lattice steps: 2-9 of 150 000 rays per scene differed with a window of 8 ulps).  So the walks flag every pair
// of candidates whose t differ by at most TIE_W - 1 = 2^-32 relatively (covers |o| up to a million times the distance travelled along the
// face's axis; untransformed boxes need 1.5 ulps; with 2^-40 ONE ray of 5 000 000 on the soak's lattices still differed: it started 1.7e-4 from
// the plane it grazed at 1.4 degrees, 2.6e4 times closer than it was to the origin), whichever of the two is closer, and let the
// reference-order walk decide:
// the primitive tests see [t_min, best * TIE_W] in the TIE variants, a candidate beyond `best` is never accepted, only noted.  A flag that was
// not needed -- two surfaces within 10^-12 of each other along the ray: contact lines, nothing else -- costs a re-walk and returns the same hit.
#ifndef TIE_W
#define TIE_W (1.0 + 2.3283064365386963e-10)  // 1 + 2^-32
#endif

// World::hit -> BVHNode::hit / Vec::hit / Transform::hit, flattened (common/flat.h).
// Visits nodes in the reference's own order; a leaf is accepted when t_min <= t <= best
// (inclusive, so a later leaf wins an exact tie -- sphere.rs:36, rectangle.rs:20, mesh.rs:96).
// MEDIA: the program contains ConstantMedium brackets (NK_MEDIUM_*, common/flat.h): between BEGIN and END the walk answers the
// two boundary queries of ConstantMedium::hit (medium.rs:26-27) in a scratch hit with its own range [lo, +inf), then END
// restores the outer best hit and makes the medium's one random draw from the path's stream.
// [n0, n1): the part of the program to walk (default: all of it; traverse2_media walks a medium's boundary subtree).
template <int GENERAL, bool MEDIA = false>
DEV Hit traverse(const Acc& A, D3 wo, D3 wd, double t_min, double t_max, Rng* rng = nullptr, uint32_t n0 = 0u, uint32_t n1 = 0xFFFFFFFFu) {
    D3 o = wo, d = wd;
    D3 inv = mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
    double a = sqlen(d);
    Hit h;
    h.t = t_max;
    h.node = -1;
    h.xf = -1;
    h.kp = 0;
    int cur_xf = -1;
    uint32_t n = n0;
    const uint32_t N = min(A.n_nodes, n1);
    const double t_min_outer = t_min;
    Hit h_outer = h;      // MEDIA: the outer best hit while a boundary query runs
    double t_a = 0.;      // MEDIA: t of boundary query A
    while (n < N) {
        uint2 m = A.meta[n];
        uint32_t kind = m.x & NK_MASK, pl = m.x >> NK_BITS;
        if (kind == NK_BOX) {
            n = aabb_hit(A.boxes + 3 * pl, o, inv, t_min, h.t) ? n + 1 : m.y;
        } else if (MEDIA && kind == NK_MEDIUM_BEGIN) {  // rec1 = boundary.hit(r, -inf, +inf), medium.rs:26
            h_outer = h;
            h.t = INFINITY;
            h.node = -1;
            t_min = -INFINITY;
            n++;
        } else if (MEDIA && kind == NK_MEDIUM_MID) {  // rec2 = boundary.hit(r, rec1.t + 0.0001, +inf), medium.rs:27
            if (h.node < 0) {
                n = m.y;  // no rec1: straight to END, which then restores and returns None
            } else {
                t_a = h.t;
                t_min = h.t + 0.0001;
                h.t = INFINITY;
                h.node = -1;
                n++;
            }
        } else if (MEDIA && kind == NK_MEDIUM_END) {  // medium.rs:28-50
            const bool both = h.node >= 0;  // reached with rec1 AND rec2 (a missing rec1 jumped here with node < 0 as well)
            const double t_b = h.t;
            h = h_outer;
            t_min = t_min_outer;
            if (both) {
                double r1 = fmax(t_a, t_min);
                const double r2 = fmin(t_b, h.t);
                if (!(r1 >= r2)) {
                    r1 = fmax(r1, 0.);
                    const double ray_length = sqrt(a);
                    const double distance_inside_boundary = (r2 - r1) * ray_length;
                    const double hit_distance = A.media[pl].neg_inv_density * det_ln(rng->gen_f64());  // the only draw, medium.rs:37-38
                    if (!(hit_distance > distance_inside_boundary)) {
                        h.t = r1 + hit_distance / ray_length;
                        h.node = (int)n;
                        h.xf = cur_xf;
                        h.kp = m.x;
                    }
                }
            }
            n++;
        } else if (!MEDIA && GENERAL && kind == NK_MEDIUM_BEGIN) {  // surfaces only (tie_resolve): a medium's brackets and boundary copies are passed over
            n = A.media[pl].n_end + 1u;
        } else if (kind == NK_SPHERE) {
            double t;
            if (sphere_hit(A.spheres + 2 * pl, o, d, a, t_min, h.t, t)) {
                h.t = t;
                h.node = (int)n;
                h.xf = cur_xf;
                h.kp = m.x;
            }
            n++;
        } else if (GENERAL) {
            if (kind == NK_RECT_YZ || kind == NK_RECT_XZ || kind == NK_RECT_XY) {
                double t;
                if (rect_hit(A.rects + 3 * pl, (int)kind - (int)NK_RECT_YZ, o, d, t_min, h.t, t)) {
                    h.t = t;
                    h.node = (int)n;
                    h.xf = cur_xf;
                    h.kp = m.x;
                }
            } else if (kind == NK_CUBE) {
                double t;
                uint32_t side;
                if (cube_hit(A.rects + 3 * (pl >> 3), o, d, t_min, h.t, t, side)) {
                    h.t = t;
                    h.node = (int)n;
                    h.xf = cur_xf;
                    h.kp = m.x + (side << NK_BITS);
                }
            } else if (GENERAL == 2 && kind == NK_MSPHERE) {
                double t;
                if (msphere_hit(A.msph + 10 * pl, ray_time(A), o.x, o.y, o.z, d.x, d.y, d.z, a, t_min, h.t, &t)) {
                    h.t = t;
                    h.node = (int)n;
                    h.xf = cur_xf;
                    h.kp = m.x;
                }
            } else if (kind == NK_TRI) {
                double t, b1, b2;
                if (tri_hit(A.tripre + 5 * pl, o, d, t_min, h.t, t, b1, b2)) {
                    h.t = t;
                    h.node = (int)n;
                    h.xf = cur_xf;
                    h.kp = m.x;
                }
            } else if (kind == NK_XFORM_BEGIN) {  // transform.rs:153-156
                const double* Minv = A.xforms + 32 * pl;
                o = xf_point(Minv, wo);
                d = xf_dir(Minv, wd);
                inv = mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
                a = sqlen(d);
                cur_xf = (int)pl;
            } else {  // NK_XFORM_END
                o = wo;
                d = wd;
                inv = mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
                a = sqlen(d);
                cur_xf = -1;
            }
            n++;
        } else {
            n++;
        }
    }
    return h;
}

// The accel walk found two objects at the best t (see "EXACT ties" above): the reference's own walk decides.  Out of line, and
// called after the loops: rare, and the walks keep their registers.
template <int GENERAL>
#ifdef RT_TIE_NOCALL  // A/B build: the flag is kept in the loops but nothing is resolved
DEV Hit tie_resolve(const Acc& A, D3 wo, D3 wd, double t_min, double t_max, Hit h) { h.xf ^= TIE_FLAG; return h; }
#else
// (The Acc goes by reference: handing over the ten values the walk needs one by one cost the Cornell box 11 % instead of 2 %.)
__device__ __noinline__ Hit tie_resolve(const Acc& A, D3 wo, D3 wd, double t_min, double t_max, Hit) {
    Acc G = A;  // the program's tables in global memory: the accel kernels stage other things, or nothing, into LDS
    const uint32_t* tv = A.tie_view;
    const char* g = (const char*)tv - tv[6];
    G.meta = (const uint2*)(g + tv[0]);
    G.boxes = (const double2*)(g + tv[1]);
    G.spheres = (const double2*)(g + tv[2]);
    G.rects = (const double2*)(g + tv[3]);
    G.tripre = (const double2*)(g + tv[4]);
    G.xforms = (const double*)(g + tv[5]);
    G.n_nodes = tv[7];
    return traverse<GENERAL, false>(G, wo, wd, t_min, t_max);
}
#endif

// The same for kernels 5 / 6, whose scene view lives in registers (LDS pointers): handing over the Acc by reference put it into scratch memory
// for the whole kernel (pt_kernel_coop: 560 -> 976 bytes of scratch per lane, C4 -30 %).  These kernels render neither media nor the book-2
// kinds, so the walk needs nothing but the program's tables, which the tie view (one pointer) locates.
__device__ __noinline__ Hit tie_resolve_view(const uint32_t* tv, double ox, double oy, double oz, double dx, double dy, double dz, double t_min, double t_max) {
    Acc G;
    const char* g = (const char*)tv - tv[6];
    G.meta = (const uint2*)(g + tv[0]);
    G.boxes = (const double2*)(g + tv[1]);
    G.spheres = (const double2*)(g + tv[2]);
    G.rects = (const double2*)(g + tv[3]);
    G.tripre = (const double2*)(g + tv[4]);
    G.xforms = (const double*)(g + tv[5]);
    G.n_nodes = tv[7];
    G.media = nullptr;  // (never read: a scene with a ConstantMedium does not reach kernels 5 / 6)
    G.msph = nullptr;
    G.time_lds = 0u;
    return traverse<1, false>(G, mk(ox, oy, oz), mk(dx, dy, dz), t_min, t_max);
}

// ---------------------------------------------------------------- kernel 2 traversal ----
// Conservative f32 slab test of one child box of a Node2: 6 fma + 6 min/max + max3/min3 + one multiply.
//
// Claim: if some real t in [t_min, best] (t_min >= 0) puts o + t*d inside the exact f64 box B, the test passes.
// With e = 2^-24, of = fl32(o), df = fl32(d), iv = rcp32(df) clamped to |iv| <= 2^90 (v_rcp_f32: at most 1 ulp = 2 e off;
// no f64 division in the ray setup), c = fl(of*iv) (per ray, make_ray32):
//   iv = (1/d)(1+d1), |d1| <= 3.001 e   (df = d (1+a), |a| <= e; rcp32 = (1/df)(1+b), |b| <= 2 e)
//   p = fl(lo*iv - c) = (lo' - o) * iv * (1+d3),  lo' = lo - (of - o) - of*d2,  |d2|,|d3| <= e   (one fma; all operands are
//   finite because |lo|,|of| < 2^36 (flatten.cpp refuses larger scenes) -- no inf-inf, no 0*inf, no NaN).
//   |lo' - lo| <= 2.001 e |o|max, and the host stores lo <= B.min - pad, hi >= B.max + pad with pad = 4 e |o|max
//   (rounded outward), so lo' <= B.min and hi' >= B.max: per axis {p, q} = {nu (1+th), phi (1+th')} where, unless iv was
//   clamped, nu <= true slab entry, phi >= true slab exit, |th|,|th'| < 4.1 e.
//   For t as above (nu <= t <= phi on every axis, t >= 0):
//     tn = max3(min(p,q)) <= t (1 + 4.1 e)            (a non-positive nu gives a non-positive value)
//     tf = min3(max(p,q)) >= t (1 - 4.1 e) >= 0,   fl(tf * W) >= t (1 - 4.1 e)(1 - e)(1 + 16 e) >= t (1 + 10 e),  W = 1 + 2^-20
//   hence tn <= fl(tf*W);  tn <= best (1 + 4.1 e) <= r.best = best (1 + 8 e) rounded up;  r.tmin <= t_min <= t <= fl(tf*W):
//   the test passes.  Only the far side is widened; r.tmin = t_min rounded down.
//   Clamped iv (|d_axis| < 2^-90, including d_axis = +-0 where rcp32 = +-inf, and f32-denormal df): the axis constrains nothing
//   for a ray that starts inside [lo', hi'] (|p|,|q| >= 2 e |o|max * 2^90, beyond any t a path can reach: directions have a
//   component >= 1e-8, so t <= 2e8 * extent) and culls a ray that starts outside it, which is what the exact test does.
//   A NaN component gives iv = -2^90 through the clamp (fmaxf drops the NaN), as the f64 division did.
//   A negative t_min is outside this proof: the host then renders with kernel 1 (render_tiles, accel_usable).
struct Ray32 {
    float cx, cy, cz, ix, iy, iz;  // c = fl(of * iv)
    float tmin, best;  // rounded outward
    uint32_t ax, ay, az;  // WIDE nodes only: LDS byte address of the node table + 8 * axis + 32 * (iv_axis < 0), see NodeW
};
DEV float f32_down(double x) { float f = (float)x; return __builtin_fmaf(-fabsf(f), 1.1920929e-7f, f); }
DEV float f32_up(double x) { float f = (float)x; return __builtin_fmaf(fabsf(f), 1.1920929e-7f, f); }
DEV float ray32_best(double best) { return f32_up(best * (1.0 + 4.76837158203125e-7)); }
DEV float inv32(double d) {  // |iv| <= 2^90 keeps lo*iv and of*iv finite
    float f = __builtin_amdgcn_rcpf((float)d);
    const float L = 1.2379400e27f;
    return fminf(fmaxf(f, -L), L);
}
DEV Ray32 make_ray32(D3 o, D3 d, double t_min, double best) {  // d: the ray's direction
    Ray32 r;
    r.ix = inv32(d.x); r.iy = inv32(d.y); r.iz = inv32(d.z);
    r.cx = (float)o.x * r.ix; r.cy = (float)o.y * r.iy; r.cz = (float)o.z * r.iz;
    r.tmin = f32_down(t_min);
    r.best = ray32_best(best);
    return r;
}
// WIDE node table (LDS-resident scenes): the lanes pick each axis' NEAR and FAR plane by ADDRESS instead of by min / max.
// A NodeW is three 32-byte blocks [x pair, y pair, z pair, child refs] = {lo, hi, lo again}, a pair = (child 0, child 1),
// NODEW_FAR bytes apart: the block at + NODEW_FAR * s (s = 1 iff the ray runs down that axis, iv < 0) holds the near planes, the
// block NODEW_FAR further the far planes, so one address per axis (cur + r.a?) and the immediate offsets 0 / + NODEW_FAR read them,
// and the child refs sit at +24 of every block.  With lo <= hi (host) and fma monotonic in its first operand, fma(near) =
// min(fma(lo), fma(hi)) and fma(far) = max(...) value for value (a NaN c gives NaN on both sides either way), so the test decides
// exactly as box32 does with 12 fewer VALU instructions per node.  Nodes are stored in chunks of NODEW_CHUNK: chunk c holds the
// three block rows of its nodes, each row NODEW_FAR bytes; inner refs in this table are BYTE offsets (wide_ref).  NODEW_FAR is
// beyond the reach of ds_read2_b64's 8-bit offsets on purpose: merged, a near / far pair costs 8 LDS cycles instead of 2 + 2
// (MI355X_MICROARCH.md, LDS table), and the LDS was 47 % busy with the merged form.
static const uint32_t NODEW_CHUNK = 129;  // (129, not 128: a row stride of 4096 is within reach of ds_read2st64_b64)
static const uint32_t NODEW_FAR = NODEW_CHUNK * 32;  // 4128
DEV uint32_t nodew_bytes(uint32_t n_nodes) { return ((n_nodes + NODEW_CHUNK - 1) / NODEW_CHUNK) * 3u * NODEW_FAR; }
DEV void ray32_wide_addr(Ray32& r, uint32_t n2w_lds) {
    r.ax = n2w_lds + (r.ix < 0.f ? NODEW_FAR : 0u);
    r.ay = n2w_lds + 8u + (r.iy < 0.f ? NODEW_FAR : 0u);
    r.az = n2w_lds + 16u + (r.iz < 0.f ? NODEW_FAR : 0u);
}
DEV uint32_t wide_ref(uint32_t ref) {
    return (ref >> REF_TAG_SHIFT) == 0u ? (ref / NODEW_CHUNK) * (3u * NODEW_FAR) + (ref % NODEW_CHUNK) * 32u : ref;
}
// one child of a NodeW: near / far planes already selected
DEV bool box32w(float nx, float ny, float nz, float fx, float fy, float fz, const Ray32& r, float& entry) {
    const float px = __builtin_fmaf(nx, r.ix, -r.cx), qx = __builtin_fmaf(fx, r.ix, -r.cx);
    const float py = __builtin_fmaf(ny, r.iy, -r.cy), qy = __builtin_fmaf(fy, r.iy, -r.cy);
    const float pz = __builtin_fmaf(nz, r.iz, -r.cz), qz = __builtin_fmaf(fz, r.iz, -r.cz);
    // No widening factor W here (box32 multiplies tf by 1 + 2^-20): the pad is 12 e |o|max since round 3 (flatten.cpp), of which
    // 2.001 e |o|max cover lo' - lo as in the proof above box32, leaving a margin m >= 9.99 e |o|max between [lo', hi'] and the
    // exact box B on every side.  For a real t >= 0 with o + t d in B, on every axis: nu <= t - m/|d| and phi >= t + m/|d|, and
    // t |d| = the distance travelled along the axis <= 2 |o|max (|o|max bounds origins AND item coordinates: flatten.cpp's
    // origin_limit / oo), so nu (1 + th) <= t - m/|d| + 4.1 e t <= t and phi (1 + th') >= (t + m/|d|)(1 - 4.1 e) >= t because
    // 4.1 e * 2 |o|max (1 + 4.1 e) < m.  Hence tn <= t <= tf, tn <= best <= r.best, r.tmin <= t_min <= t <= tf: the test passes.
    // (non-positive nu, clamped iv, NaN: as for box32.)
    const float tn = fmaxf(fmaxf(px, py), pz);
    const float tf = fminf(fminf(qx, qy), qz);
    entry = tn;
    // tn <= min(tf, best): the compiler merges the two compares against tf and best into a v_min + one compare by itself, but
    // then re-quiets r.best (a loop-carried value from another block: v_max_f32 v, v, v) in front of it on every node; the
    // v_min_f32 is spelled out instead (IEEE mode: the non-NaN operand wins, as fminf)
    float m;
    asm("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(tf), "v"(r.best));
    return !(tn > m) && !(r.tmin > tf);
}
// box32 with each axis' near / far plane already SELECTED by the sign of the ray's inverse direction (the caller picks them with one
// v_cndmask per axis on the packed NodeQ words, serving both children): lo <= hi and fma is monotonic in its first operand, so
// fma(near) = min(fma(lo), fma(hi)) and fma(far) = max(...) value for value, and the test decides exactly as box32 does -- six
// v_min / v_max fewer per child.  (Same widening factor as box32: the grid's pad P has no spare margin to drop it.)
DEV bool box32s(float nx, float ny, float nz, float fx, float fy, float fz, const Ray32& r, float& entry) {
    const float px = __builtin_fmaf(nx, r.ix, -r.cx), qx = __builtin_fmaf(fx, r.ix, -r.cx);
    const float py = __builtin_fmaf(ny, r.iy, -r.cy), qy = __builtin_fmaf(fy, r.iy, -r.cy);
    const float pz = __builtin_fmaf(nz, r.iz, -r.cz), qz = __builtin_fmaf(fz, r.iz, -r.cz);
    const float tn = fmaxf(fmaxf(px, py), pz);
    const float tf = fminf(fminf(qx, qy), qz) * (1.0f + 9.5367431640625e-7f);
    entry = tn;
    return !(tn > tf) && !(tn > r.best) && !(r.tmin > tf);
}
// the per-axis lane masks "the ray runs down this axis" of a wave, for box32s' callers
struct SignMasks {
    uint64_t x, y, z;
};
DEV SignMasks sign_masks(const Ray32& r) {
    SignMasks m;
    m.x = __ballot(r.ix < 0.f);
    m.y = __ballot(r.iy < 0.f);
    m.z = __ballot(r.iz < 0.f);
    return m;
}
DEV uint32_t sel32(uint32_t if_clear, uint32_t if_set, uint64_t lane_mask) {  // per lane: lane_mask[lane] ? if_set : if_clear
    uint32_t v;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(v) : "v"(if_clear), "v"(if_set), "s"(lane_mask));
    return v;
}
DEV bool box32(float lox, float loy, float loz, float hix, float hiy, float hiz, const Ray32& r, float& entry) {
    float px = __builtin_fmaf(lox, r.ix, -r.cx), qx = __builtin_fmaf(hix, r.ix, -r.cx);
    float py = __builtin_fmaf(loy, r.iy, -r.cy), qy = __builtin_fmaf(hiy, r.iy, -r.cy);
    float pz = __builtin_fmaf(loz, r.iz, -r.cz), qz = __builtin_fmaf(hiz, r.iz, -r.cz);
    // max(tn, tmin) <= min(tf, best) written as three compares (tmin <= best always): the loop-invariant tmin / best then need no
    // per-node canonicalising v_max (the compiler re-quiets values that come from another basic block): 2 of 46 VALU per node.
    // NaN rays (degenerate cameras, Q2) must keep passing every box -- the reference accepts NaN hits -- hence the negated
    // compares: fminf / fmaxf drop a NaN axis as before, and an all-NaN tn or tf fails no "greater than".
    float tn = fmaxf(fmaxf(fminf(px, qx), fminf(py, qy)), fminf(pz, qz));
    float tf = fminf(fminf(fmaxf(px, qx), fmaxf(py, qy)), fmaxf(pz, qz)) * (1.0f + 9.5367431640625e-7f);
    entry = tn;
    return !(tn > tf) && !(tn > r.best) && !(r.tmin > tf);
}

// Closest hit through the accel (common/flat.h): near-child-first BVH2 descent with a per-lane stack in LDS,
// "while-while" form (all lanes descend inner nodes until each holds a leaf, then all test leaves).
// Primitive tests, ranges and the tie rule are the reference's (see traverse<> above); only the order in which
// primitives are met differs, which cannot change the result.
// DEFER (cooperative kernel 5): an instance item is not entered; its index is recorded in *pend and its object-space BVH is
// walked later by whichever wave serves the workgroup's request ring (coop_serve), starting from this walk's result.
// TOP: the scene lives in L2/HBM and the shallowest nodes are cached in LDS (A.n2_top); false for LDS-resident scenes, which then
// carry no test for it in the node loop.
// WIDE: the node table is the LDS-resident NodeW form (A.n2w_lds; implies !TOP).
// LIMIT: only items whose reference-order index is below order_limit take part (traverse2_media: "the closest surface the
// reference has seen before it visits this medium").
// TRACK (traverse2_media): beside the closest hit, the walk keeps for up to two media the smallest t among the items the reference
// visits BEFORE the medium (program index below lim[j]) -- but only as far as the medium's exit (beyond it the value is of no use):
// candidates are then taken up to  bound = max(best t, min(T[j], exit[j]))  instead of the best t, and the boxes are culled against
// that.  The root a sphere test returns does not depend on how far the range reaches (it is the smallest admissible one), so the
// closest hit is what the plain walk finds, and T[j] is what a walk restricted to those items would find (up to exit[j]).
struct MediaTrack {
    uint32_t lim[2];  // program index of the medium's BEGIN node; 0: slot unused
    double exitt[2];  // the medium's exit t on this ray (rec2.t)
    double T[2];      // out: min t of the items below lim[j] (+inf: none up to exit[j])
};
DEV double track_bound(const MediaTrack& K, double best) {
    return fmax(best, fmax(K.lim[0] != 0u ? fmin(K.T[0], K.exitt[0]) : 0., K.lim[1] != 0u ? fmin(K.T[1], K.exitt[1]) : 0.));
}
// ENTER: instance items may be entered in the lane (always, unless DEFER; with DEFER only in the MIXED variants of kernels 5 / 6,
// for the NK_INSTANCE_INLINE items of scenes that have any: compiling the enter path into their world-space walk costs C4 6 %).
// RESOLVE: an exact tie (TIE_FLAG) is settled before returning; false: the caller does it (traverse2_media: one call site for its two walks).
// TIEDEF: a DEFER walk (kernels 5 / 6) notes exact ties among the world-level items as well; its caller settles them (RESOLVE is false there).
// SLICE (round 5; the MEDIA variants of kernel 2): the walk can be SUSPENDED.  A wave stays in this loop until its longest ray is through:
// on the book-2 final scene 7.9 descend-and-leaf rounds per walk where a lane needs 1.8 (phase statistics) -- a few rays cross
// the 1000-sphere cluster or skim the ground boxes while the rest of the wave waits.  With SLICE the lanes that are still walking leave the
// loop together as soon as fewer than `slice_th` of them are left (a wave-uniform test once per round); what a walk needs to go on -- the
// node to continue at, the stack pointer (the stack itself stays in the lane's LDS column), the instance it is inside of; the best hit so
// far goes back to the caller as the result -- is handed back in *ws, the finished lanes shade and start their next segment, and the
// next call continues the suspended walks beside the fresh ones.  The walk itself is unchanged: same nodes, same order, same candidates,
// hence the same hit.  (On the LDS-resident scenes the rounds' tail is short -- 3.5 rounds against 1.5 on the headline scene -- and the
// slicing only costs registers: measured -1.4 % there, -21 % on the Cornell box; the switch RT_SLICE_TH compiles it into the MEDIA variants
// alone, and it is OFF in the product: no gain there either, see RT_SLICE_TH.)
struct WalkState {
    uint32_t cur;  // node / leaf to continue at; REF_DONE: no walk in progress
    uint32_t sp;   // stack offset in words (WIDE: the LDS byte address)
    int cur_xf;    // the instance the walk is inside of (xform index), -1: world space
};
template <int GENERAL, bool DEFER = false, bool TOP = true, bool WIDE = false, class PEND = uint32_t, bool LIMIT = false, bool ENTER = !DEFER, bool TRACK = false, bool RESOLVE = true,
          bool TIEDEF = false, bool SLICE = false>
DEV Hit traverse2(const Acc& A, uint32_t* stk, const int stride, D3 wo, D3 wd, double t_min, double t_max, PEND* pend = nullptr, uint32_t order_limit = 0xFFFFFFFFu,
                  MediaTrack* track = nullptr, WalkState* ws = nullptr, const Hit* h_in = nullptr, int slice_th = 0) {
    D3 o = wo, d = wd;
    Hit h;
    h.t = t_max;
    h.node = -1;
    h.xf = -1;
    h.kp = 0;
    int cur_xf = -1;
    const bool resume = SLICE && ws->cur != REF_DONE;
    if (resume) {  // a suspended walk goes on: its best hit so far, and the ray in the space of the instance it is inside of
        h = *h_in;
        cur_xf = ws->cur_xf;
        if (GENERAL && cur_xf >= 0) {
            const double* Minv = A.xforms + 32 * cur_xf;
            o = xf_point(Minv, wo);
            d = xf_dir(Minv, wd);
        }
    }
    double a = sqlen(d);
    // exact ties are noted here unless the walk sees only a part of the scene (LIMIT: only t is used); a walk that defers instances
    // (kernels 5 / 6) notes them in its TIE variants (TIEDEF)
    constexpr bool TIE = TIE_RULE && GENERAL != 0 && (!DEFER || TIEDEF) && !LIMIT && !(TRACK && RT_TIE_NOTRACK);
    Ray32 r = make_ray32(o, d, t_min, (SLICE && TRACK) ? track_bound(*track, h.t) : (SLICE ? h.t : t_max));
    float best_all32 = SLICE ? ray32_best(h.t) : r.best;  // TRACK: the best hit's own (outward-rounded) t, beside r.best = the track bound
    if (WIDE) ray32_wide_addr(r, A.n2w_lds);
    int sp = 0;  // stack offset in words (a multiple of stride): avoids an integer multiply per push/pop
    // WIDE: the stack pointer is the LDS byte address itself (one add per push / pop instead of shift-add + add)
    const uint32_t spw0 = WIDE ? (uint32_t)(uintptr_t)(AS_L uint32_t*)stk : 0u;
    const uint32_t spw_step = 4u * (uint32_t)stride;
    uint32_t spw = spw0;
    uint32_t cur = WIDE ? wide_ref(A.root2) : A.root2;
    if (resume) {
        cur = ws->cur;
        if (WIDE) spw = ws->sp;
        else sp = (int)ws->sp;
    }
    bool suspended = false;
    for (;;) {
        if (SLICE && (int)__popcll(__ballot(1)) < slice_th) {  // (the lanes whose walks are over have left the loop: the ballot counts the walkers)
            suspended = true;
            break;
        }
        while ((cur >> REF_TAG_SHIFT) == 0u) {  // inner node: test both children
            PH_EV(10);
#ifdef RTAMD_PHASE_STATS
            if (GENERAL && cur_xf >= 0) PH_EV(15);  // ... of which inside an instance's object-space BVH
#endif
            if (WIDE) {
                const uint32_t axx = cur + r.ax, ayy = cur + r.ay, azz = cur + r.az;
                // seven 8-byte LDS reads, spelled out: left to itself the compiler pairs them into ds_read2_b64 / ds_read2st64_b64
                // (8 LDS cycles per pair instead of 2 + 2) wherever two offsets are in reach of each other
                f32x2 nx, fx, ny, fy, nz, fz, cc;
                asm volatile(
                    "ds_read_b64 %0, %7\n\t"
                    "ds_read_b64 %1, %7 offset:%10\n\t"
                    "ds_read_b64 %2, %8\n\t"
                    "ds_read_b64 %3, %8 offset:%10\n\t"
                    "ds_read_b64 %4, %9\n\t"
                    "ds_read_b64 %5, %9 offset:%10\n\t"
                    "ds_read_b64 %6, %7 offset:24\n\t"
                    "s_waitcnt lgkmcnt(0)"
                    : "=&v"(nx), "=&v"(fx), "=&v"(ny), "=&v"(fy), "=&v"(nz), "=&v"(fz), "=&v"(cc)
                    : "v"(axx), "v"(ayy), "v"(azz), "n"(NODEW_FAR)
                    : "memory");
                float e0, e1;
                const bool h0 = box32w(nx.x, ny.x, nz.x, fx.x, fy.x, fz.x, r, e0);
                const bool h1 = box32w(nx.y, ny.y, nz.y, fx.y, fy.y, fz.y, r, e1);
                const uint32_t c0 = __float_as_uint(cc.x), c1 = __float_as_uint(cc.y);
                if (h0 && h1) {
                    const bool swap = e1 < e0;
                    *(AS_L uint32_t*)(uintptr_t)spw = swap ? c0 : c1;
                    spw += spw_step;
                    cur = swap ? c1 : c0;
                } else if (h0) {
                    cur = c0;
                } else if (h1) {
                    cur = c1;
                } else if (spw != spw0) {
                    spw -= spw_step;
                    cur = *(const AS_L uint32_t*)(uintptr_t)spw;
                } else {
                    cur = REF_DONE;
                }
                continue;
            }
            f32x4 q0, q1, q2;  // (lox0,lox1,loy0,loy1) (loz0,loz1,hix0,hix1) (hiy0,hiy1,hiz0,hiz1) (c0,c1,-,-)
            f32x2 q3;          // (only the two child refs of the fourth quad: an 8-byte read)
            f32x2 q3m = {0.f, 0.f};  // LIMIT / TRACK: the smallest program index in either child's subtree (Node2.pad, accel.cpp)
            if (TOP && cur < A.n2_top_count) {  // the shallowest levels are cached in LDS when the scene lives in L2/HBM
                const AS_L f32x4* p = A.n2_top + NODE2_F4 * cur;
                q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = *(const AS_L f32x2*)(p + 3);
                if (LIMIT || TRACK) q3m = ((const AS_L f32x2*)(p + 3))[1];
            } else {
                const f32x4* p = (const f32x4*)A.n2 + NODE2_F4 * cur;
                q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = *(const f32x2*)(p + 3);
                if (LIMIT || TRACK) q3m = ((const f32x2*)(p + 3))[1];
            }
            float e0, e1;
            bool h0 = box32(q0.x, q0.z, q1.x, q1.z, q2.x, q2.z, r, e0);
            bool h1 = box32(q0.y, q0.w, q1.y, q1.w, q2.y, q2.w, r, e1);
            if (LIMIT) {  // a subtree without an item below the limit holds nothing for this walk
                h0 = h0 && __float_as_uint(q3m.x) < order_limit;
                h1 = h1 && __float_as_uint(q3m.y) < order_limit;
            }
            if (TRACK) {  // ... and in a tracked walk such a subtree is only of interest as far as the best hit (r.best reaches to the track bound)
                const uint32_t lim = max(track->lim[0], track->lim[1]);
                h0 = h0 && (__float_as_uint(q3m.x) < lim || !(e0 > best_all32));
                h1 = h1 && (__float_as_uint(q3m.y) < lim || !(e1 > best_all32));
            }
            uint32_t c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
            if (h0 && h1) {
                bool swap = e1 < e0;
                uint32_t nearc = swap ? c1 : c0, farc = swap ? c0 : c1;
                stk[sp] = farc;
                sp += stride;
                cur = nearc;
            } else if (h0) {
                cur = c0;
            } else if (h1) {
                cur = c1;
            } else if (sp > 0) {
                sp -= stride;
                cur = stk[sp];
            } else {
                cur = REF_DONE;
            }
        }
        if (cur == REF_DONE) break;
        PH_BEGIN(ph_leaf0);
        if ((cur >> REF_TAG_SHIFT) == 1u) {  // leaf: test its items with the reference's f64 routines
            uint32_t first = cur & REF_LEAF_FIRST_MASK, cnt = ((cur >> REF_LEAF_COUNT_SHIFT) & 7u) + 1u;
            uint32_t enter = REF_DONE;
            for (uint32_t i = 0; i < cnt; i++) {
                PH_EV(11);
                uint2 it = A.items2[first + i];
                uint32_t kind = it.x & NK_MASK, pl = it.x >> NK_BITS;
                double t = 0.;
                bool got = false;
                uint32_t cube_side = 0u;  // NK_CUBE: the winning side goes into the hit's kp
                // how far a candidate may lie (TIE: a little beyond the best hit, to note near ties -- see TIE_W)
                const double t_far = TRACK ? track_bound(*track, TIE ? h.t * TIE_W : h.t) : (TIE ? h.t * TIE_W : h.t);
                if (LIMIT && it.y >= order_limit) {
                    // visited by the reference after the medium in question (an instance's subtree is contiguous in the program and
                    // media are world-level, so an instance lies wholly before or wholly after it: its item's order decides)
                } else if (kind == NK_SPHERE) {
                    got = sphere_hit(A.spheres + 2 * pl, o, d, a, t_min, t_far, t);
                } else if (GENERAL) {
                    if (kind == NK_RECT_YZ || kind == NK_RECT_XZ || kind == NK_RECT_XY) {
                        got = rect_hit(A.rects + 3 * pl, (int)kind - (int)NK_RECT_YZ, o, d, t_min, t_far, t);
                    } else if (GENERAL == 2 && kind == NK_MSPHERE) {
                        got = msphere_hit(A.msph + 10 * pl, ray_time(A), o.x, o.y, o.z, d.x, d.y, d.z, a, t_min, t_far, &t);
                    } else if (kind == NK_CUBE) {
                        PH_EV(12);
                        got = cube_hit(A.rects + 3 * (pl >> 3), o, d, t_min, t_far, t, cube_side);
                    } else if (kind == NK_TRI) {
                        double b1, b2;
                        got = tri_hit(A.tripre2 + 5 * (first + i), o, d, t_min, t_far, t, b1, b2);
                    } else if (DEFER && kind == NK_INSTANCE) {  // deferred (at most 32 / 64 instances, checked on the host)
                        *pend |= (PEND)1 << pl;
                    } else if (ENTER) {  // NK_INSTANCE (or NK_INSTANCE_INLINE: an instance kernels 5 / 6 cannot defer): descend into its object-space BVH after the remaining items
                        enter = pl;
                    }
                }
                if (TRACK) {
                    if (got) {
                        if (it.y < track->lim[0] && t < track->T[0]) track->T[0] = t;
                        if (it.y < track->lim[1] && t < track->T[1]) track->T[1] = t;
                        if (t < h.t || (t == h.t && (int)it.y > h.node) || !(t == t)) {  // (a candidate may lie beyond the best hit here)
                            const bool tied = TIE && h.node >= 0 && t * TIE_W >= h.t;  // (near) tie with the hit it replaces
                            h.t = t;
                            h.node = (int)it.y;
                            h.xf = tied ? (cur_xf ^ TIE_FLAG) : cur_xf;
                            h.kp = it.x + (cube_side << NK_BITS);
                            best_all32 = ray32_best(t);
                        } else if (TIE && !(t > h.t * TIE_W) && (int)it.y < h.node && !tie_flagged(h.xf)) {  // the candidate is the earlier party of a (near) tie
                            h.xf ^= TIE_FLAG;
                        }
                        r.best = ray32_best(track_bound(*track, h.t));
                    }
                } else if (got) {
                    // candidates satisfy t <= h.t (TIE: h.t * TIE_W); an exact tie goes to the later one in reference order; (near) ties are noted: tie_resolve
                    if (TIE ? !(t < h.t || (t == h.t && (int)it.y > h.node) || !(t == t)) : !(t < h.t || (int)it.y > h.node || !(t == t))) {
                        if (TIE && (int)it.y < h.node && !tie_flagged(h.xf)) h.xf ^= TIE_FLAG;  // t within the tie margin of h.t and the candidate is the earlier party
                        continue;
                    }
                    const bool tied = TIE && h.node >= 0 && t * TIE_W >= h.t;
                    h.t = t;
                    h.node = (int)it.y;
                    h.xf = tied ? (cur_xf ^ TIE_FLAG) : cur_xf;
                    h.kp = it.x + (cube_side << NK_BITS);
                    r.best = ray32_best(t);
                }
            }
            if (GENERAL && ENTER && enter != REF_DONE) {  // Transform::hit, transform.rs:153-156
                uint2 in = A.inst2[enter];
                const double* Minv = A.xforms + 32 * in.x;
                o = xf_point(Minv, wo);
                d = xf_dir(Minv, wd);
                a = sqlen(d);
                cur_xf = (int)in.x;
                r = make_ray32(o, d, t_min, TRACK ? track_bound(*track, h.t) : h.t);
                if (WIDE) {
                    ray32_wide_addr(r, A.n2w_lds);
                    *(AS_L uint32_t*)(uintptr_t)spw = REF_RESTORE;
                    spw += spw_step;
                } else {
                    stk[sp] = REF_RESTORE;
                    sp += stride;
                }
                cur = WIDE ? wide_ref(in.y) : in.y;
                PH_END(9, ph_leaf0);
                continue;
            }
        } else {  // REF_RESTORE: leave the Transform
            o = wo;
            d = wd;
            a = sqlen(d);
            cur_xf = -1;
            r = make_ray32(o, d, t_min, TRACK ? track_bound(*track, h.t) : h.t);
            if (WIDE) ray32_wide_addr(r, A.n2w_lds);
        }
        if (WIDE) {
            if (spw != spw0) {
                spw -= spw_step;
                cur = *(const AS_L uint32_t*)(uintptr_t)spw;
            } else {
                cur = REF_DONE;
            }
        } else if (sp > 0) {
            sp -= stride;
            cur = stk[sp];
        } else {
            cur = REF_DONE;
        }
        PH_END(9, ph_leaf0);
    }
    if (SLICE) {
        ws->cur = suspended ? cur : REF_DONE;
        ws->sp = WIDE ? spw : (uint32_t)sp;
        ws->cur_xf = cur_xf;
        if (suspended) return h;  // (its flag, if any, travels in h.xf)
    }
    // the last exact tie noted, if its later party is still the hit: would the reference have visited that object at all?
    if (TIE && RESOLVE && tie_flagged(h.xf)) h = tie_resolve<GENERAL>(A, wo, wd, t_min, t_max, h);
    return h;
}

// World::hit through the accel for a scene with ConstantMedium objects (objects/medium.rs:25-53).  A medium's hit() makes its
// one random draw only if the ray is inside its boundary somewhere within [t_min, closest hit SO FAR], and "so far" means: among
// what the reference has visited BEFORE the medium.  The accel visits things in another order, so the media are handled apart, in
// the reference's order (A.media is in program order):
//   S  = closest surface of all (one accel walk, the box tests merely cull);
//   for each medium M:  S_M = closest surface among those the reference visits before M -- for the first two media the ray crosses
//       the accel walk itself keeps it (TRACK: min t over items with a smaller program index, as far as the medium's exit); for
//       further ones S itself when S comes before M in the program, otherwise another accel walk restricted to those items (LIMIT);
//       t_max = min(S_M.t, t of the latest medium hit accepted so far);   rec1 / rec2 = the two boundary queries, walked over the
//       medium's own two copies of the boundary's subtree in the reference-order program (exact f64, any sign of t);
//       then medium.rs:28-50 literally: clip, at most one draw, accept t = rec1.t + hit_distance / |d| (it is <= t_max).
//   result = the later-accepted of S and the last medium hit: smaller t, an exact tie to the larger program index.
// Equivalence with the recursion: a surface is accepted by the reference iff its t <= closest so far, whatever came before, so the
// final surface candidate is S; a medium's hit depends only on t_max at its visit = what precedes it, which is what S_M and the
// earlier media hits give; the reference's own box tests cull a medium exactly when its clipped interval is empty (no draw
// either way), up to rays that graze a reference box within f64 rounding (the measure-zero caveat of kernel 2).
// rec1 = boundary.hit(r, -inf, +inf); rec2 = boundary.hit(r, rec1.t + 0.0001, +inf) (medium.rs:26-27): false unless both exist.
// Any other boundary: the reference-order walk over the medium's own copies of the boundary subtree, out of line (two inlined walks in
// the middle of traverse2_media cost the MEDIA kernels registers whether a scene has such a boundary or not: C5 as named +6.5 %).
template <int GENERAL>
__device__ __noinline__ Hit walk_program(const Acc& A, D3 o, D3 d, double t_min, double t_max, uint32_t n0, uint32_t n1) {
    return traverse<GENERAL, false>(A, o, d, t_min, t_max, nullptr, n0, n1);
}
// A boundary that is one world-space sphere (MediumDev::boundary_kp) is asked directly -- the walk over its one-node subtree would make
// exactly these two Sphere::hit calls, after three f64 divisions for box tests that never come.
template <int GENERAL>
DEV bool medium_boundary(const Acc& A, const MediumDev& M, D3 o, D3 d, double& t_a, double& t_b) {
    if (M.boundary_kp != 0u) {
        // Both Sphere::hit calls (sphere.rs:24-43) see the same ray and sphere, so oc, half_b, c, the discriminant and its root are the
        // same numbers in both: computed once; each call's root selection is then replayed on the two roots with its own [t_min, t_max].
        const double2* s = A.spheres + 2 * (M.boundary_kp >> NK_BITS);
        const double2 c0 = s[0], c1 = s[1];
        const D3 oc = mk(o.x - c0.x, o.y - c0.y, o.z - c1.x);
        const double radius = c1.y, a = sqlen(d);
        const double half_b = dot(oc, d);
        const double c = sqlen(oc) - radius * radius;
        const double disc = half_b * half_b - a * c;
        if (disc < 0.) return false;
        const double sq = sqrt(disc);
        const double r1 = (-half_b - sq) / a, r2 = (-half_b + sq) / a;
        // rec1 = hit(r, -inf, +inf): the near root unless it is NaN, then the far one
        double ta = r1;
        if (!(ta >= -INFINITY && ta <= INFINITY)) ta = r2;
        if (!(ta >= -INFINITY && ta <= INFINITY)) return false;
        // rec2 = hit(r, rec1.t + 0.0001, +inf)
        const double lo = ta + 0.0001;
        double tb = r1;
        if (!(tb >= lo && tb <= INFINITY)) tb = r2;
        if (!(tb >= lo && tb <= INFINITY)) return false;
        t_a = ta;
        t_b = tb;
        return true;
    }
    const Hit r1h = walk_program<GENERAL>(A, o, d, -INFINITY, INFINITY, M.n_begin + 1u, M.n_mid);
    if (r1h.node < 0) return false;
    const Hit r2h = walk_program<GENERAL>(A, o, d, r1h.t + 0.0001, INFINITY, M.n_mid + 1u, M.n_end);
    if (r2h.node < 0) return false;
    t_a = r1h.t;
    t_b = r2h.t;
    return true;
}
// SLICE: the accel walk of step 2 may come back SUSPENDED (traverse2's SLICE: ws->cur != REF_DONE; the result is then its best hit so far and
// t_save[] the two tracked minima): the caller skips the shading and calls again next iteration; step 1 is then repeated (it draws no random
// number and its results depend on the ray alone), the walk resumes, and steps 3 run once, when the walk is over.
template <int GENERAL, bool TOP, bool WIDE, bool SLICE = false>
DEV Hit traverse2_media(const Acc& A, uint32_t n_media, uint32_t* stk, const int stride, D3 o, D3 d, double t_min, Rng& rng, WalkState* ws = nullptr, const Hit* h_in = nullptr,
                        double* t_save = nullptr, int slice_th = 0) {
    // 1. the boundary queries of the media (no random number is drawn here); the first two media the ray crosses are TRACKED by the
    //    accel walk (MediaTrack), so that one walk yields the closest surface S and, for each of the two, the closest surface the
    //    reference visits before it.  (A second, order-restricted walk per medium -- the first version -- cost more than the
    //    rest of the segment: 451 instead of 935 Msamples/s without it on the reduced book-2 final scene, where every ray is inside a fog.)
    MediaTrack K;
    K.lim[0] = K.lim[1] = 0u;
    K.exitt[0] = K.exitt[1] = 0.;
    K.T[0] = K.T[1] = INFINITY;
    uint32_t tk0 = 0xFFFFFFFFu, tk1 = 0xFFFFFFFFu, k_rest = n_media;  // the tracked media; from k_rest on: not examined yet
    double ta0 = 0., tb0 = 0., ta1 = 0., tb1 = 0.;
    for (uint32_t k = 0; k < n_media; k++) {
        const MediumDev M = A.media[k];
        double qa, qb;
        if (!medium_boundary<GENERAL>(A, M, o, d, qa, qb)) continue;
        if (tk0 == 0xFFFFFFFFu) {
            tk0 = k; ta0 = qa; tb0 = qb;
            K.lim[0] = M.n_begin; K.exitt[0] = qb;
        } else {
            tk1 = k; ta1 = qa; tb1 = qb;
            K.lim[1] = M.n_begin; K.exitt[1] = qb;
            k_rest = k + 1u;
            break;
        }
    }
    // 2. the accel walk
    if (SLICE && ws->cur != REF_DONE) {  // a suspended walk goes on: what it had tracked so far
        K.T[0] = t_save[0];
        K.T[1] = t_save[1];
    }
    Hit S = (tk0 != 0xFFFFFFFFu)
                ? traverse2<GENERAL, false, TOP, WIDE, uint32_t, false, true, true, false, false, SLICE>(A, stk, stride, o, d, t_min, INFINITY, nullptr, 0xFFFFFFFFu, &K, ws, h_in, slice_th)
                : traverse2<GENERAL, false, TOP, WIDE, uint32_t, false, true, false, false, false, SLICE>(A, stk, stride, o, d, t_min, INFINITY, nullptr, 0xFFFFFFFFu, nullptr, ws, h_in, slice_th);
    if (SLICE && ws->cur != REF_DONE) {  // suspended: the rest happens when the walk is over
        t_save[0] = K.T[0];
        t_save[1] = K.T[1];
        return S;
    }
    if (TIE_RULE && GENERAL && tie_flagged(S.xf)) S = tie_resolve<GENERAL>(A, o, d, t_min, INFINITY, S);  // two surfaces share the best t ("EXACT ties" above)
    // 3. the media in the reference's order
    Hit best = S;
    double t_med = INFINITY;  // t of the latest accepted medium hit
    const double ray_length = sqrt(sqlen(d));
    for (uint32_t k = 0; k < n_media; k++) {
        double t_a, t_b, t_max = t_med;
        MediumDev M;
        if (k == tk0 || k == tk1) {
            M = A.media[k];
            t_a = (k == tk0) ? ta0 : ta1;
            t_b = (k == tk0) ? tb0 : tb1;
            t_max = fmin(t_max, (k == tk0) ? K.T[0] : K.T[1]);
        } else if (k >= k_rest) {  // a third, fourth ... crossed medium: its own queries, and a restricted walk where needed
            M = A.media[k];
            if (!medium_boundary<GENERAL>(A, M, o, d, t_a, t_b)) continue;
            if (S.node >= 0) {
                if ((uint32_t)S.node < M.n_begin) {
                    t_max = fmin(t_max, S.t);
                } else if (S.t < t_b) {  // (S comes after M; a surface before M is no closer than S: if even S lies beyond the exit nothing clips)
                    const Hit SM = traverse2<GENERAL, false, TOP, WIDE, uint32_t, true>(A, stk, stride, o, d, t_min, INFINITY, nullptr, M.n_begin);
                    if (SM.node >= 0) t_max = fmin(t_max, SM.t);
                }
            }
        } else {
            continue;  // examined in step 1: not crossed
        }
        double r1 = fmax(t_a, t_min);
        const double r2 = fmin(t_b, t_max);
        if (r1 >= r2) continue;
        r1 = fmax(r1, 0.);
        const double distance_inside_boundary = (r2 - r1) * ray_length;
        const double hit_distance = M.neg_inv_density * det_ln(rng.gen_f64());  // the only draw, medium.rs:37-38
        if (hit_distance > distance_inside_boundary) continue;
        t_med = r1 + hit_distance / ray_length;
        Hit hm;
        hm.t = t_med;
        hm.node = (int)M.n_end;
        hm.xf = -1;
        hm.kp = NK_MEDIUM_END | (k << NK_BITS);
        // the reference accepts it (t <= its t_max); against S it wins when closer, or on an exact tie when it is visited later
        if (S.node < 0 || hm.t < S.t || (hm.t == S.t && hm.node > S.node)) best = hm;
        else best = S;
    }
    return best;
}

struct Rec {  // HitRecord, hit.rs:7-14
    D3 p, normal;
    bool front_face;
    double u, v;
    int mat;
};

// Sign of sin(a) without evaluating it: with k = floor(a/pi), sin(a) < 0 iff k is odd.  r = a*(1/pi) carries an
// absolute error < 2.3e-16*|r| < 6e-11 for |r| < 2^18, so k is the true floor whenever frac(r) is farther than
// 1e-9 from 0 and 1; otherwise (and for huge |a|) the caller falls back to the real sin().  Returns 0 (+), 1 (-), 2 (?).
DEV int sin_sign_fast(double a) {
    const double INV_PI = 0.318309886183790671537767526745028724;
    double r = a * INV_PI;
    double kf = floor(r);
    double f = r - kf;
    if (!(fabs(r) < 262144.0) || f < 1e-9 || f > 1.0 - 1e-9) return 2;
    return ((int)kf) & 1;
}
// This cold path is kept OUT OF LINE: the hot loop is bound by instruction issue/fetch, and three inlined sin() argument
// reductions in the middle of it cost ~1 % although they almost never run.
__device__ __attribute__((noinline)) bool checker_sines_negative(D3 p) {  // the literal test of CheckerTexture, material.rs:62-69
    double sines = sin(10. * p.x) * sin(10. * p.y) * sin(10. * p.z);
    return sines < 0.;
}
template <int GENERAL = 1>
DEV D3 tex_color(const Acc& A, int tex, const Rec& rec) {  // material.rs:52-84
    const TexDev* t = &A.texs[tex];
    int type = t->type;
    if (type == 1) {  // CheckerTexture: .0 when sines < 0 ; only the SIGN of sin(10x)sin(10y)sin(10z) is used
        D3 p = rec.p;
        int s0 = sin_sign_fast(10. * p.x), s1 = sin_sign_fast(10. * p.y), s2 = sin_sign_fast(10. * p.z);
        bool negative;
        if ((s0 | s1 | s2) & 2) {  // within 1e-9 of a zero of some factor (or huge argument): evaluate literally
            negative = checker_sines_negative(p);
        } else {
            negative = ((s0 ^ s1 ^ s2) & 1) != 0;  // |factors| > 3e-9, so the product cannot underflow to zero
        }
        t = &A.texs[negative ? t->t0 : t->t1];
        type = 0;
    }
    if (type == 0) return mk(t->color[0], t->color[1], t->color[2]);
    if (GENERAL == 2 && type == 3) {  // D9: noise_texture::value = color(1, 1, 1) * 0.5 * (1 + sin(scale p.z + 10 turb(p)))
        const double m = noise_marble(A.texels + t->texel_off, t->color[0], rec.p.x, rec.p.y, rec.p.z);
        return mk(1. * m, 1. * m, 1. * m);
    }
    // ImageTexture: nearest texel, v flipped; x == w at u == 1 is clamped (Q11)
    double u = fmin(fmax(rec.u, 0.), 1.), v = 1. - fmin(fmax(rec.v, 0.), 1.);
    int x = (int)floor((double)t->w * u), y = (int)floor((double)t->h * v);
    if (x > t->w - 1) x = t->w - 1;
    if (y > t->h - 1) y = t->h - 1;
    const uint8_t* px = A.texels + t->texel_off + ((size_t)y * t->w + x) * 3;
    return mk(px[0] / 255., px[1] / 255., px[2] / 255.);
}

DEV void sphere_uv(D3 outward, double& u, double& v) {  // get_uv, sphere.rs:16-20 (inline: a call here costs Cornell 4 %)
    const double PI = 3.14159265358979323846264338327950288, FRAC_1_PI = 0.318309886183790671537767526745028724;
    double theta = acos(-outward.y);
    double phi = atan2(-outward.z, outward.x) + PI;
    u = phi * FRAC_1_PI * 0.5;
    v = theta * FRAC_1_PI;
}
// Build the HitRecord of the winning leaf only (the reference builds one per candidate).
template <int GENERAL>
DEV Rec materialize(const Acc& A, const Hit& h, D3 wo, D3 wd, int* err) {
    Rec rec;
    uint32_t kind = h.kp & NK_MASK, pl = h.kp >> NK_BITS;
    D3 o = wo, d = wd;
    if (GENERAL && h.xf >= 0) {
        const double* Minv = A.xforms + 32 * h.xf;
        o = xf_point(Minv, wo);
        d = xf_dir(Minv, wd);
    }
    D3 outward;
    rec.u = 0.;
    rec.v = 0.;
    bool want_uv = false;
    if (kind == NK_SPHERE) {  // sphere.rs:45-53
        double2 c0 = A.spheres[2 * pl], c1 = A.spheres[2 * pl + 1];
        rec.mat = A.sphere_mat[pl];
        D3 p = add(o, muls(d, h.t));
        outward = divs(sub(p, mk(c0.x, c0.y, c1.x)), c1.y);
        want_uv = A.texs[A.mats[rec.mat].tex].type == 2;
        if (want_uv) sphere_uv(outward, rec.u, rec.v);  // get_uv, sphere.rs:16-20 (only an ImageTexture reads it)
    } else if (GENERAL == 2 && kind == NK_MSPHERE) {  // D9: outward_normal = (p - center(r.time)) / radius, get_uv as a sphere
        const double* q = A.msph + 10 * pl;
        rec.mat = (int)q[9];
        D3 p = add(o, muls(d, h.t));
        outward = divs(sub(p, msphere_center(q, ray_time(A))), q[8]);
        want_uv = A.texs[A.mats[rec.mat].tex].type == 2;
        if (want_uv) sphere_uv(outward, rec.u, rec.v);
    } else if (GENERAL && kind == NK_MEDIUM_END) {  // ConstantMedium: arbitrary normal (1,0,0), uv (0,0), phase function (medium.rs:43-49)
        rec.mat = A.media[pl].mat;
        outward = mk(1., 0., 0.);
    } else if (GENERAL && kind != NK_TRI) {  // rectangles, and the winning side of a Cube (cube.rs:17-54: the side's rectangle from the corners)
        int axis = (int)kind - (int)NK_RECT_YZ;
        double2 r0, r1;
        if (kind == NK_CUBE) {
            const uint32_t side = pl & 7u;
            pl >>= 3;
            const double2 c0 = A.rects[3 * pl], c1 = A.rects[3 * pl + 1], c2 = A.rects[3 * pl + 2];  // (min.x, min.y) (min.z, max.x) (max.y, max.z)
            axis = side < 2u ? 2 : (side < 4u ? 1 : 0);
            if (axis == 2) { r0 = c0; r1.x = c1.y; r1.y = c2.x; }                                   // xy0 = min.xy(), xy1 = max.xy()
            else if (axis == 1) { r0.x = c0.x; r0.y = c1.x; r1.x = c1.y; r1.y = c2.y; }             // xz0 = min.xz(), xz1 = max.xz()
            else { r0.x = c0.y; r0.y = c1.x; r1.x = c2.x; r1.y = c2.y; }                            // yz0 = min.yz(), yz1 = max.yz()
        } else {
            r0 = A.rects[3 * pl];
            r1 = A.rects[3 * pl + 1];
        }
        rec.mat = A.rect_mat[pl];
        outward = mk(axis == 0 ? 1. : 0., axis == 1 ? 1. : 0., axis == 2 ? 1. : 0.);
        if (A.texs[A.mats[rec.mat].tex].type == 2) {
            D3 p = add(o, muls(d, h.t));
            double a, b;
            if (axis == 2) { a = p.x; b = p.y; }
            else if (axis == 1) { a = p.x; b = p.z; }
            else { a = p.y; b = p.z; }
            rec.u = (a - r0.x) / (r1.x - r0.x);
            rec.v = (b - r0.y) / (r1.y - r0.y);
        }
    } else if (GENERAL) {  // triangle, mesh.rs:104-137 : recompute the barycentrics (same ops, same bits)
        uint4 tr = A.tris[pl];
        rec.mat = (int)tr.w;
        double t, b1 = 0., b2 = 0.;
        tri_hit(A.tripre + 5 * pl, o, d, -INFINITY, INFINITY, t, b1, b2);
        double b0 = 1.0 - b1 - b2;
        D3 na = ld3(A.vnrm, tr.x), nb = ld3(A.vnrm, tr.y), nc = ld3(A.vnrm, tr.z);
        outward = unit(add(add(muls(na, b0), muls(nb, b1)), muls(nc, b2)), err);
    } else {
        outward = mk(0., 0., 0.);
        rec.mat = 0;
    }
    // HitRecord::new, hit.rs:16-39
    rec.p = add(o, muls(d, h.t));
    rec.front_face = dot(d, outward) < 0.;
    rec.normal = unit(rec.front_face ? outward : neg(outward), err);
    if (GENERAL && h.xf >= 0) {  // Transform::hit, transform.rs:157-161 (Q7, Q8)
        const double* M = A.xforms + 32 * h.xf + 16;
        D3 on = xf_dir(M, rec.normal);
        rec.p = xf_point(M, rec.p);
        rec.front_face = dot(d, on) < 0.;  // d is still the object-space ray
        D3 un = unit(on, err);
        rec.normal = rec.front_face ? un : neg(un);
    }
    return rec;
}

// Material::emitted + Material::scatter, material.rs:88-212.  Returns false on Absorb.
// The material types run as divergent branches of one wave, so what they have in common is done ONCE, before the
// branches: the texture lookup (every type reads its texture exactly once), the unit-sphere sample that Lambertian,
// DiffuseLight and Metal all draw first, and the one normalisation each type performs (of that sample for the diffuse
// types, of the incoming direction for Metal and Dielectric).  Per lane the operations and their order are unchanged.
template <int GENERAL = 1>
DEV bool shade(const Acc& A, const Rec& rec, D3 rdir, Rng& rng, D3& emitted, D3& att, D3& out_dir, bool& diffuse, int* err) {
    const MatDev mt = A.mats[rec.mat];
    const int type = mt.type;
    const bool lamb = (type == 0 || type == 3);  // Lambertian / DiffuseLight: Interaction::Diffuse (material.rs:111,207)
    diffuse = lamb;
    const D3 tc = tex_color<GENERAL>(A, mt.tex, rec);
    const double FRAC_1_PI = 0.318309886183790671537767526745028724;
    emitted = (type == 3) ? tc : mk(0., 0., 0.);                                         // material.rs:209-211 (no face test)
    att = (type == 3) ? mk(1. * FRAC_1_PI, 1. * FRAC_1_PI, 1. * FRAC_1_PI) : tc;         // material.rs:201-203
    D3 rs = mk(0., 0., 0.);
    if (type != 2) rs = random_in_unit_sphere(rng);  // Metal draws its fuzz sample even when fuzz == 0 (Q4)
    const D3 u = unit(lamb ? rs : rdir, err);
    if (lamb) {  // scattered_direction, material.rs:92-98
        PH_BEGIN(ph_b);
        D3 dir = add(rec.normal, u);
        if (near_zero(dir)) dir = rec.normal;
        out_dir = dir;
        PH_END(4, ph_b);
        return true;
    }
    if (type == 4) {  // Isotropic (material.rs:213-231, commented out in the reference): Ray(p, random_in_unit_sphere()), albedo
        PH_BEGIN(ph_b);
        out_dir = rs;
        PH_END(5, ph_b);
        return true;
    }
    if (type == 1) {  // Metal, material.rs:126-139
        PH_BEGIN(ph_b);
        D3 reflected = reflect(u, rec.normal);
        D3 dir = add(reflected, muls(rs, mt.param));
        if (dot(dir, rec.normal) > 0.) {
            out_dir = dir;
            PH_END(6, ph_b);
            return true;
        }
        PH_END(6, ph_b);
        return false;  // Absorb (Q15)
    }
    // Dielectric, material.rs:157-188
    PH_BEGIN(ph_b);
    double ratio = rec.front_face ? mt.inv_ir : mt.param;  // 1.0 / ir, from the host
    const D3 ud = u;
    double cos_theta = fmin(dot(neg(ud), rec.normal), 1.0);
    double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
    bool cannot_refract = ratio * sin_theta > 1.0;
    bool do_reflect = cannot_refract;
    if (!do_reflect) {  // the random number is consumed only when refraction is possible (Q16)
        double r0 = rec.front_face ? mt.r0_front : mt.r0_back;  // ((1 - ratio) / (1 + ratio))^2, from the host (material.rs:150-153)
        double b = 1. - cos_theta;
        double b2 = b * b;
        double b4 = b2 * b2;
        double refl = r0 + (1. - r0) * (b * b4);
        do_reflect = refl > rng.gen_f64();
    }
    out_dir = do_reflect ? reflect(ud, rec.normal) : refract(ud, rec.normal, ratio);
    PH_END(7, ph_b);
    return true;
}

// ------------------------------------------------------ light importance sampling ----
// Integrator 1 (rt_params.integrator): book-3 MixturePDF semantics on Diffuse interactions; the reference has no pdf
// code (its sketch: the dead Light::sample_li, light.rs:107-124,170-183, and random_point_on_area, light.rs:148-154).
// Trig-free so that the CPU oracle and the device agree bit for bit.  See the oracle's sample_ray_mixture for the spec.
DEV double light_pdf_value(const Acc& A, uint2 l, D3 o, D3 v) {
    const double PI = 3.14159265358979323846264338327950288;
    double t;
    if (l.x == NK_SPHERE) {
        const double2* s = A.spheres + 2 * l.y;
        if (!sphere_hit(s, o, v, sqlen(v), 0.001, INFINITY, t)) return 0.;
        double2 c0 = s[0], c1 = s[1];
        double cos_theta_max = sqrt(1. - c1.y * c1.y / sqlen(sub(mk(c0.x, c0.y, c1.x), o)));
        double solid_angle = 2. * PI * (1. - cos_theta_max);
        return 1. / solid_angle;
    }
    const double2* r = A.rects + 3 * l.y;
    if (!rect_hit(r, 1, o, v, 0.001, INFINITY, t)) return 0.;
    double2 r0 = r[0], r1 = r[1];
    double area = (r1.x - r0.x) * (r1.y - r0.y);
    double distance_squared = t * t * sqlen(v);
    double cosine = fabs(v.y / sqrt(sqlen(v)));
    return distance_squared / (cosine * area);
}
DEV D3 light_random(const Acc& A, uint2 l, D3 o, Rng& rng, int* err) {
    if (l.x == NK_SPHERE) {
        const double2* s = A.spheres + 2 * l.y;
        double2 c0 = s[0], c1 = s[1];
        D3 direction = sub(mk(c0.x, c0.y, c1.x), o);
        double distance_squared = sqlen(direction);
        D3 w = unit(direction, err);
        D3 a = (fabs(w.x) > 0.9) ? mk(0., 1., 0.) : mk(1., 0., 0.);
        D3 vv = unit(cross(w, a), err);
        D3 uu = cross(w, vv);
        D3 dsk = random_in_unit_disk(rng);
        double s2 = dsk.x * dsk.x + dsk.y * dsk.y;
        double cos_theta_max = sqrt(1. - c1.y * c1.y / distance_squared);
        double z = 1. + s2 * (cos_theta_max - 1.);
        double rr = sqrt(fmax(0., 1. - z * z));
        double inv_s = (s2 > 0.) ? 1. / sqrt(s2) : 0.;
        double x = dsk.x * inv_s * rr, y = dsk.y * inv_s * rr;
        return add(add(muls(uu, x), muls(vv, y)), muls(w, z));
    }
    const double2* r = A.rects + 3 * l.y;
    double2 r0 = r[0], r1 = r[1], r2 = r[2];
    double u = rng.gen_range_01(), v = rng.gen_range_01();
    D3 p = mk(r0.x + (r1.x - r0.x) * u, r2.x, r0.y + (r1.y - r0.y) * v);
    return sub(p, o);
}
// mixture step after a Diffuse scatter: returns false when the path ends (weight not > 0)
DEV bool mixture_step(const Acc& A, const Rec& rec, Rng& rng, D3 att, D3& beta, D3& dir, int* err) {
    const double PI = 3.14159265358979323846264338327950288;
    if (rng.gen_f64() < 0.5) {
        uint32_t li = (uint32_t)(rng.gen_f64() * (double)A.n_lights);
        if (li >= A.n_lights) li = A.n_lights - 1;
        dir = light_random(A, A.lights[li], rec.p, rng, err);
    }
    double cosine = dot(rec.normal, unit(dir, err));
    double scattering_pdf = (cosine < 0.) ? 0. : cosine / PI;
    double lp = 0.;
    for (uint32_t i = 0; i < A.n_lights; i++) lp = lp + light_pdf_value(A, A.lights[i], rec.p, dir);
    double pdf_val = 0.5 * (lp / (double)A.n_lights) + 0.5 * scattering_pdf;
    double wgt = scattering_pdf / pdf_val;
    if (!(wgt > 0.)) return false;
    beta = muls(elemul(beta, att), wgt);
    return true;
}

// ------------------------------------------------------------ pt_kernel ---
// INTEG: 0 = sample_ray with BSDF sampling; 1 = light/cosine mixture pdf; 2 = the reference's literal SPPM sample_ray:
// the first Diffuse hit adds the pixel's pre-computed photon estimates and ends the path (photon_mapper.rs:345-352)
//
// Where the samples go.  `pixel_color += sample` runs in sample order in the reference (camera.rs:96-101) and f64 addition does
// not commute, so the accumulator must see a pixel's samples in index order whatever the schedule.  A work unit = (8x8 tile,
// <= UNIT_SPP consecutive sample indices).  A wave owns a small RING of unit buffers in global memory (RING_UNITS x 12 KB);
// finished paths store their radiance there, an LDS counter per ring slot tells when a unit is complete, and the wave then
// FOLDS it: lane = pixel, accum += the unit's samples in index order.  Units of one tile are folded in order across waves
// through a per-tile ticket (= number of sample blocks folded).  A wave takes JOBS of JOB_UNITS consecutive units of one tile, so
// only every JOB_UNITS-th fold crosses waves; jobs are dealt sample-major (all tiles' job 0, then all tiles' job 1, ...), so a
// job's predecessor on the same tile was handed out thousands of jobs earlier and is normally long folded; when it is not, the
// wave keeps tracing its newer units and retries (it only waits when it has nothing else to do).  No deadlock: the unfolded job
// with the smallest index is at the head of its wave's ring and its predecessor is folded.
// Accumulator and ticket cross waves on different XCDs (L2s are not coherent): both are accessed ONLY with 8-/4-byte agent-scope
// atomics (served at the memory side), the folding wave drains its stores (s_waitcnt vmcnt(0)) before it advances the ticket.
#ifndef REGEN_MIN
#define REGEN_MIN 8
#endif
#ifndef FOLD_RETRY
#define FOLD_RETRY 4   // main-loop iterations between two looks at a complete head unit whose ticket has not come
#endif
#ifndef COOP_EARLY_FOLD
#define COOP_EARLY_FOLD 1
#endif
#ifndef POOL_MODE
#define POOL_MODE 1    // launches of single-unit jobs fold their unit buffers out of order (next_unit_pool)
#endif
#ifndef FOLD_BATCH
#define FOLD_BATCH 1   // sample indices whose loads are in flight together in fold_units (2: 507.1, 4: 512.4 ms against 502.3 for the headline frame)
#endif
#ifndef FOLD_PERIOD
#define FOLD_PERIOD 4  // the head of the ring is looked at every FOLD_PERIOD-th iteration (every iteration: -1 % on the whole frame)
#endif
#ifndef UNIT_SPP_N
#define UNIT_SPP_N 8
#endif
static const int UNIT_SPP = UNIT_SPP_N;                     // sample indices per work unit, at most
#ifndef RING_UNITS
#define RING_UNITS 8                                         // unit buffers per wave (6 -> 8: a rank's eighth of the headline frame 68.7 -> 67.2 ms, the whole frame 507.6 -> 503.0)
#endif
static const int UNIT_DOUBLES = UNIT_SPP * TILE_PIX * 3;    // 12 KB per unit
static_assert(RING_UNITS * 4 <= 64, "a wave clears its ring bookkeeping with one store per lane");
#ifndef SINGLE_UNITS_BELOW_WAVES
#define SINGLE_UNITS_BELOW_WAVES 1                           // jobs of single units when the rank owns fewer tiles than the GPU has waves (render_tiles)
#endif
#ifndef JOB_UNITS
#define JOB_UNITS 2                                          // consecutive units of one tile per job (one accumulator hand-off per job)
#endif
DEV uint64_t ld_agent(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV void st_agent(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// the same through pointers the compiler knows to be global memory (a generic pointer makes these FLAT accesses that also wait on lgkmcnt)
DEV uint64_t ld_agent_g(const AS_G uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV void st_agent_g(AS_G uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Fold the wave's completed units, oldest first, into the accumulator (see "Where the samples go" above).  Returns how many
// units were folded.  `idle`: the wave has nothing else to do, so it waits for the tile's previous job instead of returning.
// Units of one JOB (consecutive sample blocks of one tile, traced by this wave) sit next to each other in the ring: the tile's
// ticket is looked at once per job (its first unit), the accumulator is loaded once and stored once per run of the job's
// units, and the ticket moves after the job's last unit -- the ~3 memory round trips of a fold are paid per job, not per unit.
// rmeta per ring slot: {tile (local), first sample block of the unit within the launch, samples | first << 8 | last << 9, paths running}
#ifdef RT_TAIL_STATS  // tools-only build: when every wave of pt_kernel entered, reached the main loop and left (s_memrealtime)
__device__ unsigned long long g_tail_end[8192], g_tail_beg[8192], g_tail_entry[8192];
__device__ unsigned int g_tail_xcc[8192];
#endif
#ifdef RT_FOLD_STATS  // tools-only build: {calls, cycles in fold_units, units folded, returns on a ticket mismatch, sleeps}
__device__ unsigned long long g_fold_stats[8];
#endif
__device__ __attribute__((noinline)) int fold_units(const uint32_t* rmeta, const double* wring, double* accum, unsigned int* tickets, bool first_launch,
                                                    int r_head, int r_cnt, bool idle, int lane, int ring_units) {
#ifdef RT_FOLD_STATS
    const unsigned long long t_in = __builtin_amdgcn_s_memtime();
    unsigned long long n_mis = 0, n_sleep = 0;
#endif
    // address spaces spelled out: rmeta is the wave's bookkeeping in LDS, everything else global memory
    const AS_L uint32_t* rm = (const AS_L uint32_t*)rmeta;
    AS_G unsigned int* tk_g = (AS_G unsigned int*)tickets;
    int folded = 0;
    bool have = false;
    uint32_t cur_tile = 0;
    double ax = 0., ay = 0., az = 0.;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's sample stores are in L2
    while (r_cnt > 0) {
        const AS_L uint32_t* m = rm + 4 * r_head;
        if (__hip_atomic_load(&m[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) break;  // paths still running
        const uint32_t tile = m[0], blk = m[1], ns = m[2] & 0xffu;
        const bool first = (m[2] & 0x100u) != 0u, last = (m[2] & 0x200u) != 0u;
        if (first) {
            const uint32_t tk = __hip_atomic_load(&tk_g[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tk != blk) {  // the tile's previous job is not folded yet (rare): trace on unless nothing else is left to do
#ifdef RT_FOLD_STATS
                if (!idle) n_mis++;
                else n_sleep++;
#endif
                if (!idle) break;
                __builtin_amdgcn_s_sleep(4);
                continue;
            }
        }
        AS_G uint64_t* acc = (AS_G uint64_t*)accum + ((size_t)tile * TILE_PIX + (size_t)lane) * 3;
        if (!have) {
            ax = 0.; ay = 0.; az = 0.;
            if (!(blk == 0u && first_launch)) {
                ax = __longlong_as_double(ld_agent_g(acc));
                ay = __longlong_as_double(ld_agent_g(acc + 1));
                az = __longlong_as_double(ld_agent_g(acc + 2));
            }
            have = true;
            cur_tile = tile;
        }
        // pixel_color += sample, in sample order (camera.rs:96-101).  One memory round trip per sample index: loading several indices
        // together (FOLD_BATCH) shortens the fold but its registers come out of pt_kernel's budget -- the kernel's scratch grows from 288 to
        // 336 B per lane with 4 and the whole frame loses 2 %; the waiting wave costs little, three others run on its SIMD.
        const AS_G uint64_t* p = (const AS_G uint64_t*)wring + ((size_t)r_head * UNIT_SPP * TILE_PIX + (size_t)lane) * 3;
        uint32_t si = 0;
#if FOLD_BATCH > 1
        for (; si + FOLD_BATCH <= ns; si += FOLD_BATCH) {
            uint64_t v[3 * FOLD_BATCH];
#pragma unroll
            for (int k = 0; k < 3 * FOLD_BATCH; k++) v[k] = ld_agent_g(p + (size_t)(k / 3) * (TILE_PIX * 3) + (k % 3));
#pragma unroll
            for (int j = 0; j < FOLD_BATCH; j++) {
                ax = ax + __longlong_as_double(v[3 * j]);
                ay = ay + __longlong_as_double(v[3 * j + 1]);
                az = az + __longlong_as_double(v[3 * j + 2]);
            }
            p += FOLD_BATCH * TILE_PIX * 3;
        }
#endif
        for (; si < ns; si++) {
            const uint64_t v0 = ld_agent_g(p), v1 = ld_agent_g(p + 1), v2 = ld_agent_g(p + 2);
            ax = ax + __longlong_as_double(v0);
            ay = ay + __longlong_as_double(v1);
            az = az + __longlong_as_double(v2);
            p += TILE_PIX * 3;
        }
        r_head = (r_head + 1 == ring_units) ? 0 : r_head + 1;
        r_cnt--;
        folded++;
        // the next unit continues this job iff it exists, is complete and is not the first of another job
        bool more = false;
        if (!last && r_cnt > 0) {
            const AS_L uint32_t* n = rm + 4 * r_head;
            more = __hip_atomic_load(&n[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u;
        }
        if (!more) {
            st_agent_g(acc, __double_as_longlong(ax));
            st_agent_g(acc + 1, __double_as_longlong(ay));
            st_agent_g(acc + 2, __double_as_longlong(az));
            have = false;
            if (last) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // accumulator stores done before the ticket moves
                if (lane == 0) __hip_atomic_store(&tk_g[cur_tile], blk + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                break;  // the job's next unit still runs: come back later (it needs no ticket)
            }
        }
    }
#ifdef RT_FOLD_STATS
    if (lane == 0) {
        atomicAdd(&g_fold_stats[0], 1ull);
        atomicAdd(&g_fold_stats[1], (unsigned long long)__builtin_amdgcn_s_memtime() - t_in);
        atomicAdd(&g_fold_stats[2], (unsigned long long)folded);
        atomicAdd(&g_fold_stats[3], n_mis);
        atomicAdd(&g_fold_stats[4], n_sleep);
    }
#endif
    return folded;
}

// What a wave does when its pool is exhausted and lanes are free: fold completed units, then start the next unit of its job or
// fetch the next job.  Everything it needs lives in LDS (per-wave state `wst`, launch constants `cfg`), so that the hot loop
// carries only {pool, next, s0, tx, ty, cur_slot, finished} for all of this; out of line for the same reason (it runs about once
// per 512 paths).
//   wst: {r_head, r_cnt, job_tile, job_blk0, job_n, job_k, more_jobs, job's level}      cfg: see CFG_* below
enum { CFG_N_JOBS, CFG_TILES_OWNED, CFG_JOB_UNITS, CFG_SUBS_PER_TILE, CFG_WORLD, CFG_RANK, CFG_TILES_X, CFG_S_BEGIN, CFG_S_END, CFG_SUB_SPP,
       CFG_WIDTH, CFG_HEIGHT, CFG_RING_UNITS /* unit buffers per wave: RING_UNITS, kernel 6: WF_RING_UNITS */,
       CFG_JOB_LISTS /* lists next_unit() deals the tiles from: JOB_LISTS (one per XCD) or 1 */, CFG_LVL = 16 /* RenderK::lvl, 25 words */,
       CFG_WORDS = 48 };
static_assert(SCHED_LEVELS == 4, "RenderK::lvl holds SCHED_LEVELS + 1 rows");
struct UnitInfo {
    int pool;      // paths of the unit that was started (0: none was)
    int s0, tx, ty, cur_slot;
    int finished;  // no job left, every unit folded: the wave may exit once its lanes are dead
};
// next_unit()'s result comes back in vector registers; it is the same in every lane, and reading it through readfirstlane tells the
// compiler so (the unit's six words then live in scalar registers across pt_kernel's main loop: -3 ms on the 507 ms headline frame.  NOT in pt_kernel_coop: C4 888 -> 835 Msamples/s with it)
__device__ __forceinline__ UnitInfo uniform_unit(const UnitInfo& v) {
    UnitInfo u;
    u.pool = __builtin_amdgcn_readfirstlane(v.pool); u.s0 = __builtin_amdgcn_readfirstlane(v.s0);
    u.tx = __builtin_amdgcn_readfirstlane(v.tx); u.ty = __builtin_amdgcn_readfirstlane(v.ty);
    u.cur_slot = __builtin_amdgcn_readfirstlane(v.cur_slot); u.finished = __builtin_amdgcn_readfirstlane(v.finished);
    return u;
}
// next_unit() for launches of single-unit jobs (RenderK::pool_mode, the POOL variants of the kernels; a rank that owns fewer tiles than
// the GPU has waves): the wave's unit
// buffers are a POOL.  With the FIFO a complete unit at the head whose tile's ticket has not come keeps every unit behind it from being
// folded, whatever the state of THEIR tiles, and with two or three waves per tile such heads are the rule (1/16 of the headline frame:
// 1.9 M ticket mismatches for 0.2 M units).  Here every complete unit whose turn it is gets folded, whichever slot it is in, and a new
// unit takes any free slot (kernel 6's next_unit_wf works the same way).  rmeta per slot: {tile (local), unit index, samples | 0x100 (slot
// in use), paths still running}; wst: [0] the slot of the oldest unit (the main loop looks at it for the early fold), [1] slots in use,
// [6] jobs left.  take = false: fold only; the number of units folded comes back in cur_slot.
__device__ __attribute__((noinline)) UnitInfo next_unit_pool(uint32_t* wst_, uint32_t* rmeta_, const int* cfg_, const double* wring, double* accum,
                                                             unsigned int* tickets, unsigned int* counter, bool all_dead, int lane, bool take = true) {
    AS_L uint32_t* wst = (AS_L uint32_t*)wst_;
    AS_L uint32_t* rmeta = (AS_L uint32_t*)rmeta_;
    const AS_L int* cfg = (const AS_L int*)cfg_;
    int n_used = (int)wst[1];
    bool more_jobs = wst[6] != 0u;
    const bool first_launch = cfg[CFG_S_BEGIN] == 0;
    const int ring_units = cfg[CFG_RING_UNITS];  // <= 64: one slot per lane
    AS_G unsigned int* tk_g = (AS_G unsigned int*)tickets;
    int folded = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's sample stores are in L2
    for (int round = 0; n_used > 0; round++) {
        bool can = false;
        if (lane < ring_units) {
            const AS_L uint32_t* m = rmeta + 4 * lane;
            if ((m[2] & 0x100u) != 0u && __hip_atomic_load(&m[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u)
                can = __hip_atomic_load(&tk_g[m[0]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == m[1];
        }
        uint64_t mask = __ballot(can);
        if (mask == 0ull) {
            // nothing can be folded now.  A wave that can neither start a unit nor trace anything waits for a ticket
            const bool idle = take && all_dead && folded == 0 && (n_used == ring_units || !more_jobs);
            if (!idle) break;
            __builtin_amdgcn_s_sleep(4);
            continue;
        }
        while (mask != 0ull) {
            const int sl = __ffsll((long long)mask) - 1;
            mask &= mask - 1ull;
            const AS_L uint32_t* m = rmeta + 4 * sl;
            const uint32_t tile = m[0], blk = m[1], ns = m[2] & 0xffu;
            AS_G uint64_t* acc = (AS_G uint64_t*)accum + ((size_t)tile * TILE_PIX + (size_t)lane) * 3;
            double ax = 0., ay = 0., az = 0.;
            if (!(blk == 0u && first_launch)) {
                ax = __longlong_as_double(ld_agent_g(acc));
                ay = __longlong_as_double(ld_agent_g(acc + 1));
                az = __longlong_as_double(ld_agent_g(acc + 2));
            }
            const AS_G uint64_t* p = (const AS_G uint64_t*)wring + ((size_t)sl * UNIT_SPP * TILE_PIX + (size_t)lane) * 3;
            for (uint32_t si = 0; si < ns; si++) {  // pixel_color += sample, in sample order (camera.rs:96-101)
                ax = ax + __longlong_as_double(ld_agent_g(p));
                ay = ay + __longlong_as_double(ld_agent_g(p + 1));
                az = az + __longlong_as_double(ld_agent_g(p + 2));
                p += TILE_PIX * 3;
            }
            st_agent_g(acc, __double_as_longlong(ax));
            st_agent_g(acc + 1, __double_as_longlong(ay));
            st_agent_g(acc + 2, __double_as_longlong(az));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // accumulator stores done before the ticket moves
            if (lane == 0) {
                __hip_atomic_store(&tk_g[tile], blk + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                rmeta[4 * sl + 2] = 0u;  // slot free
            }
            n_used--;
            folded++;
        }
        if (round >= 2) break;  // (a fold can make the tile's next unit, also in this pool, foldable: look again, a few times)
    }
    UnitInfo u;
    u.pool = 0; u.s0 = 0; u.tx = 0; u.ty = 0; u.cur_slot = take ? 0 : folded; u.finished = 0;
    if (take && n_used < ring_units && more_jobs) {
        unsigned int job = 0;
        if (lane == 0) job = atomicAdd(counter, 1u);
        job = __builtin_amdgcn_readfirstlane(job);
        if (job >= (unsigned)cfg[CFG_N_JOBS]) {
            more_jobs = false;
        } else {
            const int job_tile = (int)(job % (unsigned)cfg[CFG_TILES_OWNED]);  // sample-major: all tiles' round k before any tile's round k+1
            const int round = (int)(job / (unsigned)cfg[CFG_TILES_OWNED]);
            int lv = 0;
            while (round >= cfg[CFG_LVL + 5 * (lv + 1)]) lv++;
            const AS_L int* L = cfg + CFG_LVL + 5 * lv;
            const int sub_i = L[1] + (round - L[0]);  // (single-unit jobs: a round is a unit)
            const uint64_t fm = __ballot(lane < ring_units && (rmeta[4 * lane + 2] & 0x100u) == 0u);
            const int slot = __ffsll((long long)fm) - 1;
            const int tile = job_tile * cfg[CFG_WORLD] + cfg[CFG_RANK];
            u.tx = tile % cfg[CFG_TILES_X];
            u.ty = tile / cfg[CFG_TILES_X];
            u.s0 = L[2] + (sub_i - L[1]) * L[3];
            const int s1 = min(u.s0 + L[3], L[5 + 2]);
            u.pool = (s1 - u.s0) * TILE_PIX;
            u.cur_slot = slot;
            n_used++;
            if (lane == 0) {
                AS_L uint32_t* m = rmeta + 4 * slot;
                m[0] = (uint32_t)job_tile;
                m[1] = (uint32_t)sub_i;
                m[2] = (uint32_t)(s1 - u.s0) | 0x100u;
                const int w_in = min(TILE_W, cfg[CFG_WIDTH] - u.tx * TILE_W), h_in = min(TILE_H, cfg[CFG_HEIGHT] - u.ty * TILE_H);
                __hip_atomic_store(&m[3], (uint32_t)((s1 - u.s0) * w_in * h_in), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    u.finished = (take && u.pool == 0 && !more_jobs && n_used == 0) ? 1 : 0;
    // the oldest unit in the pool (smallest unit index: rounds are dealt in order) is the one the main loop watches
    int oldest = 0;
    uint32_t oldest_blk = 0xffffffffu;
    for (int k = 0; k < ring_units; k++) {
        const bool is_new = take && u.pool > 0 && k == u.cur_slot;  // (lane 0's write of the new unit's rmeta may not be visible yet)
        if (!is_new && (rmeta[4 * k + 2] & 0x100u) != 0u && rmeta[4 * k + 1] < oldest_blk) {
            oldest_blk = rmeta[4 * k + 1];
            oldest = k;
        }
    }
    if (oldest_blk == 0xffffffffu && take && u.pool > 0) oldest = u.cur_slot;
    if (lane == 0) {
        wst[0] = (uint32_t)oldest;
        wst[1] = (uint32_t)n_used;
        wst[6] = more_jobs ? 1u : 0u;
    }
    return u;
}

#ifndef JOB_LISTS
#define JOB_LISTS 8u    // job lists of next_unit(): one per XCD (a power of two)
#endif
#ifndef JOB_CHUNK
#define JOB_CHUNK 256u  // consecutive tiles per chunk (64: the same)
#endif
__device__ __attribute__((noinline)) UnitInfo next_unit(uint32_t* wst_, uint32_t* rmeta_, const int* cfg_, const double* wring, double* accum,
                                                        unsigned int* tickets, unsigned int* counter, bool all_dead, int lane, bool take = true) {
    // the bookkeeping lives in LDS: typed pointers keep these DS accesses instead of FLAT ones
    AS_L uint32_t* wst = (AS_L uint32_t*)wst_;
    AS_L uint32_t* rmeta = (AS_L uint32_t*)rmeta_;
    const AS_L int* cfg = (const AS_L int*)cfg_;
    int r_head = (int)wst[0], r_cnt = (int)wst[1], job_tile = (int)wst[2], job_blk0 = (int)wst[3], job_n = (int)wst[4], job_k = (int)wst[5];
    int job_lvl = (int)wst[7];
    bool more_jobs = wst[6] != 0u;
    const int ring_units = cfg[CFG_RING_UNITS];
    UnitInfo u;
    u.pool = 0; u.s0 = 0; u.tx = 0; u.ty = 0; u.cur_slot = 0; u.finished = 0;
    if (r_cnt > 0 && __hip_atomic_load(&rmeta[4 * r_head + 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) {
        const bool idle = all_dead && (r_cnt == ring_units || (job_k >= job_n && !more_jobs));  // nothing can be started: wait for the ticket
#ifdef RT_NOFOLD_TEST  // timing experiment only: results are wrong
        const int folded = 1;
        (void)idle;
#else
        const int folded = fold_units(rmeta_, wring, accum, tickets, cfg[CFG_S_BEGIN] == 0, r_head, r_cnt, idle, lane, ring_units);
#endif
        r_head += folded;
        if (r_head >= ring_units) r_head -= ring_units;
        r_cnt -= folded;
        if (!take) u.cur_slot = folded;
    }
    if (take && r_cnt < ring_units && (job_k < job_n || more_jobs)) {
        if (job_k >= job_n) {  // next job: job_units consecutive sample blocks of one tile
            unsigned int job = 0;
// The rank's tiles are dealt from JOB_LISTS lists, one per XCD: chunks of JOB_CHUNK consecutive tiles go round-robin to the lists
            // (chunk c -> list c % JOB_LISTS), the workgroups of XCD x (= blockIdx.x % 8: workgroups are dispatched round-robin over the
            // XCDs) deal list x from its own counter, round by round like the whole sequence, and move on to the next list with work
            // left when theirs is through (the wave remembers where it last found work).  An XCD then folds, and re-reads through
            // its own L2, the tiles of its own chunks: C4 858 -> 909 Msamples/s (TCC requests per sample 82.9 -> 71.1, fabric bytes
            // 3 524 -> 3 034; one list per CU instead: 850-880; profiles/r04/c4_audit.md).  Which wave traces what never changes the image.
            // Kernel 5 only: the LDS-resident scenes read nothing through L2 and lose 0.4-2 % to the lists' staggered ends (cfg: 1 list).
            const unsigned n_tiles = (unsigned)cfg[CFG_TILES_OWNED], rounds = (unsigned)cfg[CFG_N_JOBS] / n_tiles;
            const unsigned n_chunks = (n_tiles + JOB_CHUNK - 1u) / JOB_CHUNK, last_rem = n_tiles - (n_chunks - 1u) * JOB_CHUNK;
            unsigned list_tiles = 0u, list = 0u;
            bool found = false;
            const unsigned n_lists = (unsigned)cfg[CFG_JOB_LISTS];  // a power of two (1: one list = the whole sequence, sample-major)
            unsigned probe = (unsigned)job_lvl >> 8;  // lists before this one (counted from the wave's own) are through
            job_lvl &= 0xff;
            for (; probe < n_lists && !found; probe++) {
                list = (blockIdx.x + probe) & (n_lists - 1u);
                const unsigned mine = list < n_chunks ? (n_chunks - list + n_lists - 1u) / n_lists : 0u;
                list_tiles = mine * JOB_CHUNK - ((mine != 0u && ((n_chunks - 1u) & (n_lists - 1u)) == list) ? JOB_CHUNK - last_rem : 0u);
                if (list_tiles == 0u) continue;
                if (lane == 0) job = atomicAdd(counter + 16 + list, 1u);
                job = __builtin_amdgcn_readfirstlane(job);
                found = job < list_tiles * rounds;
            }
            const unsigned list_cursor = found ? probe - 1u : n_lists;
            if (!found) {
                more_jobs = false;  // every list is through, for every wave: the counters only grow
            } else {
                const unsigned idx = job % list_tiles;
                job_tile = (int)(((idx / JOB_CHUNK) * n_lists + list) * JOB_CHUNK + idx % JOB_CHUNK);
                const int round = (int)(job / list_tiles);
                job_lvl = 0;
                while (round >= cfg[CFG_LVL + 5 * (job_lvl + 1)]) job_lvl++;  // (the entry behind the last level holds the number of rounds)
                const AS_L int* L = cfg + CFG_LVL + 5 * job_lvl;
                job_blk0 = L[1] + (round - L[0]) * L[4];
                job_n = min(L[4], L[5 + 1] - job_blk0);
                job_k = 0;
            }
            job_lvl |= (int)(list_cursor << 8);
        }
        if (job_k < job_n) {  // start the job's next unit
            const int tile = job_tile * cfg[CFG_WORLD] + cfg[CFG_RANK];
            u.tx = tile % cfg[CFG_TILES_X];
            u.ty = tile / cfg[CFG_TILES_X];
            const int sub_i = job_blk0 + job_k;
            const AS_L int* L = cfg + CFG_LVL + 5 * (job_lvl & 0xff);
            u.s0 = L[2] + (sub_i - L[1]) * L[3];
            const int s1 = min(u.s0 + L[3], L[5 + 2]);
            u.pool = (s1 - u.s0) * TILE_PIX;
            u.cur_slot = r_head + r_cnt;
            if (u.cur_slot >= ring_units) u.cur_slot -= ring_units;
            r_cnt++;
            if (lane == 0) {
                AS_L uint32_t* m = rmeta + 4 * u.cur_slot;
                m[0] = (uint32_t)job_tile;
                m[1] = (uint32_t)sub_i;
                m[2] = (uint32_t)(s1 - u.s0) | (job_k == 0 ? 0x100u : 0u) | (job_k + 1 == job_n ? 0x200u : 0u);
                // paths the unit will run: its pixels inside the image (edge tiles are partial) x its sample indices
                const int w_in = min(TILE_W, cfg[CFG_WIDTH] - u.tx * TILE_W), h_in = min(TILE_H, cfg[CFG_HEIGHT] - u.ty * TILE_H);
                __hip_atomic_store(&m[3], (uint32_t)((s1 - u.s0) * w_in * h_in), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            job_k++;
        }
    }
    u.finished = (u.pool == 0 && !more_jobs && job_k >= job_n && r_cnt == 0) ? 1 : 0;
    if (lane == 0) {
        wst[0] = (uint32_t)r_head; wst[1] = (uint32_t)r_cnt; wst[2] = (uint32_t)job_tile; wst[3] = (uint32_t)job_blk0;
        wst[4] = (uint32_t)job_n; wst[5] = (uint32_t)job_k; wst[6] = more_jobs ? 1u : 0u; wst[7] = (uint32_t)job_lvl;
    }
    return u;
}

// GENERAL: 0 = spheres under BVH nodes only, 1 = every primitive of the reference, 2 = 1 + the book-2 extensions (D9: moving spheres, noise textures, an open shutter)
// A/B switch, OFF in the product (measured and rejected in round 5): with -DRT_SLICE_TH=N the MEDIA variants of kernel 2 suspend
// their accel walks (traverse2's SLICE) once fewer than N lanes of the wave are still walking -- at most half of those that entered the walk,
// so that a wave with few paths left (the end of a launch) still moves.  Reduced book-2 scene, N = 24: 34 instead of 55 node steps and 3.6 instead
// of 7.9 leaf sections per iteration, 20 % more iterations, shading at 41 instead of 50 lanes, 25 more spilled registers: 836 against 849
// Msamples/s (N = 12: 859; C5 as named 608 / 625 against 686) -- profiles/r05/phase_c5r_sliced_th24.txt.
#ifndef RT_SLICE_TH
#define RT_SLICE_TH 0
#endif
template <bool LDS, int GENERAL, int ACCEL, int INTEG, bool MEDIA = false, bool POOL = false>
__global__ void __launch_bounds__(PT_BLOCK) pt_kernel(FlatView sv, CamK cam, RenderK rk, double* __restrict__ ring, double* accum,
                                                      unsigned int* tickets, unsigned int* __restrict__ counter, int* __restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef RT_TAIL_STATS
    const unsigned long long tail_entry = __builtin_amdgcn_s_memrealtime();
#endif
    // LDS map: [staged scene tables (LDS variants) | top-of-BVH Node2 cache (scene in L2/HBM)] [kernel 2: per-lane
    // traversal stacks, stack2 x blockDim 32-bit words] [per wave: RING_UNITS x {unit, samples, paths still running}]
    const uint32_t st_begin = (ACCEL == 2) ? sv.stage2_begin : 0u;
    const uint32_t st_end = (ACCEL == 2) ? sv.stage2_end : sv.stage_bytes;
    uint32_t staged = 0;  // bytes of LDS in front of the stacks
    Acc A;
    if (LDS && ACCEL == 2) {
        // [spheres .. n2) as it is (the Node2 array is the last staged section, flatten.cpp); the Node2 array itself is EXPANDED
        // into the NodeW form behind it
        const uint32_t lo_bytes = sv.off_n2 - st_begin;
        {
            const uint4* src = (const uint4*)(sv.base + st_begin);
            uint4* dst = (uint4*)smem;
            for (uint32_t i = threadIdx.x; i < lo_bytes / 16; i += blockDim.x) dst[i] = src[i];
        }
        char* n2w = smem + lo_bytes;
        for (uint32_t i = threadIdx.x; i < sv.n_nodes2; i += blockDim.x) {
            const uint4* nd = (const uint4*)(sv.base + sv.off_n2) + (size_t)i * NODE2_F4;
            const uint4 a = nd[0], b = nd[1], c = nd[2], d = nd[3];  // (lox0,lox1,loy0,loy1) (loz0,loz1,hix0,hix1) (hiy0,hiy1,hiz0,hiz1) (c0,c1,-,-)
            const uint32_t c0 = wide_ref(d.x), c1 = wide_ref(d.y);
            char* w = n2w + wide_ref(i);
            const uint4 lo0 = make_uint4(a.x, a.y, a.z, a.w), lo1 = make_uint4(b.x, b.y, c0, c1);
            const uint4 hi0 = make_uint4(b.z, b.w, c.x, c.y), hi1 = make_uint4(c.z, c.w, c0, c1);
            ((uint4*)w)[0] = lo0; ((uint4*)w)[1] = lo1;
            ((uint4*)(w + NODEW_FAR))[0] = hi0; ((uint4*)(w + NODEW_FAR))[1] = hi1;
            ((uint4*)(w + 2 * NODEW_FAR))[0] = lo0; ((uint4*)(w + 2 * NODEW_FAR))[1] = lo1;
        }
        staged = lo_bytes + nodew_bytes(sv.n_nodes2);
        __syncthreads();
        A = make_acc(smem - st_begin, sv.base, sv);
        A.n2 = nullptr;
        A.n2w_lds = (uint32_t)(uintptr_t)(AS_L char*)n2w;
        if (MEDIA) {  // the media's boundary subtrees are walked in the reference-order program, which this kernel does not stage
            A.meta = (const uint2*)(sv.base + sv.off_meta);
            A.boxes = (const double2*)(sv.base + sv.off_boxes);
        }
    } else if (LDS) {
        const uint4* src = (const uint4*)(sv.base + st_begin);
        uint4* dst = (uint4*)smem;
        for (uint32_t i = threadIdx.x; i < (st_end - st_begin) / 16; i += blockDim.x) dst[i] = src[i];
        staged = st_end - st_begin;
        __syncthreads();
        A = make_acc(smem - st_begin, sv.base, sv);
    } else {
        A = make_acc(sv.base, sv.base, sv);
        if (ACCEL == 2 && rk.n_top > 0) {  // scene in L2/HBM: the shallowest levels of every BVH (depth-sorted array) in LDS
            const uint4* src = (const uint4*)(sv.base + sv.off_n2);
            uint4* dst = (uint4*)smem;
            for (uint32_t i = threadIdx.x; i < (uint32_t)rk.n_top * NODE2_F4; i += blockDim.x) dst[i] = src[i];
            __syncthreads();
            A.n2_top = (const AS_L f32x4*)smem;
            A.n2_top_count = (uint32_t)rk.n_top;
            staged = (uint32_t)rk.n_top * (uint32_t)sizeof(Node2);
        }
    }
    uint32_t* stk = (uint32_t*)(smem + staged) + threadIdx.x;
    const int stk_stride = (int)blockDim.x;
    const int lane = threadIdx.x & 63;
    const uint64_t lanemask_lt = (1ull << lane) - 1ull;
    const int wave = threadIdx.x >> 6;
    // bookkeeping in LDS behind the stacks: per wave RING_UNITS x {tile, sample block, samples | flags, paths still running} and
    // 8 words of job / ring state; per block the launch constants next_unit() reads
    uint32_t* book = (uint32_t*)(smem + staged) + (size_t)((ACCEL == 2) ? sv.stack2 : 0u) * PT_BLOCK;
    uint32_t* rmeta = book + (size_t)wave * RING_UNITS * 4;
    uint32_t* wst = book + (size_t)(PT_BLOCK / 64) * RING_UNITS * 4 + (size_t)wave * 8;
    int* cfg = (int*)(book + (size_t)(PT_BLOCK / 64) * (RING_UNITS * 4 + 8));
    if (lane < 8) wst[lane] = (lane == 6) ? 1u : 0u;  // nothing in flight, no job yet, jobs left
    if (POOL && lane < RING_UNITS * 4) rmeta[lane] = 0u;  // no slot in use (next_unit_pool reads the flags of slots it never wrote)
    if (threadIdx.x == 0) {
        cfg[CFG_N_JOBS] = rk.n_units; cfg[CFG_TILES_OWNED] = rk.tiles_owned; cfg[CFG_JOB_UNITS] = rk.job_units;
        cfg[CFG_SUBS_PER_TILE] = rk.subs_per_tile; cfg[CFG_WORLD] = rk.world; cfg[CFG_RANK] = rk.rank; cfg[CFG_TILES_X] = rk.tiles_x;
        cfg[CFG_S_BEGIN] = rk.s_begin; cfg[CFG_S_END] = rk.s_end; cfg[CFG_SUB_SPP] = rk.sub_spp; cfg[CFG_WIDTH] = rk.width;
        cfg[CFG_HEIGHT] = rk.height;
        cfg[CFG_RING_UNITS] = RING_UNITS;
        cfg[CFG_JOB_LISTS] = 1;
#pragma unroll
        for (int i = 0; i < 25; i++) cfg[CFG_LVL + i] = rk.lvl[i / 5][i % 5];
    }
    if (GENERAL == 2 && rk.time_slots) A.time_lds = (uint32_t)(uintptr_t)(AS_L char*)(cfg + CFG_WORDS);  // (8-aligned: every section before it is)
    __syncthreads();
    double* wring = ring + ((size_t)blockIdx.x * (PT_BLOCK / 64) + (size_t)wave) * RING_UNITS * UNIT_DOUBLES;

    // current work unit (wave-uniform); `pool` paths, of which `next` have been handed out.  A wave does not drain a unit
    // before it takes the next one: as soon as the pool is empty and a lane is free the next unit is started, so the lanes
    // still finishing long paths of the old unit run beside fresh paths of the new one (a finished path knows where its
    // sample goes: out_slot is per lane).  Only the end of the launch has a tail.
    int tx = 0, ty = 0, s0 = 0, pool = 0, next = 0, cur_slot = 0;
    bool finished = false;

    bool alive = false;
    D3 o = mk(0, 0, 0), d = mk(0, 0, 1), beta = mk(1, 1, 1), L = mk(0, 0, 0);
    int depth = 0;
    int pix_id = 0;
    uint32_t out_slot = 0;  // this path's sample in the wave's ring: (ring slot * UNIT_SPP + sample within the unit) * 64 + pixel
    Rng rng;
    rng.s = 0;
#ifdef RT_TAIL_STATS
    const unsigned long long tail_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    int fold_wait = 0;  // iterations until the head of the ring is looked at again (wave-uniform)
    constexpr bool SLICE = ACCEL == 2 && MEDIA && RT_SLICE_TH > 0;
    WalkState ws;  // SLICE: this lane's suspended walk (cur == REF_DONE: none), its best hit so far and the two tracked minima
    ws.cur = REF_DONE;
    ws.sp = 0u;
    ws.cur_xf = -1;
    Hit hs;
    hs.t = INFINITY; hs.node = -1; hs.xf = -1; hs.kp = 0;
    double ws_t[2] = {INFINITY, INFINITY};
#ifdef RTAMD_PHASE_STATS
    if (threadIdx.x < 48) s_ph[threadIdx.x] = 0ull;
    __syncthreads();
#endif
    {
        for (;;) {
            PH_BEGIN(ph_it0);
            // ---- regeneration: dead lanes pull the next (pixel, sample) of the pool ----
            uint64_t dead = __ballot(!alive);
            // the regeneration code runs for the whole wave however few lanes need it: wait until REGEN_MIN lanes are free
            // (measured: 1 -> 2142, 4 -> 2174, 8 -> 2228, 12 -> 2210, 16 -> 2189, 24 -> 2094 Msamples/s)
            if ((int)__popcll(dead) < REGEN_MIN && dead != ~0ull) dead = 0ull;
            // ---- pool exhausted: fold what is complete, start the next unit (next_unit, out of line) ----
            // A complete unit at the head of the ring is folded at once, through the same call (take = false: fold only; one call
            // site -- a second one in this loop costs 0.6 % of the whole frame in spills).  next_unit() used to fold only when the
            // wave needed a new unit, about once per millisecond; when a rank owns fewer tiles than the GPU has waves (a frame split
            // over 8 GPUs: 2 812 tiles for 4 096 waves) consecutive sample blocks of one tile are traced at the same time by
            // different waves and their folds form a chain -- every hop waited for the holder's next unit boundary, the rings
            // filled with complete units whose turn had not come and the waves slept on tickets (one rank's eighth of the headline
            // frame: 75.3 ms instead of 63.4; 81 M sleeps against 0.7 M for the whole frame on one GPU).
            const bool need_unit = dead != 0ull && next >= pool && !finished;
            bool head_ready = false;
            if (FOLD_PERIOD > 0 && !need_unit && --fold_wait < 0) {
                fold_wait = FOLD_PERIOD - 1;
                head_ready = __builtin_amdgcn_readfirstlane(wst[1]) != 0u &&
                             __builtin_amdgcn_readfirstlane(__hip_atomic_load(&rmeta[4 * __builtin_amdgcn_readfirstlane(wst[0]) + 3], __ATOMIC_RELAXED,
                                                                              __HIP_MEMORY_SCOPE_WORKGROUP)) == 0u;
            }
            if (need_unit || head_ready) {
                // (POOL: a variant of its own -- the function called here decides which registers the loop may keep across the call, and the
                // whole-frame kernel lost 1.4 % when it could reach both)
                const UnitInfo u = uniform_unit(POOL ? next_unit_pool(wst, rmeta, cfg, wring, accum, tickets, counter, need_unit && dead == ~0ull, lane, need_unit)
                                                     : next_unit(wst, rmeta, cfg, wring, accum, tickets, counter, need_unit && dead == ~0ull, lane, need_unit));
                if (need_unit) {
                    if (u.pool > 0) {
                        pool = u.pool;
                        next = 0;
                        s0 = u.s0;
                        tx = u.tx;
                        ty = u.ty;
                        cur_slot = u.cur_slot;
                    }
                    finished = u.finished != 0;
                } else if (u.cur_slot == 0) {  // (fold only: the number of units folded comes back in cur_slot)
                    fold_wait = FOLD_RETRY;  // its ticket has not come: not every iteration
                }
            }
            if (dead != 0ull && next < pool) {
                int k = next + __popcll(dead & lanemask_lt);
                next = min(next + (int)__popcll(dead), pool);
                if (!alive && k < pool) {
                    int pix = k & (TILE_PIX - 1), s = s0 + (k >> 6);
                    int x = tx * TILE_W + (pix & (TILE_W - 1)), y = ty * TILE_H + (pix >> 3);
                    if (x < rk.width && y < rk.height) {
                        // camera.rs:97-99 + Camera::get_ray camera.rs:57-64
                        rng.seed_stream(rk.seed, (uint64_t)y * (uint64_t)rk.width + (uint64_t)x, (uint64_t)s);
                        double u = ((double)x + rng.gen_f64()) / (double)(rk.width - 1);
                        double v = ((double)y + rng.gen_f64()) / (double)(rk.height - 1);
                        double st = 1.0 - v;
                        D3 rd = muls(random_in_unit_disk(rng), cam.lens_radius);  // drawn even for aperture 0 (Q4)
                        if (GENERAL == 2 && rk.time1 > rk.time0) {  // D9: ray(origin + offset, .., random_double(time0, time1)), after the lens sample
                            const double tm = rng.gen_range(rk.time0, rk.time1);
                            if (A.time_lds) *(AS_L double*)(uintptr_t)(A.time_lds + 8u * threadIdx.x) = tm;
                        } else if (GENERAL == 2 && A.time_lds) {
                            *(AS_L double*)(uintptr_t)(A.time_lds + 8u * threadIdx.x) = rk.time0;
                        }
                        D3 offset = add(muls(cam.u, rd.x), muls(cam.v, rd.y));
                        o = add(cam.origin, offset);
                        d = sub(sub(add(add(cam.llc, muls(cam.horizontal, u)), muls(cam.vertical, st)), cam.origin), offset);
                        beta = mk(1., 1., 1.);
                        L = mk(0., 0., 0.);
                        depth = rk.max_depth;
                        pix_id = y * rk.width + x;
                        out_slot = ((uint32_t)cur_slot * (uint32_t)UNIT_SPP + (uint32_t)(k >> 6)) * (uint32_t)TILE_PIX + (uint32_t)pix;
                        alive = true;
                    }  // (a pixel of an edge tile outside the image gets no path; finalize_kernel zeroes it whatever the fold adds)
                }
            }
            PH_END(0, ph_it0);
            if (__ballot(alive) == 0ull) {
                if (next >= pool && finished) break;
                continue;
            }
            // ---- one path segment: sample_ray's loop body, photon_mapper.rs:335-362 ----
            if (alive) {
                PH_EV(13);
                PH_BEGIN(ph_t0);
                Hit h;
                if (SLICE) {  // the walk goes on until fewer than slice_th lanes are left in it; those come back next iteration (ws.cur != REF_DONE)
                    const int n_walk = (int)__popcll(__ballot(1));
                    h = traverse2_media<GENERAL, !LDS, LDS, true>(A, sv.n_media, stk, stk_stride, o, d, rk.t_min, rng, &ws, &hs, ws_t, min((int)RT_SLICE_TH, n_walk >> 1));
                    hs = h;
                } else {
                    h = (ACCEL == 2) ? (MEDIA ? traverse2_media<GENERAL, !LDS, LDS>(A, sv.n_media, stk, stk_stride, o, d, rk.t_min, rng)
                                               : traverse2<GENERAL, false, !LDS, LDS>(A, stk, stk_stride, o, d, rk.t_min, INFINITY))
                                     : traverse<GENERAL, MEDIA>(A, o, d, rk.t_min, INFINITY, &rng);
                }
                PH_END(1, ph_t0);
              if (!SLICE || ws.cur == REF_DONE) {
                PH_BEGIN(ph_p0);
                bool done = true;
                if (h.node >= 0 && depth > 0) {  // Q12: depth test after the hit, before emission
                    depth -= 1;
                    PH_BEGIN(ph_m0);
                    Rec rec = materialize<GENERAL>(A, h, o, d, err);
                    PH_END(2, ph_m0);
                    D3 emitted, att, ndir;
                    bool diffuse;
                    PH_BEGIN(ph_s0);
                    bool scattered = shade<GENERAL>(A, rec, d, rng, emitted, att, ndir, diffuse, err);
                    PH_END(3, ph_s0);
                    L = add(L, elemul(beta, emitted));  // radiance += throughput * Le
                    if (scattered) {  // Diffuse continues like Specular/Reflect/Refract (photon_mapper.rs:346-347)
                        bool go = true;
                        if (INTEG == 2 && diffuse) {
                            const double* e = rk.sppm_est + 6 * (size_t)pix_id;
                            L = add(L, elemul(beta, mk(e[0], e[1], e[2])));  // caustic estimate
                            L = add(L, elemul(beta, mk(e[3], e[4], e[5])));  // global estimate
                            go = false;
                        } else if (INTEG == 1 && diffuse) {
                            PH_EV(14);
                            go = mixture_step(A, rec, rng, att, beta, ndir, err);
                        } else {
                            beta = elemul(beta, att);
                        }
                        if (go) {
                            o = rec.p;
                            d = ndir;
                            done = false;
                        }
                    }
                }
                if (done) {
                    double* dst = wring + 3 * (size_t)out_slot;
                    dst[0] = L.x;
                    dst[1] = L.y;
                    dst[2] = L.z;
                    atomicSub(rmeta + 4 * (out_slot >> 9) + 3, 1u);  // one path less running in that ring slot (UNIT_SPP * 64 = 512 per slot)
                    alive = false;
                }
                PH_END(8, ph_p0);
              }
            }
        }
    }
#ifdef RTAMD_PHASE_STATS
    __syncthreads();
    if (threadIdx.x < 48) atomicAdd(&g_phase[threadIdx.x], s_ph[threadIdx.x]);
#endif
#ifdef RT_TAIL_STATS
    if (lane == 0) {
        g_tail_end[blockIdx.x * (PT_BLOCK / 64) + wave] = __builtin_amdgcn_s_memrealtime();
        g_tail_beg[blockIdx.x * (PT_BLOCK / 64) + wave] = tail_t0;
        g_tail_entry[blockIdx.x * (PT_BLOCK / 64) + wave] = tail_entry;
        g_tail_xcc[blockIdx.x * (PT_BLOCK / 64) + wave] = 0u;
    }
#endif
}

// ------------------------------------------------------- pt_kernel_coop (kernel 5) ---
// Path tracer for scenes whose mesh instances are LARGE (object-space BVHs far outside LDS; BASELINE config C4).
// On the plain kernel a third of a wave's rays reach the instance and walk 35..100 dependent steps while the other lanes
// wait (lane utilisation 0.15, profiles/r02/pmc_summary_c4.csv).  Here those walks are re-packed ACROSS the waves of a
// workgroup, and the workgroup has more paths in flight than lanes so that no lane waits:
//   * a lane walks the world-space BVH to the end with instance items DEFERRED (traverse2<.., DEFER>: a bit per instance);
//   * it then ENTERS the first deferred instance itself: M^-1 * ray (Transform::hit, transform.rs:153-156), the ray mapped
//     onto the instance's 16-bit grid, and a walk through the shallowest NodeQ (the first COOP_ENTRY_NODES of the depth-sorted
//     array, cached in LDS).  Rays that only clip the instance's box end here (about 40 % of them) and never leave the lane;
//   * where the walk needs a node or a leaf from memory the path is PARKED: its whole state plus the walk -- node to continue
//     at, stack, best t and tie-break order so far -- goes to a slot of a per-workgroup pool in global memory (plain 16-byte
//     stores; the workgroup's waves share one L1), the slot id to the request ring RQ in LDS; the lane is free at once;
//   * when COOP_BATCH requests wait (or a wave has nothing else to do) a wave SERVES: 64 lanes take 64 requests, continue
//     the walks over the compact object-space data (NodeQ, Tri32: flat.h) and refill from RQ as lanes finish; when it runs
//     thin with nothing to refill from it SUSPENDS the remaining walks (stack to the slot, id back to RQ) instead of
//     dragging a sparse tail; answers go to the slot, its id to the answer ring AQ;
//   * free lanes ADOPT answered paths right before shading and generate new paths right before the walk, so both phases run
//     with (almost) every lane; the answer is merged with the reference's acceptance rule (smaller t, or equal t and later
//     in reference order) -- what the inline walk does when it reaches the instance last.  A path with a second deferred
//     instance on the same segment is posted again instead (fresh request from that instance's root).
// What a path computes is unchanged: every primitive still sees the reference's f64 test with [t_min, best-so-far], the
// closest hit does not depend on the order in which candidates are met (tie rule by `order`), RNG streams belong to the
// path, and a finished path stores its sample into the ring slot of the wave that generated it (any wave of the workgroup
// may finish it; the unit's LDS counter is decremented one iteration later, behind a drained store).  Bit-identical images.
#ifndef COOP_POOL
#define COOP_POOL 1024      // parked paths per workgroup (power of two; 2048: -1 %, the LDS is worth more as node cache)
#endif
// Ring capacity: twice the pool, so that a ring position is written again only after 1024 further pops have gone by since a
// consumer reserved it (a consumer reads its entry a few instructions after it has advanced the head).
#define COOP_RING (2 * COOP_POOL)
static_assert(COOP_RING >= 2 * COOP_POOL && COOP_POOL <= 32768 && (COOP_RING & (COOP_RING - 1)) == 0,
              "kernel 5's rings: 16-bit entries (id + 1), power-of-two capacity of at least twice the pool (see ring_pop)");
// Build requirement (not checkable here): no -mtgsplit -- the waves of a workgroup must share one CU's L1 (record visibility
// after s_waitcnt vmcnt(0), see COOP_REC below).
#ifndef COOP_SPIN_MAX
#define COOP_SPIN_MAX (1u << 16)  // LDS polls a consumer spends on an unpublished ring entry before it declares the ring overrun
#endif
#ifndef COOP_BATCH
#define COOP_BATCH 32       // a wave starts serving once this many requests wait (16: -3 %, 64: -2 %, 128: -5 %)
#endif
#ifndef COOP_REFILL_TH
#define COOP_REFILL_TH 56   // a serving wave goes back for more requests when fewer lanes than this still walk (48: -1 %, 40: -3 %)
#endif
#ifndef COOP_ENTRY_NODES
#define COOP_ENTRY_NODES 64  // depth-sorted NodeQ indices below this are walked by the parking lane itself
#endif
#ifndef COOP_ADOPT_MIN
#define COOP_ADOPT_MIN 8     // free lanes a wave waits for before it adopts answered paths
#endif
#ifndef COOP_SUSPEND_TH
#define COOP_SUSPEND_TH 24  // ... and suspends its walks when fewer than this are left and no request waits
#endif
static const int COOP_MAX_INST = 64;     // one pending bit per instance (the upper 32 travel in the record's last unit, only when a scene has more than 32)
static const int COOP_STACK_MAX = 40;    // stack entries a suspended walk can save (scenes with deeper BVHs use kernel 2)
static const int COOP_REC = 20 + COOP_STACK_MAX / 2 + 2;  // u64 per pool slot, read and written in 16-byte units (two u64):
//   unit 0..2  [0..5]  the segment's ray o, d in WORLD space (the serving lane applies the instance's M^-1: transform.rs:153-156)
//   unit 3     [6] best t so far   [7] kind|payload of the best hit so far << 32 | its order          (as posted with the request)
//   unit 4     [8] node to continue at (initially the instance's root) << 32 | xform of the best hit so far + 1
//              [9] instances still deferred << 32 | stack entries saved + 1 (0: fresh request) << 24 | instance << 16 | sample slot
//   unit 5     [10] t   [11] kind|payload << 32 | order      of the walk so far / of the answer (order unchanged = nothing closer inside)
//   unit 6..9  [12..14] beta   [15..17] L   [18] rng   [19] depth << 32 | pix_id
//   unit 10..  the saved stack of a suspended walk, four entries per unit
//   unit 20    [40] instances 32..63 still deferred (written and read only by scenes with more than 32 instances)
// Records are only touched by waves of one workgroup, i.e. of one CU, whose L1 they share: plain loads and stores, ordered by
// s_waitcnt vmcnt(0) before the slot id is published through LDS (workgroup-scope release / acquire on gfx950).
typedef unsigned long long U2 __attribute__((ext_vector_type(2)));
DEV U2 ld_unit(const uint64_t* rec, int unit) { return ((const U2*)rec)[unit]; }
DEV void st_unit(uint64_t* rec, int unit, uint64_t a, uint64_t b) {
    U2 v;
    v.x = a;
    v.y = b;
    ((U2*)rec)[unit] = v;
}
DEV uint64_t dbits(double x) { return (uint64_t)__double_as_longlong(x); }
DEV double bitsd(uint64_t x) { return __longlong_as_double((long long)x); }

#ifdef RTAMD_COOP_STATS  // tools-only build (tools/build_variant.sh stats -DRTAMD_COOP_STATS): schedule counters of pt_kernel_coop
__device__ unsigned long long g_coop_stats[16];
struct CoopStats {
    unsigned long long ev[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ln[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = 0;  // shader-clock cycles per phase of the main loop
};
__device__ unsigned long long g_coop_time[8];
#define COOP_RING_T0 const unsigned long long t_ring0 = __builtin_amdgcn_s_memtime()
#define COOP_RING_T1 cs.tm[7] += __builtin_amdgcn_s_memtime() - t_ring0
#define COOP_TIME(i)                                                      \
    do {                                                                  \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime();    \
        cs.tm[i] += t_now - cs.t_last;                                    \
        cs.t_last = t_now;                                                \
    } while (0)
#define COOP_STATS_ARG , CoopStats& cs
#define COOP_STATS_PASS , cs
#define COOP_STAT(i, lanes_mask)                                        \
    do {                                                                \
        cs.ev[i] += 1ull;                                               \
        cs.ln[i] += (unsigned long long)__popcll(lanes_mask);           \
    } while (0)
#else
#define COOP_STATS_ARG
#define COOP_STATS_PASS
#define COOP_STAT(i, lanes_mask) \
    do {                         \
    } while (0)
#define COOP_TIME(i) \
    do {             \
    } while (0)
#define COOP_RING_T0
#define COOP_RING_T1
#endif

// Address spaces are spelled out in the pieces that run out of line (or through volatile accesses): the compiler infers them
// from kernel arguments and `extern __shared__`, not from pointers that went through memory, and falls back to FLAT
// instructions (64-bit VALU address arithmetic, both wait counters, a slower path into LDS).
struct CoopRing {  // multi-producer multi-consumer ring of slot ids in LDS; entries are id + 1 (16 bit), 0 = not yet written
    volatile AS_L uint16_t* buf;
    AS_L uint32_t* ht;  // {head, tail}: monotonic counters
    AS_L uint32_t* abort_flag;  // workgroup-wide: set when an entry was not published in time (ring overrun): every wave leaves
};
DEV uint32_t lds_load(const AS_L uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
DEV uint32_t ring_len(const CoopRing& R) { return lds_load(&R.ht[1]) - lds_load(&R.ht[0]); }
// push the ids of the flagged lanes; `drain`: their records were just stored and must be in L2 before the ids are visible
DEV void ring_push(const CoopRing& R, bool push, uint32_t id, int lane, uint64_t lanemask_lt, bool drain) {
    const uint64_t m = __ballot(push);
    if (m == 0ull) return;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = __hip_atomic_fetch_add(&R.ht[1], (uint32_t)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    base = __shfl(base, leader);
    if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (push) R.buf[(base + (uint32_t)__popcll(m & lanemask_lt)) & (COOP_RING - 1)] = (uint16_t)(id + 1u);
}
// pop up to popcount(want) ids; lanes of `want` that get one return it, the others return -1
DEV int ring_pop(const CoopRing& R, uint64_t want, int lane, uint64_t lanemask_lt) {
    if (want == 0ull) return -1;
    uint32_t h0 = 0, k = 0;
    if (lane == 0) {
        for (;;) {
            uint32_t hd = lds_load(&R.ht[0]);
            const uint32_t tl = lds_load(&R.ht[1]);
            if (hd == tl) break;
            const uint32_t n = min((uint32_t)__popcll(want), tl - hd);
            if (__hip_atomic_compare_exchange_strong(&R.ht[0], &hd, hd + n, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                h0 = hd;
                k = n;
                break;
            }
        }
    }
    h0 = __builtin_amdgcn_readfirstlane(h0);
    k = __builtin_amdgcn_readfirstlane(k);
    const int rank = __popcll(want & lanemask_lt);
    int id = -1;
    if (((want >> lane) & 1ull) != 0ull && rank < (int)k) {
        const uint32_t slot = (h0 + (uint32_t)rank) & (COOP_RING - 1);
        // its producer reserved the slot and writes it within a few instructions.  The poll is bounded: if the entry never
        // shows up (a consumer so late that the ring wrapped over its position would have consumed another slot's id), the
        // workgroup gives up -- error bit 4, RT_ERR_INTERNAL -- instead of spinning for ever.
        uint32_t v, spins = 0u;
        do { v = R.buf[slot]; } while (v == 0u && ++spins < COOP_SPIN_MAX);
        if (v == 0u) {
            __hip_atomic_store(R.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            R.buf[slot] = (uint16_t)0;
            id = (int)v - 1;
        }
    }
    return id;
}

struct CoopLds {
    CoopRing rq, aq, fq;        // requests, answers, free pool slots
};

// One while-while pass of the object-space walk for the lanes with `act`: descend to a leaf, test its items (traverse2's
// node and leaf steps; no instances below an instance).  cur == REF_DONE afterwards means the walk is complete.
// TIE: `tied` is set when a candidate shares the best t exactly with the hit the walk holds (see "EXACT ties" above), cleared by a closer hit.
template <bool TIE>
DEV void blas_pass(const Acc& A, bool act, uint32_t* stk, const int stride, D3 o, D3 d, double a, double t_min, Ray32& r, double& ht, int& hnode,
                   uint32_t& hkp, uint32_t& cur, int& sp, int* err, bool& tied) {
    while (act && (cur >> REF_TAG_SHIFT) == 0u) {  // inner node: both children, conservative f32 boxes
        f32x4 q0, q1, q2;
        f32x2 q3;
        if (cur < A.n2_top_count) {
            const AS_L f32x4* p = A.n2_top + NODE2_F4 * cur;
            q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = *(const AS_L f32x2*)(p + 3);
        } else {
            const f32x4* p = (const f32x4*)A.n2 + NODE2_F4 * cur;
            q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = *(const f32x2*)(p + 3);
        }
        float e0, e1;
        const bool h0 = box32(q0.x, q0.z, q1.x, q1.z, q2.x, q2.z, r, e0);
        const bool h1 = box32(q0.y, q0.w, q1.y, q1.w, q2.y, q2.w, r, e1);
        const uint32_t c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
        if (h0 && h1) {
            const bool swap = e1 < e0;
            stk[sp] = swap ? c0 : c1;
            sp += stride;
            cur = swap ? c1 : c0;
        } else if (h0) {
            cur = c0;
        } else if (h1) {
            cur = c1;
        } else if (sp > 0) {
            sp -= stride;
            cur = stk[sp];
        } else {
            cur = REF_DONE;
        }
    }
    if (act && cur != REF_DONE) {  // leaf: the reference's f64 primitive tests (tie rule as in traverse2)
        const uint32_t first = cur & REF_LEAF_FIRST_MASK, cnt = ((cur >> REF_LEAF_COUNT_SHIFT) & 7u) + 1u;
        for (uint32_t i = 0; i < cnt; i++) {
            const uint2 it = A.items2[first + i];
            const uint32_t kind = it.x & NK_MASK, pl = it.x >> NK_BITS;
            double t = 0.;
            bool got = false;
            uint32_t cube_side = 0u;
            const double t_far = TIE ? ht * TIE_W : ht;  // (TIE: a little beyond the best hit, to note near ties -- see TIE_W)
            if (kind == NK_SPHERE) {
                got = sphere_hit(A.spheres + 2 * pl, o, d, a, t_min, t_far, t);
            } else if (kind == NK_RECT_YZ || kind == NK_RECT_XZ || kind == NK_RECT_XY) {
                got = rect_hit(A.rects + 3 * pl, (int)kind - (int)NK_RECT_YZ, o, d, t_min, t_far, t);
            } else if (kind == NK_CUBE) {
                got = cube_hit(A.rects + 3 * (pl >> 3), o, d, t_min, t_far, t, cube_side);
            } else if (kind == NK_TRI) {
                double b1, b2;
                got = tri_hit(A.tripre2 + 5 * (first + i), o, d, t_min, t_far, t, b1, b2);
            } else {
                atomicOr(err, 2);  // an instance below an instance: flatten.cpp refuses such scenes
            }
            if (got && (TIE ? (t < ht || (t == ht && (int)it.y > hnode) || !(t == t)) : (t < ht || (int)it.y > hnode || !(t == t)))) {
                if (TIE) tied = hnode >= 0 && t * TIE_W >= ht;
                ht = t;
                hnode = (int)it.y;
                hkp = it.x + (cube_side << NK_BITS);
                r.best = ray32_best(t);
            } else if (TIE && got && (int)it.y < hnode) {  // within the tie margin of ht and the candidate is the earlier party
                tied = true;
            }
        }
        if (sp > 0) {
            sp -= stride;
            cur = stk[sp];
        } else {
            cur = REF_DONE;
        }
    }
}


// The serving waves' view of the scene, every pointer with its address space.
struct ServeCtx {
    const AS_G u32x4* n2q;       // 2 per NodeQ
    const AS_L u32x4* n2q_top;   // LDS copy of the first n_topq NodeQ
    uint32_t n_topq;
    const AS_G u32x4* tri32;     // 3 per Tri32
    const AS_L double* qgrid;    // 8 per instance
    const AS_L uint32_t* inst2;  // 2 per instance: xform, root
    const AS_L double* xforms;   // 32 per transform
    AS_G U2* pool;               // this workgroup's parked-path records, COOP_REC / 2 units each
    CoopLds C;
    double t_min;
};
// One while-while pass of the object-space walk over the compact encoding (flat.h "Compact instance data") for the lanes with
// `act`: descend to a leaf, test its triangles.  NodeQ boxes are grid integers and `r` is the ray in grid coordinates; triangles
// come as f32 vertices with order and kind|payload in the record; (o, d) is the object-space ray.  cur == REF_DONE afterwards
// means the walk is complete.
// Conservative f32 pre-test in front of the f64 Triangle::hit of the compact records (round 5).  true ONLY IF tri_hit_v below certainly returns
// false -- through one of its barycentric rejects -- so a skipped triangle changes nothing.  Moeller-Trumbore (mesh.rs:57-102) divides three
// numerators by one determinant:  b1 = N1 / dd,  b2 = N2 / dd  with  dd = (dir x e1) . e0,  N1 = (o - pa) . (dir x e1),  N2 = dir . ((o - pa) x e0),
// and rejects b1 < 0, b1 > 1, b2 < 0, b1 + b2 > 1.  The f32 evaluation here differs from the f64 values the reference computes by at most
//   |dd32 - dd| <= 80 e Mr Me^2,      |N1,2_32 - N1,2| <= 70 e (Mo + Dm) Mr Me,      e = 2^-24,
// with Mo = max |o_i|, Mr = max |dir_i|, Me = max |edge component| (Tri32::me, from the host), Dm = max |(o32 - pa)_i|:
//   inputs: o32, dir32 are off by e Mo, e Mr; the vertices are f32 values (exact); v = o32 - pa is off by e (Mo + Dm), an edge by e Me;
//   a cross-product component (two products, one sum; factors bounded by the M's) adds 3 e of its 2 M M' magnitude to the 2 (dM M' + M dM')
//   it inherits: s0 = dir x e1 within 10 e Mr Me, s1 = v x e0 within 10 e (Mo + Dm) Me;  a dot product of three terms adds 5 e of its 3 x
//   magnitude to the 3 (da |b| + |a| db) it inherits: the bounds above; the f64 roundings of the reference's own values are 2^-29 of these.
// Both bounds are taken with the factor 256 e = 2^-16 (more than three times what the analysis needs: it also covers the roundings of the bounds
// themselves and of the final comparisons, each below 6 e of the same magnitudes) and floored at 1e-30 (products of f32 denormals).  Decisions:
//   |dd32| <= Ed: the sign of dd is not known -> no verdict;      otherwise sign(dd) is that of dd32, and
//   |N1_32| > E and its sign differs from dd32's  =>  N1 / dd < 0: the reference's b1 = fl(N1 fl(1 / dd)) is negative      (reject b1 < 0)
//   |N2_32| > E and its sign differs               =>  b2 < 0;  it is reached only if b1 passed, and rejects                (reject b2 < 0)
//   |N1_32| - E > |dd32| + Ed  =>  |b1| > 1 by a relative margin above 2^-22 (the slack of the bounds): b1 > 1 or b1 < 0       (either rejects)
//   |N1_32 + N2_32| - 2 E > |dd32| + Ed  =>  |b1 + b2| > 1 likewise:  b1 + b2 > 1, or b1 + b2 < -1 and then b1 < 0 or b2 < 0 (all reject).
// NaN or infinite inputs fail every `>` (no verdict).  Magnitudes stay far inside f32's range: coordinates are below 2^36 (flatten.cpp).
DEV bool tri_miss32(float ox, float oy, float oz, float dx, float dy, float dz, float mo, float mrk, float pax, float pay, float paz, float pbx, float pby,
                    float pbz, float pcx, float pcy, float pcz, float me) {
    const float vx = ox - pax, vy = oy - pay, vz = oz - paz;
    const float e0x = pbx - pax, e0y = pby - pay, e0z = pbz - paz;
    const float e1x = pcx - pax, e1y = pcy - pay, e1z = pcz - paz;
    const float s0x = __builtin_fmaf(dy, e1z, -(dz * e1y)), s0y = __builtin_fmaf(dz, e1x, -(dx * e1z)), s0z = __builtin_fmaf(dx, e1y, -(dy * e1x));
    const float dd = __builtin_fmaf(s0z, e0z, __builtin_fmaf(s0y, e0y, s0x * e0x));
    const float n1 = __builtin_fmaf(vz, s0z, __builtin_fmaf(vy, s0y, vx * s0x));
    const float s1x = __builtin_fmaf(vy, e0z, -(vz * e0y)), s1y = __builtin_fmaf(vz, e0x, -(vx * e0z)), s1z = __builtin_fmaf(vx, e0y, -(vy * e0x));
    const float n2 = __builtin_fmaf(dz, s1z, __builtin_fmaf(dy, s1y, dx * s1x));
    const float dm = fmaxf(fmaxf(fabsf(vx), fabsf(vy)), fabsf(vz));
    const float mm = mrk * me;                          // 2^-16 Mr Me
    const float en = fmaxf((mo + dm) * mm, 1e-30f);     // >= |N1_32 - N1|, |N2_32 - N2|
    const float ed = fmaxf(mm * me, 1e-30f);            // >= |dd32 - dd|
    const float add = fabsf(dd), an1 = fabsf(n1), an2 = fabsf(n2);
    if (!(add > ed)) return false;
    const bool dneg = dd < 0.f;
    const bool out1 = an1 > en && ((n1 < 0.f) != dneg);
    const bool out2 = an2 > en && ((n2 < 0.f) != dneg);
    const float lim = add + ed;
    const bool big1 = an1 - en > lim;
    const bool big12 = fabsf(n1 + n2) - (en + en) > lim;
    return out1 || out2 || big1 || big12;
}
// MEASURED AND REJECTED (round 5, profiles/r05/c4_audit.md): with the pre-test C4 runs at 614 Msamples/s, without it at 847.  The f64 test is
// staged -- 30 VALU (nine of them the vertex conversions) up to its first reject, 22, 18 and 9 for the later stages -- and most candidates
// of a leaf leave at the first stage, so there are ~120 issue cycles to save per rejected triangle; the pre-test costs 50 VALU in five
// basic blocks (~150 cycles) for EVERY candidate.  Kept as an A/B switch, off.
#ifndef RT_TRI_F32_REJECT
#define RT_TRI_F32_REJECT 0
#endif

template <bool TIE>
DEV void blas_pass_q(const ServeCtx& X, bool act, AS_L uint32_t* stk, const int stride, D3 o, D3 d, double t_min, Ray32& r, double& ht, int& hnode,
                     uint32_t& hkp, uint32_t& cur, int& sp, bool& tied) {
    // (measured and dropped: speculative descent -- a leaf reached early is set aside while the lane walks on -- 805 against 924;
    // publishing a pass's answers one pass later, behind their stores' round trip, -1 %; ending the descent early once fewer than 16 / 24 / 32 / 40 lanes still descend -- 631 / 574 / 547 / 517
    // against 632 Msamples/s -- and testing at most 1 or 2 triangles of a leaf per pass, 543 / 612)
    const SignMasks sm = sign_masks(r);  // (the rays of a pass do not change: refills happen between passes)
    while (act && (cur >> REF_TAG_SHIFT) == 0u) {
        u32x4 u0, u1;  // (lox, loy, loz, hix) (hiy, hiz, c0, c1); child 0 in the low halves
        if (cur < X.n_topq) {
            const AS_L u32x4* p = X.n2q_top + 2 * cur;
            u0 = p[0]; u1 = p[1];
        } else {
            const AS_G u32x4* p = X.n2q + 2 * (size_t)cur;
            u0 = p[0]; u1 = p[1];
        }
        // near / far plane words per axis (both children at once), then the 16-bit halves
        const uint32_t nxw = sel32(u0.x, u0.w, sm.x), fxw = sel32(u0.w, u0.x, sm.x);
        const uint32_t nyw = sel32(u0.y, u1.x, sm.y), fyw = sel32(u1.x, u0.y, sm.y);
        const uint32_t nzw = sel32(u0.z, u1.y, sm.z), fzw = sel32(u1.y, u0.z, sm.z);
        float e0, e1;
        const bool h0 = box32s((float)(nxw & 0xffffu), (float)(nyw & 0xffffu), (float)(nzw & 0xffffu), (float)(fxw & 0xffffu), (float)(fyw & 0xffffu),
                               (float)(fzw & 0xffffu), r, e0);
        const bool h1 = box32s((float)(nxw >> 16), (float)(nyw >> 16), (float)(nzw >> 16), (float)(fxw >> 16), (float)(fyw >> 16), (float)(fzw >> 16), r, e1);
        const uint32_t c0 = u1.z, c1 = u1.w;
        if (h0 && h1) {
            const bool swap = e1 < e0;
            stk[sp] = swap ? c0 : c1;
            sp += stride;
            cur = swap ? c1 : c0;
        } else if (h0) {
            cur = c0;
        } else if (h1) {
            cur = c1;
        } else if (sp > 0) {
            sp -= stride;
            cur = stk[sp];
        } else {
            cur = REF_DONE;
        }
    }
    if (act && (cur >> REF_TAG_SHIFT) == 1u) {  // leaf: Triangle::hit in f64 (mesh.rs:57-102), tie rule as in traverse2
        const uint32_t first = cur & REF_LEAF_FIRST_MASK, cnt = ((cur >> REF_LEAF_COUNT_SHIFT) & 7u) + 1u;
        // the ray in f32 and its magnitudes, for tri_miss32 (per leaf: the lane's registers are worth more than the six conversions)
        const float ox32 = (float)o.x, oy32 = (float)o.y, oz32 = (float)o.z, dx32 = (float)d.x, dy32 = (float)d.y, dz32 = (float)d.z;
        const float mo32 = fmaxf(fmaxf(fabsf(ox32), fabsf(oy32)), fabsf(oz32));
        const float mrk32 = fmaxf(fmaxf(fabsf(dx32), fabsf(dy32)), fabsf(dz32)) * 1.52587890625e-05f;  // 2^-16 Mr
        for (uint32_t i = 0; i < cnt; i++) {
            const AS_G u32x4* p = X.tri32 + 3 * (size_t)(first + i);
            const u32x4 a = p[0], b = p[1], c = p[2];
            if (RT_TRI_F32_REJECT &&
                tri_miss32(ox32, oy32, oz32, dx32, dy32, dz32, mo32, mrk32, __uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w),
                           __uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w), __uint_as_float(c.x), __uint_as_float(c.w)))
                continue;  // Triangle::hit certainly returns None
            const D3 pa = mk((double)__uint_as_float(a.x), (double)__uint_as_float(a.y), (double)__uint_as_float(a.z));
            const D3 pb = mk((double)__uint_as_float(a.w), (double)__uint_as_float(b.x), (double)__uint_as_float(b.y));
            const D3 pc = mk((double)__uint_as_float(b.z), (double)__uint_as_float(b.w), (double)__uint_as_float(c.x));
            double t = 0., b1, b2;
            const bool got = tri_hit_v(pa, sub(pb, pa), sub(pc, pa), o, d, t_min, TIE ? ht * TIE_W : ht, t, b1, b2);  // (TIE: see TIE_W)
            if (got && (TIE ? (t < ht || (t == ht && (int)c.y > hnode) || !(t == t)) : (t < ht || (int)c.y > hnode || !(t == t)))) {
                if (TIE) tied = hnode >= 0 && t * TIE_W >= ht;
                ht = t;
                hnode = (int)c.y;
                hkp = c.z;
                r.best = ray32_best(t);
            } else if (TIE && got && (int)c.y < hnode) {  // within the tie margin of ht and the triangle is the earlier party
                tied = true;
            }
        }
        if (sp > 0) {
            sp -= stride;
            cur = stk[sp];
        } else {
            cur = REF_DONE;
        }
    }
}

// Out-of-line pieces of pt_kernel_coop get what they need from a block in LDS (written once per workgroup): a call passes
// arguments in VGPRs, these are wave-uniform and belong in SGPRs, and the pointers keep their address spaces this way.
struct CoopArgs {
    const char* base;  // scene blob (global)
    uint64_t* pool;    // this workgroup's parked-path records
    int* err;
    double t_min;
    uint32_t off_n2, off_items2, off_tripre2, off_spheres, off_rects, off_inst2, off_xforms, n_top;
    uint32_t lds_top, lds_coop;  // byte offsets of the Node2 cache and of the rings within the workgroup's LDS
    uint32_t lds_inst2, lds_xforms;  // ... and of the LDS copies of the instance table and the transforms
    uint32_t off_n2q, off_tri32, lds_qgrid, lds_topq, n_topq;  // compact object-space data; its grids and top nodes in LDS
};
#define AS_GLOBAL(T, p) ((T*)(__attribute__((address_space(1))) T*)(p))
#define AS_LDS(T, p) ((T*)(__attribute__((address_space(3))) T*)(p))
DEV uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
DEV uint64_t rfl64(uint64_t v) { return ((uint64_t)rfl((uint32_t)(v >> 32)) << 32) | (uint64_t)rfl((uint32_t)v); }
DEV CoopLds coop_rings(AS_L char* at) {  // [RQ | AQ | FQ: COOP_RING 16-bit entries each][8 counters]
    CoopLds C;
    AS_L uint16_t* cb = (AS_L uint16_t*)at;
    C.rq.buf = cb;
    C.aq.buf = cb + COOP_RING;
    C.fq.buf = cb + 2 * COOP_RING;
    AS_L uint32_t* cnt = (AS_L uint32_t*)(cb + 3 * COOP_RING);
    C.rq.ht = cnt;
    C.aq.ht = cnt + 2;
    C.fq.ht = cnt + 4;
    C.rq.abort_flag = C.aq.abort_flag = C.fq.abort_flag = cnt + 6;
    return C;
}
struct CoopCtx {
    Acc A;  // only the fields the object-space walk reads
    CoopLds C;
    uint64_t* pool;
    int* err;
    double t_min;
};
DEV CoopCtx coop_ctx(const CoopArgs* args_generic, char* lds0_generic) {  // lds0: start of the workgroup's dynamic LDS
    const CoopArgs* ga = AS_LDS(const CoopArgs, args_generic);
    char* lds0 = AS_LDS(char, lds0_generic);
    CoopCtx X;
    const char* base = AS_GLOBAL(const char, (const char*)rfl64((uint64_t)ga->base));
    X.pool = AS_GLOBAL(uint64_t, (uint64_t*)rfl64((uint64_t)ga->pool));
    X.err = AS_GLOBAL(int, (int*)rfl64((uint64_t)ga->err));
    X.t_min = __longlong_as_double((long long)rfl64((uint64_t)__double_as_longlong(ga->t_min)));
    X.A.n2 = (const float4*)(base + rfl(ga->off_n2));
    X.A.items2 = (const uint2*)(base + rfl(ga->off_items2));
    X.A.tripre2 = (const double2*)(base + rfl(ga->off_tripre2));
    X.A.spheres = (const double2*)(base + rfl(ga->off_spheres));
    X.A.rects = (const double2*)(base + rfl(ga->off_rects));
    X.A.inst2 = (const uint2*)(lds0 + rfl(ga->lds_inst2));
    X.A.xforms = (const double*)(lds0 + rfl(ga->lds_xforms));
    X.A.n2_top = (const AS_L f32x4*)(lds0 + rfl(ga->lds_top));
    X.A.n2_top_count = rfl(ga->n_top);
    X.A.n2q = (const uint4*)(base + rfl(ga->off_n2q));
    X.A.tri32 = (const uint4*)(base + rfl(ga->off_tri32));
    X.A.qgrid = (const double*)(lds0 + rfl(ga->lds_qgrid));
    X.A.n2q_top = (const uint4*)(lds0 + rfl(ga->lds_topq));
    X.A.n2q_top_count = rfl(ga->n_topq);
    X.C = coop_rings((AS_L char*)lds0_generic + rfl(ga->lds_coop));
    return X;
}
// the serving waves' context: the same block, every pointer typed with its address space
DEV ServeCtx serve_ctx(const CoopArgs* args_generic, char* lds0_generic) {
    const AS_L CoopArgs* ga = (const AS_L CoopArgs*)args_generic;
    AS_L char* lds0 = (AS_L char*)lds0_generic;
    ServeCtx X;
    const AS_G char* base = (const AS_G char*)rfl64((uint64_t)ga->base);
    X.pool = (AS_G U2*)rfl64((uint64_t)ga->pool);
    X.t_min = __longlong_as_double((long long)rfl64((uint64_t)__double_as_longlong(ga->t_min)));
    X.n2q = (const AS_G u32x4*)(base + rfl(ga->off_n2q));
    X.tri32 = (const AS_G u32x4*)(base + rfl(ga->off_tri32));
    X.n2q_top = (const AS_L u32x4*)(lds0 + rfl(ga->lds_topq));
    X.n_topq = rfl(ga->n_topq);
    X.qgrid = (const AS_L double*)(lds0 + rfl(ga->lds_qgrid));
    X.inst2 = (const AS_L uint32_t*)(lds0 + rfl(ga->lds_inst2));
    X.xforms = (const AS_L double*)(lds0 + rfl(ga->lds_xforms));
    X.C = coop_rings(lds0 + rfl(ga->lds_coop));
    return X;
}

// Serve the request ring with this wave: walk, refill, suspend the tail.  `may_suspend`: the wave has other work to go back to.
// Out of line: the caller's paths stay in callee-saved registers (saved once per call) instead of squeezing the walk's loop.
// TIE: the walks note exact ties (blas_pass_q); the bit travels in the order word of the answer / of a suspended walk (tie_order).
DEV uint32_t tie_order(int hnode, bool tied) { return (uint32_t)(tied ? hnode ^ TIE_FLAG : hnode); }
template <bool TIE>
__device__ __attribute__((noinline)) void coop_serve(const CoopArgs* args, char* lds0, uint32_t* stk_generic, bool may_suspend COOP_STATS_ARG) {
    const ServeCtx X = serve_ctx(args, lds0);
    const CoopLds& C = X.C;
    const double t_min = X.t_min;
    AS_L uint32_t* stk = (AS_L uint32_t*)stk_generic;
    const int stride = PT_BLOCK;
    const int lane = (int)(threadIdx.x & 63u);
    const uint64_t lanemask_lt = (1ull << lane) - 1ull;
    may_suspend = __ballot(may_suspend) != 0ull;
    int rid = -1;
    D3 o = mk(0, 0, 0), d = mk(0, 0, 1);
    Ray32 r = make_ray32(o, mk(1, 1, 1), t_min, 0.);
    double ht = 0.;
    int hnode = -1;
    uint32_t hkp = 0, cur = REF_DONE;
    int sp = 0;
    bool tied = false;
    for (;;) {
        // ---- refill: idle lanes take requests (fresh ones start at the instance's root, suspended ones where they stopped) ----
        const int got = ring_pop(C.rq, __ballot(rid < 0), lane, lanemask_lt);
        if (got >= 0) {
            rid = got;
            const AS_G U2* q = X.pool + (size_t)(COOP_REC / 2) * (size_t)rid;
            const U2 u0 = q[0], u1 = q[1], u2 = q[2], u3 = q[3], u4 = q[4];
            const uint32_t inst = (uint32_t)(u4.y >> 16) & 0xffu;
            const int n_saved = (int)((u4.y >> 24) & 0xffu);
            const AS_L double* m = X.xforms + 32 * X.inst2[2 * inst];  // M^-1: Transform::hit, transform.rs:153-156 (xf_point / xf_dir)
            const D3 wo = mk(bitsd(u0.x), bitsd(u0.y), bitsd(u1.x)), wd = mk(bitsd(u1.y), bitsd(u2.x), bitsd(u2.y));
            o = mk(m[0] * wo.x + m[1] * wo.y + m[2] * wo.z + m[3] * 1., m[4] * wo.x + m[5] * wo.y + m[6] * wo.z + m[7] * 1.,
                   m[8] * wo.x + m[9] * wo.y + m[10] * wo.z + m[11] * 1.);
            d = mk(m[0] * wd.x + m[1] * wd.y + m[2] * wd.z + m[3] * 0., m[4] * wd.x + m[5] * wd.y + m[6] * wd.z + m[7] * 0.,
                   m[8] * wd.x + m[9] * wd.y + m[10] * wd.z + m[11] * 0.);
            cur = (uint32_t)(u4.x >> 32);
            if (n_saved == 0) {  // fresh request
                ht = bitsd(u3.x);
                hnode = (int)(uint32_t)u3.y;
                hkp = 0u;
                sp = 0;
                tied = false;  // (a posted hit is never a pending tie: ties are settled before a path is parked or re-posted)
            } else {  // suspended walk: its best hit so far in unit 5; the stack comes back into this lane's LDS stack
                const U2 u5 = q[5];
                ht = bitsd(u5.x);
                hnode = (int)(uint32_t)u5.y;
                tied = TIE && tie_flagged(hnode);
                if (tied) hnode ^= TIE_FLAG;
                hkp = (uint32_t)(u5.y >> 32);
                const int n = n_saved - 1;
                for (int i = 0; i < n; i += 4) {
                    const U2 w = q[10 + (i >> 2)];
                    stk[i * stride] = (uint32_t)w.x;
                    if (i + 1 < n) stk[(i + 1) * stride] = (uint32_t)(w.x >> 32);
                    if (i + 2 < n) stk[(i + 2) * stride] = (uint32_t)w.y;
                    if (i + 3 < n) stk[(i + 3) * stride] = (uint32_t)(w.y >> 32);
                }
                sp = n * stride;
            }
            const AS_L double* g = X.qgrid + 8 * inst;  // the ray on the instance's grid: same t (QGrid, flat.h)
            const D3 og = mk((o.x - g[0]) * g[3] + g[6], (o.y - g[1]) * g[4] + g[6], (o.z - g[2]) * g[5] + g[6]);
            const D3 dg = mk(d.x * g[3], d.y * g[4], d.z * g[5]);
            r = make_ray32(og, dg, t_min, ht);
        }
        if (__ballot(rid >= 0) == 0ull) return;
        COOP_STAT(0, __ballot(rid >= 0));  // serve rounds: lanes holding a request at the start of a round
        bool thin = false;
        for (;;) {
            COOP_STAT(1, __ballot(rid >= 0));  // serve passes: busy lanes
            blas_pass_q<TIE>(X, rid >= 0, stk, stride, o, d, t_min, r, ht, hnode, hkp, cur, sp, tied);
            const bool fin = rid >= 0 && cur == REF_DONE;
            if (fin) {
                U2 ans;
                ans.x = dbits(ht);
                ans.y = ((uint64_t)hkp << 32) | (uint64_t)tie_order(hnode, TIE && tied);
                X.pool[(size_t)(COOP_REC / 2) * (size_t)rid + 5] = ans;
            }
            ring_push(C.aq, fin, (uint32_t)rid, lane, lanemask_lt, true);
            if (fin) rid = -1;
            const int busy = __popcll(__ballot(rid >= 0));
            if (busy == 0) break;
            if (busy < COOP_REFILL_TH) {
                if (ring_len(C.rq) != 0u) break;  // more requests wait: refill the idle lanes
                if (may_suspend && busy < COOP_SUSPEND_TH) {
                    thin = true;
                    break;
                }
            }
        }
        if (thin) {  // hand the remaining walks back: a denser batch will continue them
            const bool sus = rid >= 0;
            COOP_STAT(4, __ballot(sus));
            if (sus) {
                AS_G U2* q = X.pool + (size_t)(COOP_REC / 2) * (size_t)rid;
                const int n = sp / stride;
                const U2 u4 = q[4];
                U2 w4, w5;
                w4.x = ((uint64_t)cur << 32) | (uint64_t)(uint32_t)u4.x;
                w4.y = (u4.y & ~(0xffull << 24)) | ((uint64_t)(n + 1) << 24);
                w5.x = dbits(ht);
                w5.y = ((uint64_t)hkp << 32) | (uint64_t)tie_order(hnode, TIE && tied);
                q[4] = w4;
                q[5] = w5;
                for (int i = 0; i < n; i += 4) {
                    const uint64_t e0 = stk[i * stride], e1 = (i + 1 < n) ? stk[(i + 1) * stride] : 0u;
                    const uint64_t e2 = (i + 2 < n) ? stk[(i + 2) * stride] : 0u, e3 = (i + 3 < n) ? stk[(i + 3) * stride] : 0u;
                    U2 w;
                    w.x = e0 | (e1 << 32);
                    w.y = e2 | (e3 << 32);
                    q[10 + (i >> 2)] = w;
                }
            }
            ring_push(C.rq, sus, (uint32_t)rid, lane, lanemask_lt, true);
            return;
        }
    }
}

// A path whose deferred instances cannot be parked (pool exhausted; rare): walk them here, as the plain kernel would.
// TIE: an exact tie noted by these walks comes back as TIE_FLAG in the hit's xf; the caller settles it.
template <bool TIE>
__device__ __attribute__((noinline)) Hit coop_walk_inline(const CoopArgs* args, char* lds0, uint32_t* stk_generic, D3 o, D3 d, Hit h, uint64_t pend) {
    const CoopCtx X = coop_ctx(args, lds0);
    const Acc& A = X.A;
    uint32_t* stk = AS_LDS(uint32_t, stk_generic);
    bool tied = false;
    while (pend != 0ull) {
        const uint32_t ni = (uint32_t)(__ffsll((long long)pend) - 1);
        pend &= pend - 1ull;
        const uint2 in = A.inst2[ni];
        const double* Minv = A.xforms + 32 * in.x;
        const D3 oo = xf_point(Minv, o), dd = xf_dir(Minv, d);
        const double a = sqlen(dd);
        Ray32 r = make_ray32(oo, dd, X.t_min, h.t);
        double ht = h.t;
        int hnode = h.node, sp = 0;
        uint32_t hkp = 0u, cur = in.y;
        while (cur != REF_DONE) blas_pass<TIE>(A, true, stk, PT_BLOCK, oo, dd, a, X.t_min, r, ht, hnode, hkp, cur, sp, X.err, tied);
        if (hnode != h.node) {
            h.t = ht;
            h.node = hnode;
            h.kp = hkp;
            h.xf = (int)in.x;
        }
    }
    if (TIE && tied) h.xf ^= TIE_FLAG;
    return h;
}

// PEND: the type of the per-path mask of deferred instances: uint32_t for scenes with up to 32 instances, uint64_t for 33..64 (the wider
// mask costs the 32-instance scenes 5 % in registers: C4 837 instead of 883 Msamples/s, so it is a variant, not the default)
// TIE: the exact-tie rule of the reference ("EXACT ties" above) -- for scenes with rectangles or cubes; its own variant because the flag
// in the world-space walk and the resolver's call sites cost the others registers (C4 -6 % when it was a build switch, round 4)
template <int INTEG, bool MIXED = false, bool EARLY = false, typename PEND = uint32_t, bool TIE = false>
__global__ void __launch_bounds__(PT_BLOCK) pt_kernel_coop(FlatView sv, CamK cam, RenderK rk, double* __restrict__ ring, double* accum,
                                                           unsigned int* tickets, unsigned int* __restrict__ counter, int* __restrict__ err, uint64_t* coop) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS map: [world-level tables][top-of-BVH Node2 cache][stacks: stack2 x PT_BLOCK words][ring / job bookkeeping as in pt_kernel]
    //          [RQ, AQ, FQ: COOP_RING 16-bit entries each][8 counters][CoopArgs]     (the scene itself stays in L2 / HBM)
    uint32_t staged = 0;
    Acc A = make_acc(sv.base, sv.base, sv);
    // world-level tables into LDS (coop_world_bytes, flat.h); A then is the view the world-space walk and the shading use
    uint32_t lds_inst2 = 0, lds_xforms = 0, lds_qgrid = 0;
    const double* qgrid_lds = nullptr;
    {
        auto stage = [&](uint32_t off, uint32_t bytes) {
            const uint4* src = (const uint4*)(sv.base + off);
            uint4* dst = (uint4*)(smem + staged);
            for (uint32_t i = threadIdx.x; i < coop_a16(bytes) / 16; i += blockDim.x) dst[i] = src[i];
            const char* at = smem + staged;
            staged += coop_a16(bytes);
            return at;
        };
        A.spheres = (const double2*)stage(sv.off_spheres, sv.off_rects - sv.off_spheres);
        A.rects = (const double2*)stage(sv.off_rects, sv.off_tris - sv.off_rects);
        lds_xforms = staged;
        A.xforms = (const double*)stage(sv.off_xforms, sv.stage_bytes - sv.off_xforms);
        lds_inst2 = staged;
        A.inst2 = (const uint2*)stage(sv.off_inst2, 8u * sv.n_inst2);
        A.items2 = (const uint2*)stage(sv.off_items2, 8u * sv.n_world_items2);  // world leaves only: instances are deferred
        lds_qgrid = staged;
        qgrid_lds = (const double*)stage(sv.off_qgrid, (uint32_t)sizeof(QGrid) * sv.n_inst2);
    }
    const uint32_t lds_top = staged;
    if (rk.n_top > 0) {
        const uint4* src = (const uint4*)(sv.base + sv.off_n2);
        uint4* dst = (uint4*)(smem + staged);
        for (uint32_t i = threadIdx.x; i < (uint32_t)rk.n_top * NODE2_F4; i += blockDim.x) dst[i] = src[i];
        A.n2_top = (const AS_L f32x4*)(smem + staged);
        A.n2_top_count = (uint32_t)rk.n_top;
        staged += (uint32_t)rk.n_top * (uint32_t)sizeof(Node2);
    }
    const uint32_t lds_topq = staged;  // the shallowest NodeQ of every object-space BVH: for the serving waves and for the entry walk
    const uint4* n2q_lds = (const uint4*)(smem + staged);
    // the entry walk stays in the first COOP_ENTRY_NODES cached nodes (measured: 16 / 64 nodes 870, 256 866, 1024 860, all cached 856 Msamples/s)
    const uint32_t entry_top = min((uint32_t)rk.n_topq, (uint32_t)COOP_ENTRY_NODES);
    {
        const uint4* src = (const uint4*)(sv.base + sv.off_n2q);
        uint4* dst = (uint4*)(smem + staged);
        for (uint32_t i = threadIdx.x; i < (uint32_t)rk.n_topq * 2u; i += blockDim.x) dst[i] = src[i];
        staged += (uint32_t)rk.n_topq * (uint32_t)sizeof(NodeQ);
    }
    uint32_t* stk = (uint32_t*)(smem + staged) + threadIdx.x;
    const int stk_stride = (int)blockDim.x;
    const int lane = threadIdx.x & 63;
    const uint64_t lanemask_lt = (1ull << lane) - 1ull;
    const int wave = threadIdx.x >> 6;
    uint32_t* book = (uint32_t*)(smem + staged) + (size_t)rk.coop_stack * PT_BLOCK;  // stacks: max(world depth, object-space depth) + 2 entries
    uint32_t* rmeta = book + (size_t)wave * RING_UNITS * 4;
    uint32_t* wst = book + (size_t)(PT_BLOCK / 64) * RING_UNITS * 4 + (size_t)wave * 8;
    int* cfg = (int*)(book + (size_t)(PT_BLOCK / 64) * (RING_UNITS * 4 + 8));
    char* coop_base = (char*)(cfg + CFG_WORDS);
    const CoopLds C = coop_rings((AS_L char*)coop_base);
    AS_L uint32_t* cnt = C.rq.ht;
    CoopArgs* cargs = (CoopArgs*)(coop_base + 3 * COOP_RING * sizeof(uint16_t) + 8 * sizeof(uint32_t));
    for (uint32_t i = threadIdx.x; i < (uint32_t)COOP_RING; i += blockDim.x) {
        C.rq.buf[i] = (uint16_t)0;
        C.aq.buf[i] = (uint16_t)0;
        C.fq.buf[i] = (uint16_t)(i < (uint32_t)rk.coop_pool ? i + 1u : 0u);  // every pool slot in use is free
    }
    if (threadIdx.x < 8) cnt[threadIdx.x] = (threadIdx.x == 5) ? (uint32_t)rk.coop_pool : 0u;  // FQ tail = number of slots
    if (lane < 8) wst[lane] = (lane == 6) ? 1u : 0u;
    if (EARLY && lane < RING_UNITS * 4) rmeta[lane] = 0u;  // no slot in use (next_unit_pool)
    if (threadIdx.x == 0) {
        cfg[CFG_N_JOBS] = rk.n_units; cfg[CFG_TILES_OWNED] = rk.tiles_owned; cfg[CFG_JOB_UNITS] = rk.job_units;
        cfg[CFG_SUBS_PER_TILE] = rk.subs_per_tile; cfg[CFG_WORLD] = rk.world; cfg[CFG_RANK] = rk.rank; cfg[CFG_TILES_X] = rk.tiles_x;
        cfg[CFG_S_BEGIN] = rk.s_begin; cfg[CFG_S_END] = rk.s_end; cfg[CFG_SUB_SPP] = rk.sub_spp; cfg[CFG_WIDTH] = rk.width;
        cfg[CFG_HEIGHT] = rk.height;
        cfg[CFG_RING_UNITS] = RING_UNITS;
        cfg[CFG_JOB_LISTS] = (int)JOB_LISTS;
#pragma unroll
        for (int i = 0; i < 25; i++) cfg[CFG_LVL + i] = rk.lvl[i / 5][i % 5];
        cargs->base = sv.base;
        cargs->pool = coop + (size_t)blockIdx.x * COOP_POOL * COOP_REC;
        cargs->err = err;
        cargs->t_min = rk.t_min;
        cargs->off_n2 = sv.off_n2; cargs->off_items2 = sv.off_items2; cargs->off_tripre2 = sv.off_tripre2;
        cargs->off_spheres = sv.off_spheres; cargs->off_rects = sv.off_rects; cargs->off_inst2 = sv.off_inst2;
        cargs->off_xforms = sv.off_xforms;
        cargs->n_top = (uint32_t)rk.n_top;
        cargs->lds_top = lds_top;
        cargs->lds_inst2 = lds_inst2;
        cargs->lds_xforms = lds_xforms;
        cargs->off_n2q = sv.off_n2q; cargs->off_tri32 = sv.off_tri32; cargs->lds_qgrid = lds_qgrid;
        cargs->lds_topq = lds_topq; cargs->n_topq = (uint32_t)rk.n_topq;
        cargs->lds_coop = (uint32_t)((char*)coop_base - smem);
    }
    __syncthreads();
    double* bring = ring + (size_t)blockIdx.x * (PT_BLOCK / 64) * RING_UNITS * UNIT_DOUBLES;  // the workgroup's rings, wave-major
    double* wring = bring + (size_t)wave * RING_UNITS * UNIT_DOUBLES;
    uint64_t* pool_mem = coop + (size_t)blockIdx.x * COOP_POOL * COOP_REC;

    int tx = 0, ty = 0, s0 = 0, pool = 0, next = 0, cur_slot = 0;
    bool finished = false;

    bool alive = false;   // the lane holds a path
    bool ready = false;   // ... whose closest hit is known (shade next); otherwise it needs its world-space walk
    D3 o = mk(0, 0, 0), d = mk(0, 0, 1), beta = mk(1, 1, 1), L = mk(0, 0, 0);
    int depth = 0, pix_id = 0;
    uint32_t out_slot = 0;  // owner wave << 12 | (ring slot * UNIT_SPP + sample within the unit) * 64 + pixel
    PEND pend = 0;  // deferred instances of the current segment, one bit each
    constexpr bool wide_pend = sizeof(PEND) == 8;  // the upper half travels in the record's last unit
    int dec_slot = -1;      // a finished path whose unit counter still has to be decremented (owner wave << 12 | slot index)
    int fold_wait = 0;      // iterations until the head of the ring is looked at again (wave-uniform)
    Rng rng;
    rng.s = 0;
    Hit h;
    h.t = INFINITY; h.node = -1; h.xf = -1; h.kp = 0;
#ifdef RTAMD_COOP_STATS
    CoopStats cs;
    cs.t_last = __builtin_amdgcn_s_memtime();
#endif

    for (;;) {
        // ---- unit counters of the paths finished last iteration: their sample stores have completed by now ----
        if (__ballot(dec_slot >= 0) != 0ull) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (dec_slot >= 0) {
                atomicSub(book + (size_t)(dec_slot >> 12) * RING_UNITS * 4 + 4 * ((dec_slot & 0xfff) >> 9) + 3, 1u);
                dec_slot = -1;
            }
        }
        // ---- free lanes generate new paths right before the walk ----
        uint64_t dead = __ballot(!alive);
        if ((int)__popcll(dead) < REGEN_MIN && dead != ~0ull) dead = 0ull;
        {
            // EARLY: as in pt_kernel, a complete unit at the head of the ring is folded at once through the same call (take = false).  A variant of
            // its own because the check costs this kernel 3 % on a whole frame (C4: 409 -> 423 ms at 256 spp) and pays only for a rank's share
            // (1/8 of the frame: 64 -> 56..63 ms, 1/16: 31.1 -> 29.0); render_tiles picks it when the rank owns fewer tiles than 2 x waves
            const bool need_unit = dead != 0ull && next >= pool && !finished;
            bool head_ready = false;
            if (EARLY && !need_unit && --fold_wait < 0) {
                fold_wait = FOLD_PERIOD - 1;
                head_ready = __builtin_amdgcn_readfirstlane(wst[1]) != 0u &&
                             __builtin_amdgcn_readfirstlane(__hip_atomic_load(&rmeta[4 * __builtin_amdgcn_readfirstlane(wst[0]) + 3], __ATOMIC_RELAXED,
                                                                              __HIP_MEMORY_SCOPE_WORKGROUP)) == 0u;
            }
            if (need_unit || head_ready) {
                const UnitInfo u = (EARLY && rk.pool_mode != 0) ? next_unit_pool(wst, rmeta, cfg, wring, accum, tickets, counter, false, lane, need_unit)
                                                                : next_unit(wst, rmeta, cfg, wring, accum, tickets, counter, false, lane, need_unit);
                if (need_unit) {
                    if (u.pool > 0) {
                        pool = u.pool;
                        next = 0;
                        s0 = u.s0;
                        tx = u.tx;
                        ty = u.ty;
                        cur_slot = u.cur_slot;
                    }
                    finished = u.finished != 0;
                } else if (__builtin_amdgcn_readfirstlane(u.cur_slot) == 0) {
                    fold_wait = FOLD_RETRY;
                }
            }
        }
        if (dead != 0ull && next < pool) {
            int k = next + __popcll(dead & lanemask_lt);
            next = min(next + (int)__popcll(dead), pool);
            if (!alive && k < pool) {
                int pix = k & (TILE_PIX - 1), s = s0 + (k >> 6);
                int x = tx * TILE_W + (pix & (TILE_W - 1)), y = ty * TILE_H + (pix >> 3);
                if (x < rk.width && y < rk.height) {  // camera.rs:97-99 + Camera::get_ray camera.rs:57-64
                    rng.seed_stream(rk.seed, (uint64_t)y * (uint64_t)rk.width + (uint64_t)x, (uint64_t)s);
                    double u = ((double)x + rng.gen_f64()) / (double)(rk.width - 1);
                    double v = ((double)y + rng.gen_f64()) / (double)(rk.height - 1);
                    double stt = 1.0 - v;
                    D3 rd = muls(random_in_unit_disk(rng), cam.lens_radius);
                    D3 offset = add(muls(cam.u, rd.x), muls(cam.v, rd.y));
                    o = add(cam.origin, offset);
                    d = sub(sub(add(add(cam.llc, muls(cam.horizontal, u)), muls(cam.vertical, stt)), cam.origin), offset);
                    beta = mk(1., 1., 1.);
                    L = mk(0., 0., 0.);
                    depth = rk.max_depth;
                    pix_id = y * rk.width + x;
                    out_slot = ((uint32_t)wave << 12) | (((uint32_t)cur_slot * (uint32_t)UNIT_SPP + (uint32_t)(k >> 6)) * (uint32_t)TILE_PIX + (uint32_t)pix);
                    alive = true;
                    ready = false;
                    pend = (PEND)0;
                }
            }
        }
        COOP_TIME(0);
        // ---- world-space walk of the lanes that start a segment, instances deferred ----
        if (__ballot(alive && !ready && pend == (PEND)0) != 0ull) COOP_STAT(2, __ballot(alive && !ready && pend == (PEND)0));
        if (alive && !ready && pend == (PEND)0) {  // (pend != 0: a path between two deferred instances of one segment)
            h = traverse2<true, true, true, false, PEND, false, MIXED, false, false, TIE>(A, stk, stk_stride, o, d, rk.t_min, INFINITY, &pend);
            // two world-level objects share the best t (TIE_FLAG in h.xf): the reference-order walk will answer for the whole ray, instances
            // included -- right before the shading, the one call site of this kernel
            if (TIE && tie_flagged(h.xf)) pend = (PEND)0;
            if (pend == (PEND)0) ready = true;
        }
        COOP_TIME(1);
        // ---- enter the first deferred instance of each such path: the top of its BVH is in LDS (NodeQ cache), so the lane walks
        //      it here; most rays that only clip the instance's box end there and never leave the lane.  Where the walk needs a
        //      node or a leaf from memory it stops, and what is left of it (node + stack) is parked as a suspended walk ----
        {
            const bool want = alive && !ready;  // (pend != 0)
            const uint64_t mw = __ballot(want);
            if (mw != 0ull) {
                uint32_t ent_cur = REF_DONE, ni = 0u;
                int ent_sp = 0;
                if (want) {
                    ni = (uint32_t)(__ffsll((long long)(uint64_t)pend) - 1);
                    const double* Minv = A.xforms + 32 * A.inst2[ni].x;
                    const D3 oo = xf_point(Minv, o), dd = xf_dir(Minv, d);
                    const double* g = qgrid_lds + 8 * ni;
                    const D3 og = mk((oo.x - g[0]) * g[3] + g[6], (oo.y - g[1]) * g[4] + g[6], (oo.z - g[2]) * g[5] + g[6]);
                    const D3 dg = mk(dd.x * g[3], dd.y * g[4], dd.z * g[5]);
                    const Ray32 r = make_ray32(og, dg, rk.t_min, h.t);
                    const SignMasks sm = sign_masks(r);
                    ent_cur = A.inst2[ni].y;
                    while ((ent_cur >> REF_TAG_SHIFT) == 0u && ent_cur < entry_top) {
                        const uint4* p = n2q_lds + 2 * ent_cur;
                        const uint4 u0 = p[0], u1 = p[1];
                        const uint32_t nxw = sel32(u0.x, u0.w, sm.x), fxw = sel32(u0.w, u0.x, sm.x);
                        const uint32_t nyw = sel32(u0.y, u1.x, sm.y), fyw = sel32(u1.x, u0.y, sm.y);
                        const uint32_t nzw = sel32(u0.z, u1.y, sm.z), fzw = sel32(u1.y, u0.z, sm.z);
                        float e0, e1;
                        const bool h0 = box32s((float)(nxw & 0xffffu), (float)(nyw & 0xffffu), (float)(nzw & 0xffffu), (float)(fxw & 0xffffu),
                                               (float)(fyw & 0xffffu), (float)(fzw & 0xffffu), r, e0);
                        const bool h1 = box32s((float)(nxw >> 16), (float)(nyw >> 16), (float)(nzw >> 16), (float)(fxw >> 16), (float)(fyw >> 16),
                                               (float)(fzw >> 16), r, e1);
                        const uint32_t c0 = u1.z, c1 = u1.w;
                        if (h0 && h1) {
                            const bool swap = e1 < e0;
                            stk[ent_sp] = swap ? c0 : c1;
                            ent_sp += stk_stride;
                            ent_cur = swap ? c1 : c0;
                        } else if (h0) {
                            ent_cur = c0;
                        } else if (h1) {
                            ent_cur = c1;
                        } else if (ent_sp > 0) {
                            ent_sp -= stk_stride;
                            ent_cur = stk[ent_sp];
                        } else {
                            ent_cur = REF_DONE;
                        }
                    }
                    if (ent_cur == REF_DONE) {  // nothing of this instance within reach: next one (next iteration), or shade
                        pend &= pend - (PEND)1;
                        if (pend == (PEND)0) ready = true;
                    }
                }
                const bool need = want && ent_cur != REF_DONE;
                COOP_RING_T0;
                const int id = ring_pop(C.fq, __ballot(need), lane, lanemask_lt);
                COOP_RING_T1;
                const bool park = need && id >= 0;
                if (park) {
                    pend &= pend - (PEND)1;
                    const int n = ent_sp / stk_stride;
                    uint64_t* q = pool_mem + (size_t)COOP_REC * (size_t)id;
                    st_unit(q, 0, dbits(o.x), dbits(o.y));
                    st_unit(q, 1, dbits(o.z), dbits(d.x));
                    st_unit(q, 2, dbits(d.y), dbits(d.z));
                    st_unit(q, 3, dbits(h.t), ((uint64_t)h.kp << 32) | (uint64_t)(uint32_t)h.node);
                    st_unit(q, 4, ((uint64_t)ent_cur << 32) | (uint64_t)(uint32_t)(h.xf + 1),
                            ((uint64_t)pend << 32) | ((uint64_t)(n + 1) << 24) | (uint64_t)(ni << 16) | (uint64_t)out_slot);  // a suspended walk
                    if (wide_pend) st_unit(q, 10 + COOP_STACK_MAX / 4, (uint64_t)pend >> 32, 0ull);
                    st_unit(q, 5, dbits(h.t), (uint64_t)(uint32_t)h.node);  // its best so far: what the world-space walk found
                    st_unit(q, 6, dbits(beta.x), dbits(beta.y));
                    st_unit(q, 7, dbits(beta.z), dbits(L.x));
                    st_unit(q, 8, dbits(L.y), dbits(L.z));
                    st_unit(q, 9, rng.s, ((uint64_t)(uint32_t)depth << 32) | (uint64_t)(uint32_t)pix_id);
                    for (int i = 0; i < n; i += 4) {
                        const uint64_t e0 = stk[i * stk_stride], e1 = (i + 1 < n) ? stk[(i + 1) * stk_stride] : 0u;
                        const uint64_t e2 = (i + 2 < n) ? stk[(i + 2) * stk_stride] : 0u, e3 = (i + 3 < n) ? stk[(i + 3) * stk_stride] : 0u;
                        st_unit(q, 10 + (i >> 2), e0 | (e1 << 32), e2 | (e3 << 32));
                    }
                    alive = false;
                }
                ring_push(C.rq, park, (uint32_t)id, lane, lanemask_lt, true);  // (publishing an iteration later, behind the stores' round trip: no gain)
                // pool exhausted (rare): walk the deferred instances in this lane, sparsely, as the plain kernel does
                if (__ballot(need && id < 0) != 0ull) {
                    if (need && id < 0) {
                        h = coop_walk_inline<TIE>(cargs, smem, stk, o, d, h, pend);  // (TIE: a tie comes back as TIE_FLAG in h.xf)
                        pend = (PEND)0;
                        ready = true;
                    }
                }
            }
        }
        COOP_TIME(2);
        // ---- free lanes adopt answered paths right before shading ----
        {
            uint64_t fr = __ballot(!alive);
            if ((int)__popcll(fr) < COOP_ADOPT_MIN && fr != ~0ull) fr = 0ull;
            if (fr != 0ull && ring_len(C.aq) != 0u) {
                COOP_RING_T0;
                const int id = ring_pop(C.aq, fr, lane, lanemask_lt);
                COOP_RING_T1;
                COOP_STAT(6, __ballot(id >= 0));
                bool repost = false, freed = false, hit_inside = false;
                if (id >= 0) {
                    uint64_t* q = pool_mem + (size_t)COOP_REC * (size_t)id;
                    // the whole record in one round trip (a re-posted path wastes the second half; rare: several instances on one ray)
                    const U2 u3 = ld_unit(q, 3), u4 = ld_unit(q, 4), u5 = ld_unit(q, 5);
                    const U2 u0 = ld_unit(q, 0), u1 = ld_unit(q, 1), u2 = ld_unit(q, 2);
                    const U2 u6 = ld_unit(q, 6), u7 = ld_unit(q, 7), u8 = ld_unit(q, 8), u9 = ld_unit(q, 9);
                    const uint32_t inst = (uint32_t)(u4.y >> 16) & 0xffu;
                    h.t = bitsd(u3.x);  // best hit before this instance: as posted
                    h.node = (int)(uint32_t)u3.y;
                    h.kp = (uint32_t)(u3.y >> 32);
                    h.xf = (int)(uint32_t)u4.x - 1;
                    int a_node = (int)(uint32_t)u5.y;
                    const bool a_tied = TIE && tie_flagged(a_node);  // the walk met a triangle at exactly the best t (tie_order)
                    if (a_tied) a_node ^= TIE_FLAG;
                    if (a_node != h.node) {  // the walk accepted a candidate of this instance
                        hit_inside = true;
                        h.t = bitsd(u5.x);
                        h.node = a_node;
                        h.kp = (uint32_t)(u5.y >> 32);
                        h.xf = (int)A.inst2[inst].x;
                    }
                    pend = (PEND)(u4.y >> 32);
                    if (wide_pend) pend |= (PEND)(ld_unit(q, 10 + COOP_STACK_MAX / 4).x << 32);
                    if (a_tied) pend = (PEND)0;  // the reference-order walk below answers for the whole ray
                    out_slot = (uint32_t)u4.y & 0xffffu;
                    if (pend != (PEND)0) {  // next deferred instance of the same segment: the path stays parked, new request
                        const uint32_t ni = (uint32_t)(__ffsll((long long)(uint64_t)pend) - 1);
                        pend &= pend - (PEND)1;
                        st_unit(q, 3, dbits(h.t), ((uint64_t)h.kp << 32) | (uint64_t)(uint32_t)h.node);
                        st_unit(q, 4, ((uint64_t)A.inst2[ni].y << 32) | (uint64_t)(uint32_t)(h.xf + 1),
                                ((uint64_t)pend << 32) | (uint64_t)(ni << 16) | (uint64_t)out_slot);
                        if (wide_pend) st_unit(q, 10 + COOP_STACK_MAX / 4, (uint64_t)pend >> 32, 0ull);
                        repost = true;
                    } else {
                        o = mk(bitsd(u0.x), bitsd(u0.y), bitsd(u1.x));
                        d = mk(bitsd(u1.y), bitsd(u2.x), bitsd(u2.y));
                        beta = mk(bitsd(u6.x), bitsd(u6.y), bitsd(u7.x));
                        L = mk(bitsd(u7.y), bitsd(u8.x), bitsd(u8.y));
                        rng.s = u9.x;
                        depth = (int)(uint32_t)(u9.y >> 32);
                        pix_id = (int)(uint32_t)u9.y;
                        alive = true;
                        ready = true;
                        freed = true;
                        if (a_tied) h.xf ^= TIE_FLAG;  // settled right before the shading
                    }
                }
                COOP_STAT(7, __ballot(hit_inside));
                (void)hit_inside;
                ring_push(C.rq, repost, (uint32_t)id, lane, lanemask_lt, true);
                {
                    const uint64_t mf = __ballot(freed);
                    if (mf != 0ull) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the slot's loads have returned before it can be reused
                    }
                    ring_push(C.fq, freed, (uint32_t)id, lane, lanemask_lt, false);
                }
            }
        }
        COOP_TIME(3);
        // ---- shade: sample_ray's loop body after World::hit, photon_mapper.rs:336-362 ----
        if (__ballot(alive && ready) != 0ull) COOP_STAT(3, __ballot(alive && ready));
        if (alive && ready) {
            bool done = true;
            // an exact or near tie was noted on this segment (by the world-space walk, an object-space walk or the in-lane fallback): the
            // reference's own walk decides ("EXACT ties" above traverse<>)
            if (TIE && tie_flagged(h.xf)) h = tie_resolve_view(A.tie_view, o.x, o.y, o.z, d.x, d.y, d.z, rk.t_min, INFINITY);
            if (h.node >= 0 && depth > 0) {
                depth -= 1;
                Rec rec = materialize<true>(A, h, o, d, err);
                D3 emitted, att, ndir;
                bool diffuse;
                bool scattered = shade(A, rec, d, rng, emitted, att, ndir, diffuse, err);
                L = add(L, elemul(beta, emitted));
                if (scattered) {
                    bool go = true;
                    if (INTEG == 2 && diffuse) {
                        const double* e = rk.sppm_est + 6 * (size_t)pix_id;
                        L = add(L, elemul(beta, mk(e[0], e[1], e[2])));
                        L = add(L, elemul(beta, mk(e[3], e[4], e[5])));
                        go = false;
                    } else if (INTEG == 1 && diffuse) {
                        go = mixture_step(A, rec, rng, att, beta, ndir, err);
                    } else {
                        beta = elemul(beta, att);
                    }
                    if (go) {
                        o = rec.p;
                        d = ndir;
                        done = false;
                    }
                }
            }
            ready = false;
            pend = (PEND)0;
            if (done) {  // into the ring slot of the wave that generated the path; its unit counter moves next iteration
                double* dst = bring + (size_t)(out_slot >> 12) * RING_UNITS * UNIT_DOUBLES + 3 * (size_t)(out_slot & 0xfffu);
                dst[0] = L.x;
                dst[1] = L.y;
                dst[2] = L.z;
                dec_slot = (int)out_slot;
                alive = false;
            }
        }
        COOP_TIME(4);
        // ---- serve: a wave's worth of requests waits ----
        if (ring_len(C.rq) >= min((uint32_t)COOP_BATCH, (uint32_t)rk.coop_pool / 4u + 1u)) coop_serve<TIE>(cargs, smem, stk, true COOP_STATS_PASS);
        COOP_TIME(5);
        // ---- nothing in the lanes, nothing to adopt, no path to generate: fold / fetch, else serve whatever waits, else leave ----
        if (__ballot(alive) == 0ull && next >= pool && ring_len(C.aq) == 0u && __ballot(dec_slot >= 0) == 0ull) {
            bool got_unit = false;
            if (!finished) {  // units still running somewhere (parked, or adopted by other waves)
                const UnitInfo u = (EARLY && rk.pool_mode != 0) ? next_unit_pool(wst, rmeta, cfg, wring, accum, tickets, counter, false, lane)
                                                                : next_unit(wst, rmeta, cfg, wring, accum, tickets, counter, false, lane);
                if (u.pool > 0) {
                    pool = u.pool;
                    next = 0;
                    s0 = u.s0;
                    tx = u.tx;
                    ty = u.ty;
                    cur_slot = u.cur_slot;
                    got_unit = true;
                }
                finished = u.finished != 0;
            }
            if (!got_unit) {
                if (ring_len(C.rq) != 0u) {
                    coop_serve<TIE>(cargs, smem, stk, false COOP_STATS_PASS);
                } else if (finished && ring_len(C.fq) == (uint32_t)rk.coop_pool) {  // every pool slot is free again: no path is parked
                    break;
                } else if (lds_load(C.rq.abort_flag) != 0u) {  // a ring overran (ring_pop): the frame is lost, leave instead of hanging
                    if (lane == 0) atomicOr(err, 4);
                    break;
                } else {
                    COOP_STAT(5, 0ull);
                    __builtin_amdgcn_s_sleep(8);  // other waves hold what this one waits for
                }
            }
        }
        COOP_TIME(6);
    }
#ifdef RTAMD_COOP_STATS
    if (lane == 0) {
        for (int i = 0; i < 8; i++) {
            atomicAdd(&g_coop_stats[2 * i], cs.ev[i]);
            atomicAdd(&g_coop_stats[2 * i + 1], cs.ln[i]);
        }
        for (int i = 0; i < 8; i++) atomicAdd(&g_coop_time[i], cs.tm[i]);
    }
#endif
}

// pixel_color /= spp (camera.rs:102); pixels of edge tiles that fall outside the image are zeroed
__global__ void finalize_kernel(const double* __restrict__ accum, double* __restrict__ tiles, int64_t n_pix, int spp, int width,
                                int height, int tiles_x, int rank, int world) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pix) return;
    int64_t lt = i >> 6;
    int pix = (int)(i & 63);
    int64_t tile = lt * world + rank;
    int x = (int)(tile % tiles_x) * TILE_W + (pix & 7), y = (int)(tile / tiles_x) * TILE_H + (pix >> 3);
    bool inside = x < width && y < height;
    double n = (double)spp;
    tiles[3 * i] = inside ? accum[3 * i] / n : 0.;
    tiles[3 * i + 1] = inside ? accum[3 * i + 1] / n : 0.;
    tiles[3 * i + 2] = inside ? accum[3 * i + 2] / n : 0.;
}
// the stitch, camera.rs:115-123: gathered[rank][local tile][pix][3] -> frame[y][x][3]
__global__ void assemble_kernel(const double* __restrict__ gathered, int64_t stride_tiles, double* __restrict__ frame, int width,
                                int height, int tiles_x, int world) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t n = (int64_t)width * height;
    if (i >= n) return;
    int x = (int)(i % width), y = (int)(i / width);
    int64_t tile = (int64_t)(y >> 3) * tiles_x + (x >> 3);
    int r = (int)(tile % world);
    int64_t lt = tile / world;
    int pix = ((y & 7) << 3) | (x & 7);
    const double* src = gathered + (((int64_t)r * stride_tiles + lt) * TILE_PIX + pix) * 3;
    frame[3 * i] = src[0];
    frame[3 * i + 1] = src[1];
    frame[3 * i + 2] = src[2];
}

#include "wavefront.inc"
#include "sppm.inc"

// ------------------------------------------------------- debug kernels ----
__global__ void rng_kernel(uint64_t seed, uint64_t pixel, uint64_t sample, int n, uint64_t* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Rng r;
    r.seed_stream(seed, pixel, sample);
    for (int i = 0; i < n; i++) out[i] = r.next_u64();
}
__global__ void rng_floats_kernel(uint64_t seed, uint64_t pixel, uint64_t sample, int n, double lo, double hi, double* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Rng r;
    r.seed_stream(seed, pixel, sample);
    for (int i = 0; i < n; i++) out[i] = r.gen_f64();
    r.seed_stream(seed, pixel, sample);
    for (int i = 0; i < n; i++) out[n + i] = (lo == -1. && hi == 1.) ? r.gen_range_pm1() : (lo == 0. && hi == 1.) ? r.gen_range_01() : r.gen_range(lo, hi);  // (the forms the samplers call)
}
__global__ void math_kernel(int op, size_t n, const double* a, const double* b, double* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = (op == 0) ? sqrt(a[i]) : (op == 2) ? det_ln(a[i]) : (op == 3) ? det_sin(a[i]) : a[i] / b[i];
}
__global__ void hit_kernel(FlatView sv, int accel, size_t n, const double* rays, double t_min, double t_max, double* out, int* err) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Acc A = make_acc(sv.base, sv.base, sv);
    uint32_t stack_at = 0;
    if (accel == 3) {  // the NodeW table of pt_kernel's LDS variants (box32w), expanded here as that kernel expands it; tables stay global
        for (uint32_t k = threadIdx.x; k < sv.n_nodes2; k += blockDim.x) {
            const uint4* nd = (const uint4*)(sv.base + sv.off_n2) + (size_t)k * NODE2_F4;
            const uint4 a = nd[0], b = nd[1], c = nd[2], d4 = nd[3];
            const uint32_t c0 = wide_ref(d4.x), c1 = wide_ref(d4.y);
            char* w = smem + wide_ref(k);
            const uint4 lo0 = make_uint4(a.x, a.y, a.z, a.w), lo1 = make_uint4(b.x, b.y, c0, c1);
            const uint4 hi0 = make_uint4(b.z, b.w, c.x, c.y), hi1 = make_uint4(c.z, c.w, c0, c1);
            ((uint4*)w)[0] = lo0; ((uint4*)w)[1] = lo1;
            ((uint4*)(w + NODEW_FAR))[0] = hi0; ((uint4*)(w + NODEW_FAR))[1] = hi1;
            ((uint4*)(w + 2 * NODEW_FAR))[0] = lo0; ((uint4*)(w + 2 * NODEW_FAR))[1] = lo1;
        }
        __syncthreads();
        A.n2w_lds = (uint32_t)(uintptr_t)(AS_L char*)smem;
        stack_at = nodew_bytes(sv.n_nodes2);
    }
    if (i >= n) return;
    D3 o = mk(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]), d = mk(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
    uint32_t* stk = (uint32_t*)(smem + stack_at) + threadIdx.x;
    Hit h;
    if (accel == 5 || accel == 6) {
        // The walks of kernels 5 / 6 in one lane: the world-space walk with the large instances DEFERRED and exact ties noted (TIEDEF), then
        // every deferred instance's object-space walk from that result -- 5: over Node2 / items / hoisted triangle records (coop_walk_inline's
        // blas_pass), 6: over the compact NodeQ / Tri32 copies (the serving waves' blas_pass_q) -- with their tie bit, and the
        // reference-order re-walk when a flag turns up.  (The parking, serving and adopting around these walks is pt_kernel_coop's.)
        uint64_t pend = 0ull;
        h = traverse2<true, true, false, false, uint64_t, false, true, false, false, true>(A, stk, (int)blockDim.x, o, d, t_min, t_max, &pend);
        if (tie_flagged(h.xf)) {
            h = tie_resolve<1>(A, o, d, t_min, t_max, h);
            pend = 0ull;
        }
        bool tied = false;
        ServeCtx X;
        X.n2q = (const AS_G u32x4*)(sv.base + sv.off_n2q);
        X.tri32 = (const AS_G u32x4*)(sv.base + sv.off_tri32);
        X.n2q_top = nullptr;
        X.n_topq = 0u;
        while (pend != 0ull) {
            const uint32_t ni = (uint32_t)(__ffsll((long long)pend) - 1);
            pend &= pend - 1ull;
            const uint2 in = A.inst2[ni];
            const double* Minv = A.xforms + 32 * in.x;
            const D3 oo = xf_point(Minv, o), dd = xf_dir(Minv, d);
            double ht = h.t;
            int hnode = h.node, sp = 0;
            uint32_t hkp = 0u, cur = in.y;
            if (accel == 5) {
                const double a = sqlen(dd);
                Ray32 r = make_ray32(oo, dd, t_min, h.t);
                while (cur != REF_DONE) blas_pass<true>(A, true, stk, (int)blockDim.x, oo, dd, a, t_min, r, ht, hnode, hkp, cur, sp, err, tied);
            } else {
                const double* g = (const double*)(sv.base + sv.off_qgrid) + 8 * ni;  // the ray on the instance's grid: same t (QGrid, flat.h)
                const D3 og = mk((oo.x - g[0]) * g[3] + g[6], (oo.y - g[1]) * g[4] + g[6], (oo.z - g[2]) * g[5] + g[6]);
                const D3 dg = mk(dd.x * g[3], dd.y * g[4], dd.z * g[5]);
                Ray32 r = make_ray32(og, dg, t_min, h.t);
                while (cur != REF_DONE) blas_pass_q<true>(X, true, (AS_L uint32_t*)stk, (int)blockDim.x, oo, dd, t_min, r, ht, hnode, hkp, cur, sp, tied);
            }
            if (hnode != h.node) {
                h.t = ht;
                h.node = hnode;
                h.kp = hkp;
                h.xf = (int)in.x;
            }
        }
        if (tied) h = tie_resolve<1>(A, o, d, t_min, t_max, h);
    } else {
        h = (accel == 3) ? traverse2<true, false, false, true>(A, stk, (int)blockDim.x, o, d, t_min, t_max)
          : (accel == 2) ? traverse2<true, false, false>(A, stk, (int)blockDim.x, o, d, t_min, t_max)
                         : traverse<true>(A, o, d, t_min, t_max);
    }
    double* q = out + 12 * i;
    for (int k = 0; k < 12; k++) q[k] = 0.;
    if (h.node < 0) return;
    Rec rec = materialize<true>(A, h, o, d, err);
    // uv is materialised lazily (only for image textures); recompute it here for the diagnostic
    q[0] = 1.;
    q[1] = h.t;
    q[2] = rec.p.x; q[3] = rec.p.y; q[4] = rec.p.z;
    q[5] = rec.normal.x; q[6] = rec.normal.y; q[7] = rec.normal.z;
    q[8] = rec.front_face ? 1. : 0.;
    q[9] = rec.u; q[10] = rec.v;
    q[11] = (double)h.node;
}

// ------------------------------------------------------------ host side ---
int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void* dev_alloc(size_t n) {
    void* p = nullptr;
    HIP_CHECK(hipMalloc(&p, n ? n : 16));
    return p;
}
void dev_free(void* p) { (void)hipFree(p); }
void dev_copy_to_host(void* dst, const void* src, size_t n) { HIP_CHECK(hipMemcpy(dst, src, n, hipMemcpyDeviceToHost)); }
void dev_copy_to_device(void* dst, const void* src, size_t n) { HIP_CHECK(hipMemcpy(dst, src, n, hipMemcpyHostToDevice)); }
void dev_set_device(int d) { HIP_CHECK(hipSetDevice(d)); }

static const char* device_blob(const rt_scene& s, int dev, double* upload_ms = nullptr) {
    std::lock_guard<std::mutex> g(s.dev_mu);
    for (auto& c : s.dev)
        if (c.device == dev) return (const char*)c.d_blob;
    const auto t0 = std::chrono::steady_clock::now();
    struct Timed {  // the one upload of this scene to this device (rt_stats.upload_ms of the call that made it)
        double* out;
        std::chrono::steady_clock::time_point t0;
        ~Timed() {
            if (out) *out = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
    } timed{upload_ms, t0};
    DeviceCopy c;
    c.device = dev;
    HIP_CHECK(hipMalloc(&c.d_blob, s.flat.blob.size()));
    HIP_CHECK(hipMemcpy(c.d_blob, s.flat.blob.data(), s.flat.blob.size(), hipMemcpyHostToDevice));
    s.dev.push_back(c);
    return (const char*)c.d_blob;
}
void free_device_copies(rt_scene& s) {
    std::lock_guard<std::mutex> g(s.dev_mu);
    for (auto& c : s.dev) {
        int cur = 0;
        if (hipGetDevice(&cur) == hipSuccess) {
            (void)hipSetDevice(c.device);
            (void)hipFree(c.d_blob);
            (void)hipSetDevice(cur);
        }
    }
    s.dev.clear();
}

static CamK to_camk(const CameraDev& c) {
    CamK k;
    k.origin = D3{c.origin[0], c.origin[1], c.origin[2]};
    k.llc = D3{c.llc[0], c.llc[1], c.llc[2]};
    k.horizontal = D3{c.horizontal[0], c.horizontal[1], c.horizontal[2]};
    k.vertical = D3{c.vertical[0], c.vertical[1], c.vertical[2]};
    k.u = D3{c.u[0], c.u[1], c.u[2]};
    k.v = D3{c.v[0], c.v[1], c.v[2]};
    k.lens_radius = c.lens_radius;
    return k;
}

struct DevBuf {
    void* p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { reset(); }
    void reset() {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    void alloc(size_t n) {
        reset();
        HIP_CHECK(hipMalloc(&p, n ? n : 16));
    }
};
struct Events {
    std::vector<hipEvent_t> ev;
    ~Events() {
        for (auto e : ev) (void)hipEventDestroy(e);
    }
    hipEvent_t make() {
        hipEvent_t e;
        HIP_CHECK(hipEventCreate(&e));
        ev.push_back(e);
        return e;
    }
};

// ---- per-device caches: properties and render workspaces (no hipMalloc/hipFree/property queries per call) ----
struct DevInfo {
    int cus = 0;
    size_t lds_max = 64 * 1024;
};
static std::mutex g_mu;
static std::map<int, DevInfo> g_devinfo;
static const DevInfo& dev_info(int dev) {
    std::lock_guard<std::mutex> g(g_mu);
    auto it = g_devinfo.find(dev);
    if (it != g_devinfo.end()) return it->second;
    DevInfo di;
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    di.cus = prop.multiProcessorCount;
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && v > 0) di.lds_max = (size_t)v;
    if (di.lds_max > 160 * 1024) di.lds_max = 160 * 1024;
#ifdef RTAMD_PHASE_STATS
    di.lds_max -= 1024;  // the statistics build keeps 48 counters in static LDS
#endif
    return g_devinfo.emplace(dev, di).first->second;
}
// A workspace = the waves' unit rings, the accumulator, the per-tile tickets and a small block {work counter, error flag}.
// Kept for the life of the process and reused by later calls on the same device (grown when a call needs more);
// rt_release_workspaces frees the idle ones.
struct Workspace {
    int device = -1;
    bool busy = false;
    void *ring = nullptr, *accum = nullptr, *tickets = nullptr, *small = nullptr, *coop = nullptr;
    size_t ring_bytes = 0, accum_bytes = 0, ticket_bytes = 0, coop_bytes = 0;
};
static std::vector<Workspace*> g_ws;
static const size_t WS_SMALL = 2048;
static void grow(void*& p, size_t& have, size_t need) {
    if (have >= need) return;
    if (p) (void)hipFree(p);
    p = nullptr;
    have = 0;
    HIP_CHECK(hipMalloc(&p, need));
    have = need;
}
struct WorkspaceLease {
    Workspace* w = nullptr;
    WorkspaceLease(int dev, size_t need_ring, size_t need_accum, size_t need_tickets, size_t need_coop = 0) {
        {
            std::lock_guard<std::mutex> g(g_mu);
            for (Workspace* c : g_ws)
                if (c->device == dev && !c->busy) {
                    w = c;
                    break;
                }
            if (!w) {
                w = new Workspace();
                w->device = dev;
                g_ws.push_back(w);
            }
            w->busy = true;
        }
        try {
            if (!w->small) HIP_CHECK(hipMalloc(&w->small, WS_SMALL));
            grow(w->ring, w->ring_bytes, need_ring);
            grow(w->accum, w->accum_bytes, need_accum);
            grow(w->tickets, w->ticket_bytes, need_tickets);
            if (need_coop) grow(w->coop, w->coop_bytes, need_coop);
        } catch (...) {
            std::lock_guard<std::mutex> g(g_mu);
            w->busy = false;
            throw;
        }
    }
    ~WorkspaceLease() {
        std::lock_guard<std::mutex> g(g_mu);
        w->busy = false;
    }
};
size_t release_workspaces() {  // frees every idle workspace (all devices); returns the bytes released
    std::lock_guard<std::mutex> g(g_mu);
    size_t freed = 0;
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (auto it = g_ws.begin(); it != g_ws.end();) {
        Workspace* w = *it;
        if (w->busy) {
            ++it;
            continue;
        }
        (void)hipSetDevice(w->device);
        for (void* p : {w->ring, w->accum, w->tickets, w->small, w->coop})
            if (p) (void)hipFree(p);
        freed += w->ring_bytes + w->accum_bytes + w->ticket_bytes + w->coop_bytes + WS_SMALL;
        delete w;
        it = g_ws.erase(it);
    }
    if (have_cur) (void)hipSetDevice(cur);
    return freed;
}

typedef void (*pt_fn)(FlatView, CamK, RenderK, double*, double*, unsigned int*, unsigned int*, int*);
typedef void (*pt_coop_fn)(FlatView, CamK, RenderK, double*, double*, unsigned int*, unsigned int*, int*, uint64_t*);

template <int ACCEL>
static pt_fn pick_pt_kernel(bool lds, bool general, int integ) {
    if (integ == 1) return lds ? pt_kernel<true, true, ACCEL, 1> : pt_kernel<false, true, ACCEL, 1>;  // mixture / SPPM: general primitive set only
    if (integ == 2) return lds ? pt_kernel<true, true, ACCEL, 2> : pt_kernel<false, true, ACCEL, 2>;
    return lds ? (general ? pt_kernel<true, true, ACCEL, 0> : pt_kernel<true, false, ACCEL, 0>)
               : (general ? pt_kernel<false, true, ACCEL, 0> : pt_kernel<false, false, ACCEL, 0>);
}
// the POOL variants (out-of-order fold for a rank that owns fewer tiles than the GPU has waves) exist for kernel 2
static pt_fn pick_pt_kernel_pool(bool lds, bool general, int integ) {
    if (integ == 1) return lds ? pt_kernel<true, true, 2, 1, false, true> : pt_kernel<false, true, 2, 1, false, true>;
    if (integ == 2) return lds ? pt_kernel<true, true, 2, 2, false, true> : pt_kernel<false, true, 2, 2, false, true>;
    return lds ? (general ? pt_kernel<true, true, 2, 0, false, true> : pt_kernel<true, false, 2, 0, false, true>)
               : (general ? pt_kernel<false, true, 2, 0, false, true> : pt_kernel<false, false, 2, 0, false, true>);
}

static void render_tiles_wf(const rt_scene& s, const FlatView& view, const CameraDev& cam, const RenderPlan& plan, const Tuning& tun, double* d_tiles,
                            hipStream_t stream, rt_stats* st, int dev, const DevInfo& di, uint32_t stack6, uint32_t n_entry6, size_t lds_pt, uint32_t stack6w,
                            size_t tables_w);

// Kernels 5 / 6 run their TIE variants ("EXACT ties" above traverse<>) for every scene that holds a rectangle or a cube: only those have
// boxes that can BEGIN exactly where another surface lies (bvh.rs:88 + aabb.rs:28-30).  Scenes of spheres and triangle meshes alone keep the
// variants without the flag (as the sphere-only variants of kernels 1 / 2 carry no tie code).
static bool tie_scene(const FlatView& v) {
    return TIE_RULE != 0 && (v.kinds_mask & ((1u << NK_RECT_YZ) | (1u << NK_RECT_XZ) | (1u << NK_RECT_XY) | (1u << NK_CUBE))) != 0u;
}
template <bool TIE>
static void pick_coop(int integ, bool mixed, bool wide, pt_coop_fn& fn, pt_coop_fn& fn_early) {
    if (wide) {
        fn_early = (integ == 1) ? pt_kernel_coop<1, true, true, uint64_t, TIE> : (integ == 2) ? pt_kernel_coop<2, true, true, uint64_t, TIE> : pt_kernel_coop<0, true, true, uint64_t, TIE>;
        fn = (integ == 1) ? pt_kernel_coop<1, true, false, uint64_t, TIE> : (integ == 2) ? pt_kernel_coop<2, true, false, uint64_t, TIE> : pt_kernel_coop<0, true, false, uint64_t, TIE>;
    } else if (mixed) {
        fn_early = (integ == 1) ? pt_kernel_coop<1, true, true, uint32_t, TIE> : (integ == 2) ? pt_kernel_coop<2, true, true, uint32_t, TIE> : pt_kernel_coop<0, true, true, uint32_t, TIE>;
        fn = (integ == 1) ? pt_kernel_coop<1, true, false, uint32_t, TIE> : (integ == 2) ? pt_kernel_coop<2, true, false, uint32_t, TIE> : pt_kernel_coop<0, true, false, uint32_t, TIE>;
    } else {
        fn_early = (integ == 1) ? pt_kernel_coop<1, false, true, uint32_t, TIE> : (integ == 2) ? pt_kernel_coop<2, false, true, uint32_t, TIE> : pt_kernel_coop<0, false, true, uint32_t, TIE>;
        fn = (integ == 1) ? pt_kernel_coop<1, false, false, uint32_t, TIE> : (integ == 2) ? pt_kernel_coop<2, false, false, uint32_t, TIE> : pt_kernel_coop<0, false, false, uint32_t, TIE>;
    }
}

void render_tiles(const rt_scene& s, const CameraDev& cam, const RenderPlan& plan_in, double* d_tiles, void* stream_, rt_stats* st) {
    if (!s.committed) throw RtError(RT_ERR_NOT_COMMITTED, "scene not committed");
    const Tuning tun = tuning();  // one snapshot per call
    const RenderPlan& plan = plan_in;
    hipStream_t stream = (hipStream_t)stream_;
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    const DevInfo& di = dev_info(dev);

    FlatView view = s.flat.view;
    double upload_ms = 0.;
    view.base = device_blob(s, dev, &upload_ms);
    if (st) st->upload_ms = upload_ms;
    const bool moving = (view.kinds_mask & (1u << NK_MSPHERE)) != 0;  // D9: the paths' times live in LDS, 8 bytes per lane
    // D9: moving spheres, noise textures and an open shutter (every sample draws a time) live in their own kernel variants (GENERAL == 2,
    // kernels 1 / 2, integrator 0): nothing of them is compiled into the others
    const bool book2 = moving || view.has_noise != 0 || plan.time1 > plan.time0;
    if (moving) {  // a moving sphere's box covers its positions between ITS time0 and time1 only (scene.cpp: add_moving_sphere)
        const double sh0 = plan.time0, sh1 = plan.time1 > plan.time0 ? plan.time1 : plan.time0;
        if (!(sh0 >= s.flat.msph_t0_max && sh1 <= s.flat.msph_t1_min))
            throw RtError(RT_ERR_ARG, "the shutter [time0, time1] = [" + std::to_string(sh0) + ", " + std::to_string(sh1) + "] must lie inside [time0, time1] of every moving sphere ([" +
                                          std::to_string(s.flat.msph_t0_max) + ", " + std::to_string(s.flat.msph_t1_min) + "] for this scene): their boxes are built for that range");
    }
    const bool general = (view.kinds_mask & ~((1u << NK_BOX) | (1u << NK_SPHERE))) != 0 || book2;
    const size_t lds_max = di.lds_max - (moving ? (size_t)PT_BLOCK * sizeof(double) : 0);
    // The accel kernels need the camera inside the region the f32 boxes were padded for (flatten.cpp: origin_limit2) and
    // t_min >= 0 (box32's proof); otherwise kernel 1 (reference order) renders.
    double cam_abs = std::fmax(std::fmax(std::fabs(cam.origin[0]), std::fabs(cam.origin[1])), std::fabs(cam.origin[2])) + std::fabs(cam.lens_radius);
    const bool camera_ok = cam_abs <= view.origin_limit2 && std::isfinite(cam_abs) && plan.t_min >= 0.;
    const size_t stack2_bytes = (size_t)view.stack2 * PT_BLOCK * sizeof(uint32_t);
    const size_t hot1 = (size_t)view.stage_bytes, hot2 = (size_t)(view.stage2_end - view.stage2_begin);
    const bool accel2_usable = view.accel_ok && camera_ok && stack2_bytes <= lds_max;  // per-lane stacks live in LDS
    const bool media = (view.kinds_mask & (1u << NK_MEDIUM_BEGIN)) != 0;               // kernel 1, or kernel 2's MEDIA variant (traverse2_media)
    // kernel 5 = kernel 2's BVH with the cooperative instance service (pt_kernel_coop): for scenes with LARGE mesh instances
    // the two walks of kernel 5 never share a stack; the world-space walk includes the instances it enters in the lane (NK_INSTANCE_INLINE)
    const uint32_t stack5 = std::max(std::max(view.world_depth2, view.inst_depth2) + 2u, view.stack2_inline);
    const size_t stack5_bytes = (size_t)stack5 * PT_BLOCK * sizeof(uint32_t);
    const size_t coop_world = coop_world_bytes(view);  // world-level tables, always in LDS for this kernel
    const size_t coop_lds = (size_t)3 * COOP_RING * sizeof(uint16_t) + 8 * sizeof(uint32_t) + ((sizeof(CoopArgs) + 15) & ~size_t(15)) + coop_world;  // + three rings of pool-slot ids, counters, argument block
    const bool coop_usable = accel2_usable && general && !media && !book2 && view.coop_data_ok != 0 && view.n_inst2 >= 1 && view.n_inst2 <= (uint32_t)COOP_MAX_INST &&
                             view.inst_depth2 <= (uint32_t)COOP_STACK_MAX && coop_world <= 32768 && stack5_bytes + coop_lds <= lds_max && plan.max_depth < (1 << 24);
    // kernel 6 = the same instance service across the whole GPU and across launches (wavefront.inc)
    const uint32_t stack6 = std::max<uint32_t>(std::max(view.world_depth2 + 2u, view.stack2_inline), (uint32_t)WF_ENTRY_STACK + 1u);
    const uint32_t n_entry6 = std::min<uint32_t>((uint32_t)COOP_ENTRY_NODES, view.n_nodes2);
    const size_t wf_lds_pt = coop_world + (size_t)std::min<uint32_t>(128u, view.world_top2) * sizeof(Node2) + (size_t)n_entry6 * sizeof(NodeQ) +
                             (size_t)stack6 * PT_BLOCK * sizeof(uint32_t) + ((size_t)WF_BOOK_WORDS + CFG_WORDS + 8) * sizeof(uint32_t);
    const uint32_t stack6w = view.inst_depth2 + 2u;
    const size_t wf_tables_w = coop_a16(view.stage_bytes - view.off_xforms) + coop_a16(8u * view.n_inst2) + coop_a16((uint32_t)sizeof(QGrid) * view.n_inst2);
    const size_t wf_lds_walk_min = wf_tables_w + (size_t)stack6w * WF_WALK_BLOCK * sizeof(uint32_t);
    const bool wf_usable = accel2_usable && general && !media && !book2 && view.coop_data_ok != 0 && view.n_inst2 >= 1 && view.n_inst2 <= (uint32_t)WF_MAX_INST &&
                           coop_world <= 32768 && wf_lds_pt <= lds_max && wf_lds_walk_min <= lds_max && plan.max_depth < (1 << 24);
    int kernel = plan.kernel;
    // auto: the cooperative kernel as soon as an instance is more than a handful of triangles (Cornell box + torus instance, 64 spp,
    // kernel 5 / kernel 2 in Msamples/s: 120 triangles 1351 / 1182 (kernel 2 LDS-resident), 1 600: 1263 / 822, 25 600: 1035 / 517,
    // 102 400: 861 / 427, 409 600: 746 / 370; the 12-triangle cube of the reference's Cornell box: 2507 / 2533)
    // (kernel 6, the wavefront form of the same service, reaches 766 Msamples/s on C4 where kernel 5 reaches 891 and kernel 2 469: it is
    // since round 4 kernel 5 takes up to 64 instances too, so kernel 6 runs by request only)
    if (kernel == 0)
        kernel = accel2_usable ? ((coop_usable && view.max_inst_nodes2 >= 64u) ? 5 : (wf_usable && !coop_usable && view.max_inst_nodes2 >= 64u) ? 6 : 2) : 1;
    if (kernel == 5 && !coop_usable)
        throw RtError(RT_ERR_UNSUPPORTED, "kernel 5 (cooperative instance service) needs a usable accel, 1..64 instances of which at least one holds only triangles with f32 vertices (an OBJ mesh), of BVH depth <= 40");
    if (kernel == 6 && !wf_usable)
        throw RtError(RT_ERR_UNSUPPORTED, "kernel 6 (wavefront instance service) needs a usable accel and 1..64 instances of which at least one holds only triangles with f32 vertices (an OBJ mesh)");
    if ((kernel == 2 || kernel == 5) && !accel2_usable)
        throw RtError(RT_ERR_UNSUPPORTED, "kernel 2 requested but no usable accel for this scene/camera (unbounded item, depth overflow, stacks "
                                          "larger than LDS, negative t_min, or camera farther than 64x the scene extent); use kernel 0/1");
    const size_t ring_meta = ((size_t)(PT_BLOCK / 64) * (RING_UNITS * 4 + 8) + CFG_WORDS) * sizeof(uint32_t);  // ring / job bookkeeping, behind the stacks
    const size_t stack_bytes = ((kernel == 2) ? stack2_bytes : (kernel == 5) ? stack5_bytes : 0) + ring_meta + ((kernel == 5) ? coop_lds : 0);
    if (plan.ext_accum && (kernel == 6 || plan.integrator == 2))
        throw RtError(RT_ERR_UNSUPPORTED, "resumable rendering (rt_render_accumulate_device) runs with kernels 1 / 2 / 5 and integrators 0 / 1");
    if (kernel == 6) {
        render_tiles_wf(s, view, cam, plan, tun, d_tiles, stream, st, dev, di, stack6, n_entry6, wf_lds_pt, stack6w, wf_tables_w);
        return;
    }
    // kernel 2 expands the Node2 array (64 B per node) into the NodeW form (96 B) while staging it
    const size_t nodew = ((size_t)(view.n_nodes2 + NODEW_CHUNK - 1) / NODEW_CHUNK) * 3 * NODEW_FAR;
    const size_t hot_bytes = (kernel == 2 || kernel == 5) ? hot2 - (size_t)view.n_nodes2 * sizeof(Node2) + nodew : hot1;
    const bool lds = hot_bytes > 0 && hot_bytes + stack_bytes <= lds_max && !tun.no_lds && kernel != 5;  // kernel 5: scene in L2/HBM always
    const int integ = plan.integrator;
    if (integ == 1 && view.n_lights == 0) throw RtError(RT_ERR_ARG, "integrator 1 (light importance sampling) needs rt_scene_set_lights");
    if (integ == 2 && !plan.sppm_est) throw RtError(RT_ERR_ARG, "integrator 2 (SPPM) is reached through rt_render_sppm");
    if (integ != 0 && book2)
        throw RtError(RT_ERR_UNSUPPORTED, "the book-2 extensions (moving spheres, noise textures, an open shutter) render with integrator 0 (kernels 1 and 2)");
    if (media && integ != 0)
        throw RtError(RT_ERR_UNSUPPORTED, "scenes with a ConstantMedium render with integrator 0 only (the medium's random draw is part of the "
                                          "reference-order walk; light sampling and SPPM have no volume events)");
    pt_fn fn = (kernel == 1) ? pick_pt_kernel<1>(lds, general, integ) : pick_pt_kernel<2>(lds, general, integ);
    pt_coop_fn fn_coop = nullptr;
    pt_coop_fn fn_coop_early = nullptr;  // the variant that folds from the main loop: for a rank that owns few tiles (see pt_kernel_coop)
    if (kernel == 5) {
        // 33..64 instances: the 64-bit pending mask (the MIXED code covers scenes without inline instances too); exact ties: see tie_scene()
        const bool wide = view.n_inst2 > 32u, mixed = view.n_inline2 != 0u;
        if (tie_scene(view)) pick_coop<true>(integ, mixed, wide, fn_coop, fn_coop_early);
        else pick_coop<false>(integ, mixed, wide, fn_coop, fn_coop_early);
    }
    if (media)
        fn = (kernel == 2) ? (lds ? pt_kernel<true, true, 2, 0, true> : pt_kernel<false, true, 2, 0, true>)
                           : (lds ? pt_kernel<true, true, 1, 0, true> : pt_kernel<false, true, 1, 0, true>);
    if (book2)  // GENERAL == 2
        fn = media ? ((kernel == 2) ? (lds ? pt_kernel<true, 2, 2, 0, true> : pt_kernel<false, 2, 2, 0, true>)
                                    : (lds ? pt_kernel<true, 2, 1, 0, true> : pt_kernel<false, 2, 1, 0, true>))
                   : ((kernel == 2) ? (lds ? pt_kernel<true, 2, 2, 0> : pt_kernel<false, 2, 2, 0>) : (lds ? pt_kernel<true, 2, 1, 0> : pt_kernel<false, 2, 1, 0>));
    // scene too large for LDS: spend what is left after the stacks on the shallowest BVH levels (the Node2 array is depth-sorted)
    int n_top = 0, n_topq = 0;
    if (kernel == 2 && !lds && lds_max > stack_bytes) {
        size_t room = (lds_max - stack_bytes) / sizeof(Node2);
        if (tun.n_top >= 0) room = std::min<size_t>(room, (size_t)tun.n_top);
        n_top = (int)std::min<size_t>(room, view.n_nodes2);
    }
    if (kernel == 5 && lds_max > stack_bytes) {  // f32 nodes for the world-space walk (at most 128), the rest of LDS for the compact object-space nodes
        size_t room = lds_max - stack_bytes;
        n_top = (int)std::min<size_t>(std::min<size_t>(room / sizeof(Node2), 128), view.world_top2);
        if (tun.n_top >= 0) n_top = std::min(n_top, tun.n_top);
        room -= (size_t)n_top * sizeof(Node2);
        n_topq = (int)std::min<size_t>(room / sizeof(NodeQ), view.n_nodes2);
        if (tun.n_top >= 0) n_topq = std::min(n_topq, tun.n_top);
    }
    const size_t smem = (lds ? hot_bytes : (size_t)n_top * sizeof(Node2) + (size_t)n_topq * sizeof(NodeQ)) + stack_bytes + (moving ? (size_t)PT_BLOCK * sizeof(double) : 0);
    if (kernel == 5 && COOP_EARLY_FOLD && plan.tiles_owned < (int64_t)2 * di.cus * (PT_BLOCK / 64)) fn_coop = fn_coop_early;  // (one workgroup per CU)
    const void* fptr = (kernel == 5) ? (const void*)fn_coop : (const void*)fn;
    if (smem > 48 * 1024) HIP_CHECK(hipFuncSetAttribute(fptr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    int blocks_per_cu = 0;
    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, fptr, PT_BLOCK, smem));
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    const int grid = di.cus * blocks_per_cu;
    // a rank that owns fewer tiles than the launch has waves: single-unit jobs (below) folded out of order (next_unit_pool)
    const bool pool = POOL_MODE && SINGLE_UNITS_BELOW_WAVES && plan.tiles_owned < (int64_t)grid * (PT_BLOCK / 64) &&
                      ((kernel == 2 && !media && !book2) || (kernel == 5 && fn_coop == fn_coop_early));
    if (pool && kernel == 2) {
        fn = pick_pt_kernel_pool(lds, general, integ);  // (same resources as the variant the occupancy was asked for)
        if (smem > 48 * 1024) HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    }

    const int64_t n_pix = plan.tiles_owned * TILE_PIX;
    // workspace: RING_UNITS unit buffers (12 KB) per resident wave -- independent of the image -- plus accumulator and tickets
    const size_t ring_bytes = (size_t)grid * (PT_BLOCK / 64) * RING_UNITS * UNIT_DOUBLES * sizeof(double);
    WorkspaceLease lease(dev, ring_bytes, std::max<size_t>(16, (size_t)n_pix * 3 * sizeof(double)),
                         std::max<size_t>(16, (size_t)plan.tiles_owned * sizeof(unsigned int)),
                         (kernel == 5) ? (size_t)grid * COOP_POOL * COOP_REC * sizeof(uint64_t) : 0);
    struct Ptr {
        void* p;
    };
    char* small = (char*)lease.w->small;
    Ptr ringp{lease.w->ring}, accum{plan.ext_accum ? (void*)plan.ext_accum : lease.w->accum}, tickets{lease.w->tickets}, counter{small}, err{small + 16};
    const int s_first = plan.ext_accum ? plan.s_first : 0, s_last = (plan.ext_accum && plan.s_last >= 0) ? plan.s_last : plan.spp;
    HIP_CHECK(hipMemsetAsync(small, 0, WS_SMALL, stream));
    Events events;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pt_ev;
    CamK ck = to_camk(cam);
    int launches = 0;
    for (int s0 = s_first; s0 < s_last; s0 += plan.spp_chunk) {
        const int s1 = std::min(s0 + plan.spp_chunk, s_last);
        RenderK rk = {};
        rk.width = plan.width; rk.height = plan.height; rk.max_depth = plan.max_depth;
        rk.t_min = plan.t_min; rk.seed = plan.seed;
        rk.s_begin = s0; rk.s_end = s1;
        rk.sub_spp = plan.sub_spp;
        rk.subs_per_tile = (s1 - s0 + plan.sub_spp - 1) / plan.sub_spp;
        // jobs of JOB_UNITS units halve the cross-wave hand-offs; small launches keep single units for load balance
        // (measured: headline 592 ms with 2, 606 with 1 and a 4-slot ring, 593 with 4; 9 M-sample frame 7.0 ms with 1 or 2, 9.3 with 4)
        const bool many_units = plan.tiles_owned * (int64_t)rk.subs_per_tile >= (int64_t)64 * grid * (PT_BLOCK / 64);
        // (a rank that owns fewer tiles than the GPU has waves traces consecutive jobs of one tile at the same time: single units hand the
        // tile's ticket on sooner -- 1/8 of the headline frame 68.2 -> 66.6 ms, while the whole frame loses 1.2 % with single units)
        const bool tiles_cover_waves = !SINGLE_UNITS_BELOW_WAVES || plan.tiles_owned >= (int64_t)grid * (PT_BLOCK / 64);
        rk.job_units = std::max(1, std::min(many_units && tiles_cover_waves ? JOB_UNITS : 1, rk.subs_per_tile));
        rk.tiles_owned = (int)std::max<int64_t>(1, plan.tiles_owned);
        rk.pool_mode = pool ? 1 : 0;  // (job_units is 1: the rank's tiles do not cover the waves)
        Schedule sch;
        const int jobs_per_tile = make_schedule(sch, rk.tiles_owned, grid * (PT_BLOCK / 64), rk.s_begin, rk.s_end, rk.sub_spp, rk.job_units);  // host/schedule.cpp
        static_assert(sizeof(rk.lvl) == sizeof(sch.lvl), "RenderK::lvl is Schedule::lvl");
        std::memcpy(rk.lvl, sch.lvl, sizeof(rk.lvl));
        rk.subs_per_tile = sch.units_per_tile;
        int64_t units = plan.tiles_owned * (int64_t)jobs_per_tile;
        if (units > 0x7FFFFFFF) throw RtError(RT_ERR_UNSUPPORTED, "too many work units per launch");
        rk.n_units = (int)units;
        rk.tiles_x = plan.tiles_x; rk.rank = plan.rank; rk.world = plan.world;
        rk.sppm_est = plan.sppm_est;
        rk.time0 = plan.time0;
        rk.time1 = plan.time1;
        rk.time_slots = moving ? 1 : 0;
        rk.n_top = n_top;
        rk.n_topq = n_topq;
        rk.coop_stack = (int)stack5;
        rk.coop_pool = tun.coop_pool > 0 ? std::min(tun.coop_pool, (int)COOP_POOL) : (int)COOP_POOL;
        HIP_CHECK(hipMemsetAsync(counter.p, 0, sizeof(unsigned int), stream));
        HIP_CHECK(hipMemsetAsync((char*)counter.p + 64, 0, JOB_LISTS * sizeof(unsigned int), stream));  // the per-list job counters (words 16.. of the small block)
        HIP_CHECK(hipMemsetAsync(tickets.p, 0, std::max<size_t>(16, (size_t)plan.tiles_owned * sizeof(unsigned int)), stream));
        hipEvent_t e0 = events.make(), e1 = events.make();
        HIP_CHECK(hipEventRecord(e0, stream));
        if (rk.n_units > 0) {
            if (kernel == 5)
                hipLaunchKernelGGL(fn_coop, dim3(grid), dim3(PT_BLOCK), smem, stream, view, ck, rk, (double*)ringp.p, (double*)accum.p,
                                   (unsigned int*)tickets.p, (unsigned int*)counter.p, (int*)err.p, (uint64_t*)lease.w->coop);
            else
                hipLaunchKernelGGL(fn, dim3(grid), dim3(PT_BLOCK), smem, stream, view, ck, rk, (double*)ringp.p, (double*)accum.p,
                                   (unsigned int*)tickets.p, (unsigned int*)counter.p, (int*)err.p);
            HIP_CHECK(hipGetLastError());
        }
        HIP_CHECK(hipEventRecord(e1, stream));
        pt_ev.emplace_back(e0, e1);
        launches++;
    }
    if (n_pix > 0 && !plan.ext_accum) {
        hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((n_pix + 255) / 256)), dim3(256), 0, stream, (const double*)accum.p, d_tiles,
                           n_pix, plan.spp, plan.width, plan.height, plan.tiles_x, plan.rank, plan.world);
        HIP_CHECK(hipGetLastError());
    }
    int h_err = 0;
    HIP_CHECK(hipMemcpyAsync(&h_err, err.p, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    if (st) {
        double kms = 0;
        for (auto& p : pt_ev) {
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, p.first, p.second));
            kms += ms;
        }
        st->kernel_ms = kms;
        st->reduce_ms = 0.;  // the ordered reduction happens inside pt_kernel
        st->launches = launches;
        st->kernel_used = kernel;
        st->scene_in_lds = lds ? 1 : 0;
        st->block_threads = PT_BLOCK;
        st->grid_blocks = grid;
        st->spp_chunk = plan.spp_chunk;
        st->scene_bytes = s.flat.blob.size();
        st->reserved[1] = (uint64_t)(ring_bytes + (size_t)n_pix * 3 * sizeof(double) + (size_t)plan.tiles_owned * sizeof(unsigned int) +
                                     ((kernel == 5) ? (size_t)grid * COOP_POOL * COOP_REC * sizeof(uint64_t) : 0));  // workspace bytes
    }
#ifdef RTAMD_COOP_STATS
    if (kernel == 5) {
        unsigned long long hs[16], z[16] = {0};
        HIP_CHECK(hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_coop_stats), sizeof(hs)));
        const char* names[8] = {"serve rounds (lanes at start)", "serve passes (busy lanes)", "walk phases (lanes)", "shade phases (lanes)", "suspensions (walks)", "idle sleeps", "adoptions (paths)", "... that hit inside the instance"};
        for (int i = 0; i < 8; i++)
            fprintf(stderr, "[coop stats] %-32s %12llu  lanes %14llu  (%.1f per event)\n", names[i], hs[2 * i], hs[2 * i + 1],
                    hs[2 * i] ? (double)hs[2 * i + 1] / (double)hs[2 * i] : 0.);
        HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_coop_stats), z, sizeof(z)));
        unsigned long long tm[8], tot = 0;
        HIP_CHECK(hipMemcpyFromSymbol(tm, HIP_SYMBOL(g_coop_time), sizeof(tm)));
        for (int i = 0; i < 7; i++) tot += tm[i];
        const char* tn[8] = {"fold + regenerate", "world-space walk", "park", "adopt", "shade", "serve (batch)", "tail: serve rest / idle", "(of park + adopt: the two ring pops)"};
        for (int i = 0; i < 8; i++) fprintf(stderr, "[coop time] %-26s %5.1f %%\n", tn[i], tot ? 100. * (double)tm[i] / (double)tot : 0.);
        HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_coop_time), z, sizeof(tm)));
    }
#endif
#ifdef RTAMD_PHASE_STATS
    if (kernel != 5) {
        unsigned long long hp[48], z[48] = {0};
        HIP_CHECK(hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_phase), sizeof(hp)));
        HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)));
        const double tot = (double)(hp[0] + hp[1] + hp[8]);
        const char* nm[16] = {"regeneration", "traverse (all)", "materialize", "shade (all)", "  Lambertian / light branch", "  Isotropic branch", "  Metal branch",
                              "  Dielectric branch", "after traverse (all)", "  leaf sections of traverse", "inner-node steps", "leaf items", "cube tests",
                              "segments", "mixture steps", "  node steps inside an instance"};
        fprintf(stderr, "[phase] kernel %d lds %d  wave clocks per segment-iteration %.0f (iterations %llu)\n", kernel, lds ? 1 : 0, hp[16] ? tot / (double)hp[16] : 0.,
                hp[16]);
        for (int i = 0; i < 16; i++)
            fprintf(stderr, "[phase] %-30s time %6.3f  wave-exec/iter %8.3f  lanes/exec %5.1f  (util %.3f)\n", nm[i], i < 10 ? (double)hp[i] / tot : 0.,
                    hp[16] ? (double)hp[16 + i] / (double)hp[16] : 0., hp[16 + i] ? (double)hp[32 + i] / (double)hp[16 + i] : 0.,
                    hp[16 + i] ? (double)hp[32 + i] / (64. * (double)hp[16 + i]) : 0.);
        // what material-pure waves could return at most: every branch at full lanes instead of its measured share
        double save = 0.;
        for (int i = 4; i <= 7; i++)
            if (hp[16 + i]) save += (double)hp[i] * (1. - (double)hp[32 + i] / (64. * (double)hp[16 + i]));
        fprintf(stderr, "[phase] material branches: %.4f of wave time; re-binned to full waves they would return at most %.4f\n",
                (double)(hp[4] + hp[5] + hp[6] + hp[7]) / tot, save / tot);
    }
#endif
#ifdef RT_TAIL_STATS
    if (kernel != 5 && st) {
        // s_memrealtime: the 100 MHz counter every CU sees alike (s_memtime has constant offsets of milliseconds between CUs)
        const int wpb = PT_BLOCK / 64, nw = grid * wpb;
        std::vector<unsigned long long> te(8192), tb(8192);
        HIP_CHECK(hipMemcpyFromSymbol(te.data(), HIP_SYMBOL(g_tail_end), sizeof(unsigned long long) * 8192));
        HIP_CHECK(hipMemcpyFromSymbol(tb.data(), HIP_SYMBOL(g_tail_beg), sizeof(unsigned long long) * 8192));
        std::vector<unsigned long long> tn(8192);
        std::vector<unsigned int> xc(8192);
        HIP_CHECK(hipMemcpyFromSymbol(xc.data(), HIP_SYMBOL(g_tail_xcc), sizeof(unsigned int) * 8192));
        HIP_CHECK(hipMemcpyFromSymbol(tn.data(), HIP_SYMBOL(g_tail_entry), sizeof(unsigned long long) * 8192));
        unsigned long long x0[8];
        for (int x = 0; x < 8; x++) x0[x] = ~0ull;
        for (int i = 0; i < nw; i++) x0[xc[i]] = std::min(x0[xc[i]], tn[i]);
        {
            std::vector<double> n(nw), sg(nw);
            for (int i = 0; i < nw; i++) { n[i] = (double)(tn[i] - x0[xc[i]]); sg[i] = (double)(tb[i] - tn[i]); }
            std::sort(n.begin(), n.end());
            std::sort(sg.begin(), sg.end());
            fprintf(stderr, "[tail stats] (ticks) kernel entry after the XCD's first wave: 50%% %.0f  99%% %.0f  last %.0f; entry -> main loop (staging): 50%% %.0f  last %.0f\n", n[nw / 2],
                    n[nw * 99 / 100], n[nw - 1], sg[nw / 2], sg[nw - 1]);
        }
        std::vector<double> b(nw), e(nw);
        for (int i = 0; i < nw; i++) { b[i] = (double)(tb[i] - x0[xc[i]]); e[i] = (double)(te[i] - x0[xc[i]]); }
        std::sort(b.begin(), b.end());
        std::sort(e.begin(), e.end());
        const double k = st->kernel_ms / e[nw - 1];  // ticks -> ms through the event time
        double idle = 0;
        for (int i = 0; i < nw; i++) idle += e[nw - 1] - e[i];
        fprintf(stderr, "[tail stats] kernel %.2f ms; wave starts: 50%% %.3f  99%% %.3f  last %.3f ms; wave ends: first %.2f  10%% %.2f  50%% %.2f  90%% %.2f  99%% %.2f  last %.2f ms; "
                        "mean idle wave-time at the end %.2f ms\n",
                st->kernel_ms, k * b[nw / 2], k * b[nw * 99 / 100], k * b[nw - 1], k * e[0], k * e[nw / 10], k * e[nw / 2], k * e[nw * 9 / 10], k * e[nw * 99 / 100],
                k * e[nw - 1], k * idle / nw);
    }
#endif
#ifdef RT_FOLD_STATS
    {
        unsigned long long hs[8], z[8] = {0};
        HIP_CHECK(hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_fold_stats), sizeof(hs)));
        fprintf(stderr, "[fold stats] calls %llu  cycles %llu (%.0f per call)  units folded %llu  ticket mismatches %llu  sleeps %llu\n", hs[0], hs[1],
                hs[0] ? (double)hs[1] / (double)hs[0] : 0., hs[2], hs[3], hs[4]);
        HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_fold_stats), z, sizeof(z)));
    }
#endif
    if (h_err & 1) throw RtError(RT_ERR_UNIT_ZERO, "unitizing zero vector (device)");
    if (h_err & ~1) throw RtError(RT_ERR_INTERNAL, "device invariant failed (error bits " + std::to_string(h_err) + ": 2 = an instance below an instance "
                                                   "reached an object-space walk, 4 = a kernel 5 ring entry was not published in time)");
}

// ---------------------------------------------------------------- kernel 6 driver ----
// Cycles of {pt_kernel_wf, wf_walk_kernel} on `stream` until every workgroup reports that it is done with the frame.  The host
// looks at the "unfinished" words once per batch of WF_BATCH cycles (a cycle enqueued after the frame is complete costs two empty
// launches), so there is one synchronisation per batch, not per cycle.
static const int WF_BATCH = 8;
static void render_tiles_wf(const rt_scene& s, const FlatView& view, const CameraDev& cam, const RenderPlan& plan, const Tuning& tun, double* d_tiles,
                            hipStream_t stream, rt_stats* st, int dev, const DevInfo& di, uint32_t stack6, uint32_t n_entry6, size_t lds_pt, uint32_t stack6w,
                            size_t tables_w) {
    const size_t lds_max = di.lds_max;
    const int integ = plan.integrator;
    if (integ == 1 && view.n_lights == 0) throw RtError(RT_ERR_ARG, "integrator 1 (light importance sampling) needs rt_scene_set_lights");
    if (integ == 2 && !plan.sppm_est) throw RtError(RT_ERR_ARG, "integrator 2 (SPPM) is reached through rt_render_sppm");
    typedef void (*pt_wf_fn)(FlatView, CamK, RenderK, double*, double*, unsigned int*, unsigned int*, int*, WfArgs);
    const bool tie = tie_scene(view);  // the TIE variants of both launches ("EXACT ties" above traverse<>)
    pt_wf_fn fn = tie ? ((view.n_inline2 != 0u) ? ((integ == 1) ? pt_kernel_wf<1, true, true> : (integ == 2) ? pt_kernel_wf<2, true, true> : pt_kernel_wf<0, true, true>)
                                                : ((integ == 1) ? pt_kernel_wf<1, false, true> : (integ == 2) ? pt_kernel_wf<2, false, true> : pt_kernel_wf<0, false, true>))
                      : ((view.n_inline2 != 0u) ? ((integ == 1) ? pt_kernel_wf<1, true> : (integ == 2) ? pt_kernel_wf<2, true> : pt_kernel_wf<0, true>)
                                                : ((integ == 1) ? pt_kernel_wf<1> : (integ == 2) ? pt_kernel_wf<2> : pt_kernel_wf<0>));
    void (*fn_walk)(FlatView, WfWalkK) = tie ? wf_walk_kernel<true> : wf_walk_kernel<false>;
    if (lds_pt > 48 * 1024) HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pt));
    int bpc = 0;
    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, (const void*)fn, PT_BLOCK, lds_pt));
    if (bpc < 1) bpc = 1;
    const int grid = di.cus * bpc;
    // the walk launch: two 512-thread workgroups per CU when the stacks allow it; what is left of that share of LDS caches NodeQ
    const size_t stacks_w = (size_t)stack6w * WF_WALK_BLOCK * sizeof(uint32_t);
    size_t share = lds_max / 2;
    if (tables_w + stacks_w > share) share = lds_max;
    uint32_t n_topq_w = (uint32_t)std::min<size_t>((share - tables_w - stacks_w) / sizeof(NodeQ), view.n_nodes2);
    if (tun.n_top >= 0) n_topq_w = std::min<uint32_t>(n_topq_w, (uint32_t)tun.n_top);
    const size_t lds_walk = tables_w + coop_a16(n_topq_w * (uint32_t)sizeof(NodeQ)) + stacks_w;
    if (lds_walk > 48 * 1024) HIP_CHECK(hipFuncSetAttribute((const void*)fn_walk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_walk));
    int bpc_w = 0;
    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc_w, (const void*)fn_walk, WF_WALK_BLOCK, lds_walk));
    if (bpc_w < 1) bpc_w = 1;
    const int grid_w = di.cus * bpc_w;
    // Workspace within a budget (rt_tuning.wf_workspace_mb; default 1 900 MB -- the first version took 8.6 GB flat: 96 unit buffers per
    // wave and 32 768 records per workgroup segment, twice): 60 % of it for the waves' unit buffers, never more than the launch has units
    // for; 40 % for the two record pools.  Fewer buffers / shorter segments cost speed (waves sit on full rings, the gate closes more
    // often), never results.
    const size_t budget = (size_t)(tun.wf_workspace_mb > 0 ? tun.wf_workspace_mb : 1900) * 1000000u;
    const size_t n_waves6 = (size_t)grid * (PT_BLOCK / 64);
    const int64_t units_per_tile = (std::min(plan.spp, plan.spp_chunk) + plan.sub_spp - 1) / plan.sub_spp;
    const int64_t need_units = (plan.tiles_owned * units_per_tile + (int64_t)n_waves6 - 1) / (int64_t)n_waves6 + 2;
    uint32_t ring_units = (uint32_t)std::min<size_t>((size_t)WF_RING_UNITS, (budget * 6 / 10) / (n_waves6 * UNIT_DOUBLES * sizeof(double)));
    ring_units = std::max<uint32_t>(8u, std::min<uint32_t>(ring_units, (uint32_t)std::min<int64_t>(need_units, WF_RING_UNITS)));
    uint32_t seg = (uint32_t)std::min<size_t>((size_t)WF_SEG_DEFAULT, (budget * 4 / 10) / ((size_t)2 * grid * WF_REC_U * 16) / WF_CHUNK * WF_CHUNK);
    seg = std::max<uint32_t>(seg, 2 * WF_GATE);
    if (tun.coop_pool > 0) seg = std::max<uint32_t>(2 * WF_GATE, ((uint32_t)tun.coop_pool + WF_CHUNK - 1) / WF_CHUNK * WF_CHUNK);  // rt_tuning test hook: small segments

    const int64_t n_pix = plan.tiles_owned * TILE_PIX;
    const size_t ring_bytes = n_waves6 * ring_units * UNIT_DOUBLES * sizeof(double);
    const size_t pool_bytes = (size_t)grid * seg * WF_REC_U * 16;
    const size_t off_cnt = 2 * pool_bytes, off_book = off_cnt + 2 * (size_t)grid * 4, off_misc = off_book + (size_t)grid * WF_BOOK_WORDS * 4;
    const size_t wf_bytes = off_misc + 256;  // misc: cursor, WF_BATCH unfinished words
    WorkspaceLease lease(dev, ring_bytes, std::max<size_t>(16, (size_t)n_pix * 3 * sizeof(double)),
                         std::max<size_t>(16, (size_t)plan.tiles_owned * sizeof(unsigned int)), wf_bytes);
    char* small = (char*)lease.w->small;
    char* wfb = (char*)lease.w->coop;
    unsigned int* counter = (unsigned int*)small;
    int* err = (int*)(small + 16);
    uint32_t* cursor = (uint32_t*)(wfb + off_misc);
    uint32_t* unf = (uint32_t*)(wfb + off_misc + 64);
    HIP_CHECK(hipMemsetAsync(small, 0, WS_SMALL, stream));
    Events events;
    CamK ck = to_camk(cam);
    int launches = 0;
    double kms = 0.;
    for (int s0 = 0; s0 < plan.spp; s0 += plan.spp_chunk) {
        const int s1 = std::min(s0 + plan.spp_chunk, plan.spp);
        RenderK rk = {};
        rk.width = plan.width; rk.height = plan.height; rk.max_depth = plan.max_depth;
        rk.t_min = plan.t_min; rk.seed = plan.seed;
        rk.s_begin = s0; rk.s_end = s1;
        rk.sub_spp = plan.sub_spp;
        rk.subs_per_tile = (s1 - s0 + plan.sub_spp - 1) / plan.sub_spp;
        rk.job_units = 1;  // kernel 6 deals single units (next_unit_wf)
        int64_t units = plan.tiles_owned * (int64_t)rk.subs_per_tile;
        if (units > 0x7FFFFFFF) throw RtError(RT_ERR_UNSUPPORTED, "too many work units per launch");
        rk.n_units = (int)units;
        rk.tiles_x = plan.tiles_x; rk.rank = plan.rank; rk.world = plan.world;
        rk.tiles_owned = (int)std::max<int64_t>(1, plan.tiles_owned);
        rk.sppm_est = plan.sppm_est;
        rk.time0 = plan.time0;
        rk.time1 = plan.time1;
        rk.n_top = (int)std::min<uint32_t>(128u, view.world_top2);
        rk.n_topq = (int)n_entry6;
        rk.coop_stack = (int)stack6;
        rk.coop_pool = 0;
        HIP_CHECK(hipMemsetAsync(counter, 0, sizeof(unsigned int), stream));
        HIP_CHECK(hipMemsetAsync(lease.w->tickets, 0, std::max<size_t>(16, (size_t)plan.tiles_owned * sizeof(unsigned int)), stream));
        if (rk.n_units <= 0) continue;
        hipEvent_t e0 = events.make(), e1 = events.make();
        HIP_CHECK(hipEventRecord(e0, stream));
        int cycle = 0;
        for (;;) {
            HIP_CHECK(hipMemsetAsync(unf, 0, WF_BATCH * sizeof(uint32_t), stream));
            for (int b = 0; b < WF_BATCH; b++, cycle++) {
                WfArgs wa;
                wa.pool_a = (U2*)(wfb + (size_t)((cycle + 1) & 1) * pool_bytes);
                wa.pool_b = (U2*)(wfb + (size_t)(cycle & 1) * pool_bytes);
                wa.cnt_a = (const uint32_t*)(wfb + off_cnt) + (size_t)((cycle + 1) & 1) * grid;
                wa.cnt_b = (uint32_t*)(wfb + off_cnt) + (size_t)(cycle & 1) * grid;
                wa.book = (uint32_t*)(wfb + off_book);
                wa.unfinished = unf + b;
                wa.seg = seg;
                wa.first = cycle == 0 ? 1 : 0;
                wa.ring_units = ring_units;
                hipLaunchKernelGGL(fn, dim3(grid), dim3(PT_BLOCK), lds_pt, stream, view, ck, rk, (double*)lease.w->ring, (double*)lease.w->accum,
                                   (unsigned int*)lease.w->tickets, counter, err, wa);
                HIP_CHECK(hipGetLastError());
                HIP_CHECK(hipMemsetAsync(cursor, 0, sizeof(uint32_t), stream));
                WfWalkK wk;
                wk.pool = wa.pool_b;
                wk.cnt = wa.cnt_b;
                wk.cursor = cursor;
                wk.grid_pt = (uint32_t)grid;
                wk.seg = seg;
                wk.n_topq = n_topq_w;
                wk.stack = stack6w;
                wk.t_min = plan.t_min;
                hipLaunchKernelGGL(fn_walk, dim3(grid_w), dim3(WF_WALK_BLOCK), lds_walk, stream, view, wk);
                HIP_CHECK(hipGetLastError());
                launches++;
            }
            uint32_t h_unf[WF_BATCH];
            HIP_CHECK(hipMemcpyAsync(h_unf, unf, sizeof(h_unf), hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
            if (h_unf[WF_BATCH - 1] == 0u) break;
            if (cycle > (1 << 22)) throw RtError(RT_ERR_INTERNAL, "kernel 6 made no progress");
        }
        HIP_CHECK(hipEventRecord(e1, stream));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        kms += ms;
    }
    if (n_pix > 0) {
        hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((n_pix + 255) / 256)), dim3(256), 0, stream, (const double*)lease.w->accum, d_tiles,
                           n_pix, plan.spp, plan.width, plan.height, plan.tiles_x, plan.rank, plan.world);
        HIP_CHECK(hipGetLastError());
    }
    int h_err = 0;
    HIP_CHECK(hipMemcpyAsync(&h_err, err, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    if (st) {
        st->kernel_ms = kms;
        st->reduce_ms = 0.;
        st->launches = launches;
        st->kernel_used = 6;
        st->scene_in_lds = 0;
        st->block_threads = PT_BLOCK;
        st->grid_blocks = grid;
        st->spp_chunk = plan.spp_chunk;
        st->scene_bytes = s.flat.blob.size();
        st->reserved[1] = (uint64_t)(ring_bytes + (size_t)n_pix * 3 * sizeof(double) + (size_t)plan.tiles_owned * sizeof(unsigned int) + wf_bytes);
        st->reserved[2] = (uint64_t)grid_w | ((uint64_t)n_topq_w << 32);
    }
#ifdef RTAMD_WF_STATS
    {
        unsigned long long hs[16], z[16] = {0};
        HIP_CHECK(hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_wf_stats), sizeof(hs)));
        const char* names[12] = {"parks", "adoptions", "generated", "wave iterations", "wave exits: ring full / no unit", "wave exits: gate closed", "wave exits: finished",
                                 "wave cycles in launches", "alive lanes summed", "entry walks", "walk passes", "walk pass lanes"};
        fprintf(stderr, "[wf stats] cycles %d  grid %d x %d  walk grid %d  seg %u  n_topq_w %u\n", launches, grid, PT_BLOCK, grid_w, seg, n_topq_w);
        for (int i = 0; i < 12; i++) fprintf(stderr, "[wf stats] %-34s %14llu\n", names[i], hs[i]);
        HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_wf_stats), z, sizeof(z)));
    }
#endif
    if (h_err & 1) throw RtError(RT_ERR_UNIT_ZERO, "unitizing zero vector (device)");
    if (h_err & ~1) throw RtError(RT_ERR_INTERNAL, "device invariant failed (error bits " + std::to_string(h_err) + ")");
}

// ---------------------------------------------------------------- SPPM driver ----
// SPPMIntegrator::new (photon_mapper.rs:139-233) + Camera::capture_image with the SPPM sample_ray.
namespace {
struct Grid {
    DevBuf cell_start, cursor, cell_of, index, block_sums, mnmx;
    size_t cap_cells = 0, cap_photons = 0;
    GridK k{};
};
void build_grid(Grid& g, const PhotonBuf& pb, unsigned int n, hipStream_t stream) {
    g.k = GridK{};
    g.k.n = n;
    g.k.dim[0] = g.k.dim[1] = g.k.dim[2] = 1;
    g.k.cell = 1.;
    g.k.inv_cell = 1.;
    if (n == 0) return;
    if (!g.mnmx.p) g.mnmx.alloc(6 * sizeof(unsigned long long));
    unsigned long long init[6] = {~0ull, ~0ull, ~0ull, 0ull, 0ull, 0ull}, keys[6];
    HIP_CHECK(hipMemcpyAsync(g.mnmx.p, init, sizeof(init), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(bbox_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (const double*)pb.pos, n, (unsigned long long*)g.mnmx.p);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(keys, g.mnmx.p, sizeof(keys), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    double lo[3], hi[3], ext[3];
    for (int a = 0; a < 3; a++) {
        lo[a] = f64_unkey(keys[a]);
        hi[a] = f64_unkey(keys[3 + a]);
        ext[a] = hi[a] - lo[a];
        if (!(ext[a] >= 0.) || !std::isfinite(ext[a])) throw RtError(RT_ERR_UNSUPPORTED, "photon positions are not finite");
    }
    // photons lie on surfaces: aim at ~4 per occupied cell  ->  cell ~ 2 * sqrt(area / n)
    double area = 2. * (ext[0] * ext[1] + ext[1] * ext[2] + ext[2] * ext[0]);
    double longest = std::fmax(ext[0], std::fmax(ext[1], ext[2]));
    double cell = 2. * std::sqrt(area / (double)n);
    if (!(cell > 0.)) cell = longest > 0. ? longest : 1.;
    const double min_cell = longest / 160.;  // at most 160 cells per axis
    if (cell < min_cell) cell = min_cell;
    if (!(cell > 0.)) cell = 1.;
    size_t cells = 1;
    for (int a = 0; a < 3; a++) {
        int d = (int)std::floor(ext[a] / cell) + 1;
        if (d < 1) d = 1;
        if (d > 161) d = 161;
        g.k.dim[a] = d;
        g.k.lo[a] = lo[a];
        cells *= (size_t)d;
    }
    g.k.cell = cell;
    g.k.inv_cell = 1. / cell;
    const size_t n_scan = cells + 1;
    if (g.cap_cells < n_scan) {
        g.cell_start.alloc(n_scan * 4);
        g.cursor.alloc(n_scan * 4);
        g.block_sums.alloc(((n_scan + 1023) / 1024 + 1) * 4);
        g.cap_cells = n_scan;
    }
    if (g.cap_photons < n) {
        g.cell_of.alloc((size_t)n * 4);
        g.index.alloc((size_t)n * 4);
        g.cap_photons = n;
    }
    HIP_CHECK(hipMemsetAsync(g.cell_start.p, 0, n_scan * 4, stream));
    HIP_CHECK(hipMemsetAsync(g.cursor.p, 0, n_scan * 4, stream));
    hipLaunchKernelGGL(cell_count_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (const double*)pb.pos, n, g.k, (unsigned int*)g.cell_start.p,
                       (unsigned int*)g.cell_of.p);
    const unsigned int nb = (unsigned int)((n_scan + 1023) / 1024);
    hipLaunchKernelGGL(scan_block_kernel, dim3(nb), dim3(1024), 0, stream, (unsigned int*)g.cell_start.p, (unsigned int)n_scan, (unsigned int*)g.block_sums.p);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, stream, (unsigned int*)g.block_sums.p, nb);
    hipLaunchKernelGGL(scan_add_kernel, dim3(nb), dim3(1024), 0, stream, (unsigned int*)g.cell_start.p, (unsigned int)n_scan, (const unsigned int*)g.block_sums.p, 0u);
    hipLaunchKernelGGL(cell_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, (const unsigned int*)g.cell_of.p, (const unsigned int*)g.cell_start.p,
                       (unsigned int*)g.cursor.p, (unsigned int*)g.index.p);
    HIP_CHECK(hipGetLastError());
    g.k.cell_start = (const unsigned int*)g.cell_start.p;
    g.k.index = (const unsigned int*)g.index.p;
}
struct PhotonStore {
    DevBuf pos, power, norm, count;
    DevBuf spos, spower, snorm;  // the same photons in grid order (permute_kernel)
    PhotonBuf b{};
    PhotonSorted s{};
    void alloc(unsigned int cap) {
        pos.alloc((size_t)cap * 24);
        power.alloc((size_t)cap * 24);
        norm.alloc((size_t)cap * 24);
        spos.alloc((size_t)cap * 24);
        spower.alloc((size_t)cap * 24);
        snorm.alloc((size_t)cap * 24);
        if (!count.p) count.alloc(4);
        b.pos = (double*)pos.p; b.power = (double*)power.p; b.norm = (double*)norm.p; b.count = (unsigned int*)count.p; b.cap = cap;
        s.pos = (const double*)spos.p; s.power = (const double*)spower.p; s.norm = (const double*)snorm.p;
    }
    void sort_into_grid(const Grid& g, unsigned int n, hipStream_t stream) {
        if (n == 0) return;
        hipLaunchKernelGGL(permute_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, (const unsigned int*)g.index.p, b, (double*)spos.p,
                           (double*)spower.p, (double*)snorm.p);
        HIP_CHECK(hipGetLastError());
    }
};
}  // namespace

void render_sppm(const rt_scene& s, const CameraDev& cam, RenderPlan plan, const rt_sppm_config& cfg, double* d_tiles, double* stats_host,
                 void* stream_, rt_stats* st, uint64_t* totals2) {
    if (!s.committed) throw RtError(RT_ERR_NOT_COMMITTED, "scene not committed");
    const Tuning tun = tuning();  // one snapshot per call
    if (s.lights.empty()) throw RtError(RT_ERR_ARG, "SPPM needs lights (rt_scene_set_lights)");
    if (s.flat.view.kinds_mask & (1u << NK_MEDIUM_BEGIN))
        throw RtError(RT_ERR_UNSUPPORTED, "the SPPM pre-pass does not support ConstantMedium (volume events have no photon-map estimate)");
    if (s.flat.view.n_msph != 0u || s.flat.view.has_noise != 0u || plan.time1 > plan.time0)
        throw RtError(RT_ERR_UNSUPPORTED, "the photon passes have no notion of time: the book-2 extensions (moving spheres, noise textures, an open shutter) render with integrator 0");
    if (cfg.iterations < 1 || cfg.photons_per_iter < 1 || cfg.k_global < 1 || cfg.k_caustic < 1 || cfg.max_bounces < 1 || !(cfg.alpha > 0.))
        throw RtError(RT_ERR_ARG, "bad rt_sppm_config");
    hipStream_t stream = (hipStream_t)stream_;
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    const DevInfo& di = dev_info(dev);
    FlatView view = s.flat.view;
    view.base = device_blob(s, dev);
    double cam_abs = std::fmax(std::fmax(std::fabs(cam.origin[0]), std::fabs(cam.origin[1])), std::fabs(cam.origin[2])) + std::fabs(cam.lens_radius);
    const bool accel = view.accel_ok && cam_abs <= view.origin_limit2 && std::isfinite(cam_abs) && (size_t)view.stack2 * 256 * 4 <= di.lds_max;
    const size_t smem = accel ? (size_t)view.stack2 * 256 * sizeof(uint32_t) : 0;
    // photon pass: stage the accel's hot tables into LDS when at least two 256-thread blocks still fit on a CU
    const size_t hot2 = (size_t)(view.stage2_end - view.stage2_begin);
    const size_t smem_photon = hot2 + smem;
    const bool photon_lds = accel && hot2 > 0 && 2 * smem_photon <= di.lds_max && !tun.no_lds;
    if (photon_lds && smem_photon > 48 * 1024)
        HIP_CHECK(hipFuncSetAttribute((const void*)photon_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_photon));

    // AllLights::new (light.rs:202-217) + the fixed-point shift of the flux accumulators
    const int nl = (int)s.lights.size();
    std::vector<double> h_flux(3 * nl), h_scale(nl), h_cum(nl), lp(nl);
    double tot = 0., maxp = 0.;
    for (int i = 0; i < nl; i++) {
        const ObjectRec& o = s.objects[s.lights[i]];
        for (int c = 0; c < 3; c++) h_flux[3 * i + c] = o.light_flux[c];
        h_scale[i] = o.light_scale;
        double px = o.light_flux[0] * o.light_scale, py = o.light_flux[1] * o.light_scale, pz = o.light_flux[2] * o.light_scale;
        lp[i] = std::sqrt(px * px + py * py + pz * pz);
        tot = tot + lp[i];
        maxp = std::fmax(maxp, std::fmax(std::fabs(px), std::fmax(std::fabs(py), std::fabs(pz))));
    }
    double total = 0.;
    for (int i = 0; i < nl; i++) {
        total = total + lp[i] / tot;
        h_cum[i] = total;
    }
    int S = 40;
    if (maxp > 0. && std::isfinite(maxp)) {
        int e;
        std::frexp(maxp, &e);
        S = 40 - e;
        S = S < 0 ? 0 : (S > 60 ? 60 : S);
    }
    DevBuf d_flux, d_scale, d_cum, d_cam, d_err, d_gp, d_stats, d_est, d_cursor;
    d_cursor.alloc(4);
    d_flux.alloc(h_flux.size() * 8); d_scale.alloc(nl * 8); d_cum.alloc(nl * 8);
    HIP_CHECK(hipMemcpy(d_flux.p, h_flux.data(), h_flux.size() * 8, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_scale.p, h_scale.data(), nl * 8, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_cum.p, h_cum.data(), nl * 8, hipMemcpyHostToDevice));
    LightK lk{nl, total, (const double*)d_cum.p, (const double*)d_flux.p, (const double*)d_scale.p};
    CamK ck = to_camk(cam);
    d_cam.alloc(sizeof(CamK));
    HIP_CHECK(hipMemcpy(d_cam.p, &ck, sizeof(CamK), hipMemcpyHostToDevice));
    d_err.alloc(4);
    HIP_CHECK(hipMemset(d_err.p, 0, 4));
    const size_t npix = (size_t)plan.width * plan.height;
    d_gp.alloc(npix * 7 * 8);
    d_stats.alloc(npix * 10 * 8);
    d_est.alloc(npix * 6 * 8);
    HIP_CHECK(hipMemset(d_stats.p, 0, npix * 10 * 8));

    // Two streams: the photon pass of iteration i+1 (stream P) runs beside the grid build + eye pass + gather of iteration i
    // (the caller's stream), on double-buffered photon stores.  Both are latency-bound on their own (paths of 1..max_bounces
    // segments; one wave per pixel), so together they fill the GPU.  Results do not depend on the overlap.
    struct StreamP {
        hipStream_t s = nullptr;
        ~StreamP() {
            if (s) (void)hipStreamDestroy(s);
        }
    } sp;
    HIP_CHECK(hipStreamCreateWithFlags(&sp.s, hipStreamNonBlocking));
    Events evs;
    PhotonStore pg[2], pc[2];
    unsigned int cap_g = (unsigned int)std::min<uint64_t>((uint64_t)cfg.photons_per_iter * 8 + 1024, 0x7FFFFFFFu);
    unsigned int cap_c = (unsigned int)std::min<uint64_t>((uint64_t)cfg.photons_per_iter * 2 + 1024, 0x7FFFFFFFu);
    if (tun.sppm_cap > 0) cap_g = cap_c = (unsigned int)tun.sppm_cap;  // rt_tuning test hook: forces the grow-and-retry path
    for (int j = 0; j < 2; j++) {
        pg[j].alloc(cap_g);
        pc[j].alloc(cap_c);
    }
    DevBuf d_err_p;  // error flags of the photon pass (its own word: the retry below clears it while the other stream runs)
    d_err_p.alloc(4);
    HIP_CHECK(hipMemset(d_err_p.p, 0, 4));
    Grid gg, gc;
    SppmK sk;
    sk.width = plan.width; sk.height = plan.height; sk.photons_per_iter = cfg.photons_per_iter;
    sk.k_global = cfg.k_global; sk.k_caustic = cfg.k_caustic; sk.max_bounces = cfg.max_bounces; sk.alpha = cfg.alpha;
    sk.seed = plan.seed; sk.S = S; sk.iteration = 0;
    sk.knn_cand = KNN_CAND;
    if (tun.knn_cand >= 0) sk.knn_cand = std::min(KNN_CAND, tun.knn_cand);  // rt_tuning test hook: forces the out-of-LDS selection
    // persistent photon waves: enough 256-thread blocks to fill every CU at 4 waves per SIMD, never more waves than chunks
    const int pblocks = (int)std::max<int64_t>(1, std::min<int64_t>(((int64_t)cfg.photons_per_iter + 4 * PHOTON_CHUNK - 1) / (4 * PHOTON_CHUNK), (int64_t)di.cus * 4));
    const int eblocks = (int)std::min<size_t>((npix + 255) / 256, (size_t)di.cus * 8);
    auto launch_photons = [&](int it) {  // on stream P, into store it % 2
        PhotonStore &g = pg[it & 1], &c = pc[it & 1];
        SppmK skp = sk;
        skp.iteration = it;
        HIP_CHECK(hipMemsetAsync(g.count.p, 0, 4, sp.s));
        HIP_CHECK(hipMemsetAsync(c.count.p, 0, 4, sp.s));
        HIP_CHECK(hipMemsetAsync(d_cursor.p, 0, 4, sp.s));
        if (accel && photon_lds)
            hipLaunchKernelGGL((photon_kernel<true, true>), dim3(pblocks), dim3(256), smem_photon, sp.s, view, skp, lk, g.b, c.b, (unsigned int*)d_cursor.p,
                               (int*)d_err_p.p);
        else if (accel)
            hipLaunchKernelGGL((photon_kernel<true, false>), dim3(pblocks), dim3(256), smem, sp.s, view, skp, lk, g.b, c.b, (unsigned int*)d_cursor.p,
                               (int*)d_err_p.p);
        else
            hipLaunchKernelGGL((photon_kernel<false, false>), dim3(pblocks), dim3(256), 0, sp.s, view, skp, lk, g.b, c.b, (unsigned int*)d_cursor.p,
                               (int*)d_err_p.p);
        HIP_CHECK(hipGetLastError());
    };
    uint64_t tg = 0, tc = 0;
    auto t0 = std::chrono::steady_clock::now();
    HIP_CHECK(hipStreamSynchronize(stream));  // the caller's earlier work on `stream` (uploads) precedes stream P
    launch_photons(0);
    hipEvent_t gather_done[2] = {nullptr, nullptr};  // gather of the iteration that last read store j
    for (int it = 0; it < cfg.iterations; it++) {
        sk.iteration = it;
        PhotonStore &sg = pg[it & 1], &sc = pc[it & 1];
        unsigned int ng = 0, nc = 0;
        for (;;) {  // wait for the photon pass; it is repeated with larger buffers if one overflowed (the pass is deterministic)
            int h_err = 0;
            HIP_CHECK(hipMemcpyAsync(&ng, sg.count.p, 4, hipMemcpyDeviceToHost, sp.s));
            HIP_CHECK(hipMemcpyAsync(&nc, sc.count.p, 4, hipMemcpyDeviceToHost, sp.s));
            HIP_CHECK(hipMemcpyAsync(&h_err, d_err_p.p, 4, hipMemcpyDeviceToHost, sp.s));
            HIP_CHECK(hipStreamSynchronize(sp.s));
            if (h_err & 1) throw RtError(RT_ERR_UNIT_ZERO, "unitizing zero vector (device, photon pass)");
            if (!(h_err & 2)) break;
            HIP_CHECK(hipMemsetAsync(d_err_p.p, 0, 4, sp.s));
            if (ng > sg.b.cap) {
                if (sg.b.cap >= 0x40000000u) throw RtError(RT_ERR_UNSUPPORTED, "photon map too large");
                sg.alloc(std::max(ng + 1024u, sg.b.cap * 2u));
            }
            if (nc > sc.b.cap) {
                if (sc.b.cap >= 0x40000000u) throw RtError(RT_ERR_UNSUPPORTED, "photon map too large");
                sc.alloc(std::max(nc + 1024u, sc.b.cap * 2u));
            }
            launch_photons(it);
        }
        tg += ng;
        tc += nc;
        if (it + 1 < cfg.iterations) {  // next photon pass: its store was last read by the gather of iteration it - 1
            if (gather_done[(it + 1) & 1]) HIP_CHECK(hipStreamWaitEvent(sp.s, gather_done[(it + 1) & 1], 0));
            launch_photons(it + 1);
        }
        build_grid(gg, sg.b, ng, stream);
        build_grid(gc, sc.b, nc, stream);
        sg.sort_into_grid(gg, ng, stream);
        sc.sort_into_grid(gc, nc, stream);
        if (accel) hipLaunchKernelGGL(eye_kernel<true>, dim3(eblocks), dim3(256), smem, stream, view, (const CamK*)d_cam.p, sk, (double*)d_gp.p, (int*)d_err.p);
        else hipLaunchKernelGGL(eye_kernel<false>, dim3(eblocks), dim3(256), 0, stream, view, (const CamK*)d_cam.p, sk, (double*)d_gp.p, (int*)d_err.p);
        const size_t pix_per_block = GATHER_BLOCK / 64;  // one wave per pixel
        hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((npix + pix_per_block - 1) / pix_per_block)), dim3(GATHER_BLOCK), 0, stream, sk,
                           (const double*)d_gp.p, gg.k, sg.s, gc.k, sc.s, (double*)d_stats.p, (int*)d_err.p);
        HIP_CHECK(hipGetLastError());
        if (!gather_done[it & 1]) gather_done[it & 1] = evs.make();
        HIP_CHECK(hipEventRecord(gather_done[it & 1], stream));
    }
    hipLaunchKernelGGL(estimate_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, stream, (const double*)d_stats.p, npix,
                       (double)cfg.iterations * (double)cfg.photons_per_iter, (double*)d_est.p);
    int h_err = 0;
    HIP_CHECK(hipMemcpyAsync(&h_err, d_err.p, 4, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    if (h_err & 1) throw RtError(RT_ERR_UNIT_ZERO, "unitizing zero vector (device, SPPM pre-pass)");
    const double prepass_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (stats_host) HIP_CHECK(hipMemcpy(stats_host, d_stats.p, npix * 10 * 8, hipMemcpyDeviceToHost));
    if (totals2) {
        totals2[0] = tg;
        totals2[1] = tc;
    }
    if (plan.spp > 0 && d_tiles) {
        plan.integrator = 2;
        plan.sppm_est = (const double*)d_est.p;
        render_tiles(s, cam, plan, d_tiles, stream_, st);
    }
    if (st) st->reserved[0] = (uint64_t)(prepass_s * 1e6);  // SPPM pre-pass time, microseconds
}

void finalize_tiles(const RenderPlan& plan, const double* d_accum, double* d_tiles, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t n_pix = plan.tiles_owned * TILE_PIX;
    if (n_pix > 0) {
        hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((n_pix + 255) / 256)), dim3(256), 0, stream, d_accum, d_tiles, n_pix, plan.spp, plan.width,
                           plan.height, plan.tiles_x, plan.rank, plan.world);
        HIP_CHECK(hipGetLastError());
    }
    HIP_CHECK(hipStreamSynchronize(stream));
}

void assemble_frame(const RenderPlan& plan, const double* d_gathered, int64_t stride, double* d_frame, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int64_t n = (int64_t)plan.width * plan.height;
    hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_gathered, stride, d_frame, plan.width,
                       plan.height, plan.tiles_x, plan.world);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(stream));
}

void debug_rng_device(uint64_t seed, uint64_t pixel, uint64_t sample, int n, uint64_t* out_host) {
    DevBuf b;
    b.alloc((size_t)n * 8);
    hipLaunchKernelGGL(rng_kernel, dim3(1), dim3(64), 0, 0, seed, pixel, sample, n, (uint64_t*)b.p);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpy(out_host, b.p, (size_t)n * 8, hipMemcpyDeviceToHost));
}
void debug_rng_floats_device(uint64_t seed, uint64_t pixel, uint64_t sample, int n, double lo, double hi, double* out_gen, double* out_range) {
    DevBuf b;
    b.alloc((size_t)n * 16);
    hipLaunchKernelGGL(rng_floats_kernel, dim3(1), dim3(64), 0, 0, seed, pixel, sample, n, lo, hi, (double*)b.p);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpy(out_gen, b.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(out_range, (const char*)b.p + (size_t)n * 8, (size_t)n * 8, hipMemcpyDeviceToHost));
}
void debug_math_device(int op, size_t n, const double* a, const double* bb, double* out) {
    DevBuf da, db, dc;
    da.alloc(n * 8);
    db.alloc(n * 8);
    dc.alloc(n * 8);
    HIP_CHECK(hipMemcpy(da.p, a, n * 8, hipMemcpyHostToDevice));
    if (bb) HIP_CHECK(hipMemcpy(db.p, bb, n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, op, n, (const double*)da.p, (const double*)db.p,
                       (double*)dc.p);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpy(out, dc.p, n * 8, hipMemcpyDeviceToHost));
}
void debug_hit_device(const rt_scene& s, int kernel, size_t n, const double* rays, double t_min, double t_max, double* out) {
    if (s.flat.view.kinds_mask & (1u << NK_MSPHERE)) throw RtError(RT_ERR_UNSUPPORTED, "rt_debug_hit_device has no ray time: scenes with moving spheres are not supported");
    if (!s.committed) throw RtError(RT_ERR_NOT_COMMITTED, "scene not committed");
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    FlatView view = s.flat.view;
    view.base = device_blob(s, dev);
    DevBuf dr, dout, err;
    dr.alloc(n * 48);
    dout.alloc(n * 96);
    err.alloc(4);
    HIP_CHECK(hipMemset(err.p, 0, 4));
    HIP_CHECK(hipMemcpy(dr.p, rays, n * 48, hipMemcpyHostToDevice));
    if (kernel < 1 || kernel > 6 || kernel == 4) throw RtError(RT_ERR_ARG, "rt_debug_hit_device: kernel 1, 2, 3, 5 or 6");
    if (kernel != 1 && !view.accel_ok) throw RtError(RT_ERR_UNSUPPORTED, "no accel for this scene");
    if (kernel != 1 && !(t_min >= 0.)) throw RtError(RT_ERR_UNSUPPORTED, "kernel 2 needs t_min >= 0 (box32)");
    if ((kernel == 5 || kernel == 6) && (view.coop_data_ok == 0 || view.n_inst2 < 1 || view.n_inst2 > (uint32_t)COOP_MAX_INST))
        throw RtError(RT_ERR_UNSUPPORTED, "the instance walks of kernels 5 / 6 need 1..64 instances of which at least one holds only f32-vertex triangles");
    if (view.kinds_mask & (1u << NK_MEDIUM_BEGIN))
        throw RtError(RT_ERR_UNSUPPORTED, "closest-hit queries on a scene with a ConstantMedium need the path's random stream");
    size_t smem = (kernel == 2 || kernel == 3) ? view.stack2 * 64 * sizeof(uint32_t) : 0;
    if (kernel == 5 || kernel == 6) smem = (size_t)std::max(std::max(view.world_depth2, view.inst_depth2) + 2u, view.stack2_inline) * 64 * sizeof(uint32_t);
    if (kernel == 3) {  // kernel 2's LDS node table (NodeW, box32w) in isolation: the table must fit beside the stacks
        smem += ((size_t)(view.n_nodes2 + NODEW_CHUNK - 1) / NODEW_CHUNK) * 3 * NODEW_FAR;
        if (smem > 160 * 1024) throw RtError(RT_ERR_UNSUPPORTED, "the NodeW table of this scene does not fit in LDS");
        if (smem > 48 * 1024) HIP_CHECK(hipFuncSetAttribute((const void*)hit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    }
    hipLaunchKernelGGL(hit_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), smem, 0, view,
                       kernel, n, (const double*)dr.p, t_min, t_max, (double*)dout.p, (int*)err.p);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpy(out, dout.p, n * 96, hipMemcpyDeviceToHost));
}

}  // namespace rtamd
