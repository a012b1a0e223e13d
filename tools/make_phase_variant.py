"""Diagnostic build of the path-tracing kernel with per-phase wave clocks and event counters.

    python tools/make_phase_variant.py      -> rust-raytracer_amd/variants/librtamd_phase.so
    RTAMD_LIB=$PWD/rust-raytracer_amd/variants/librtamd_phase.so python bench.py --steps 1 --warmup 0 --spp 46 --cpu-spp 0

The script patches a COPY of csrc/device/kernels.hip (the product source stays free of instrumentation) and
every render then prints to stderr
  PHASE  share of wave time (s_memtime) in regeneration / traversal (leaf part) / materialize+shade+store
  EV     per outer-loop iteration: how often the WAVE executed a block, how many LANES took part, utilisation
for the inner-node step, leaf item tests, the rejection loops, the material branches and regeneration.
Numbers of round 1 (scene_500): 26.9 inner steps per iteration at 33 % lane utilisation, 5.8 leaf items at 38 %; round 2: DESIGN.md s5.
(works on the ACCEL == 2 kernels; scenes that fall back to kernel 1 are not instrumented)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "rust-raytracer_amd")
SRC = os.path.join(PKG, "csrc", "device", "kernels.hip")
DST = os.path.join(PKG, "csrc", "device", "kernels_phase.hip")

s = open(SRC).read()


def rep(a, b):
    global s
    assert s.count(a) >= 1, a
    s = s.replace(a, b, 1)


rep("DEV D3 mk(double x, double y, double z)",
    "DEV void cntev(unsigned long long* ph, int i) {\n    ph[i] += 1;\n    unsigned long long m = __ballot(1);\n"
    "    if ((int)(__ffsll((long long)m) - 1) == (int)(threadIdx.x & 63)) ph[16 + i] += 1;\n}\nDEV D3 mk(double x, double y, double z)")
rep("// Closest hit through the accel (common/flat.h)",
    "__device__ unsigned long long g_phase[32];\n#define PH_NOW() __builtin_readcyclecounter()\n// Closest hit through the accel (common/flat.h)")
rep("DEV D3 random_in_unit_sphere(Rng& rng) {  // vec3.rs:111-129 (Marsaglia; a point ON the sphere, Q3)\n    double u, v, r2;\n    for (;;) {",
    "DEV D3 random_in_unit_sphere(Rng& rng, unsigned long long* c = nullptr) {\n    double u, v, r2;\n    for (;;) {\n        if (c) cntev(c, 0);")
rep("DEV D3 random_in_unit_disk(Rng& rng) {  // vec3.rs:153-162\n    for (;;) {",
    "DEV D3 random_in_unit_disk(Rng& rng, unsigned long long* c = nullptr) {\n    for (;;) {\n        if (c) cntev(c, 0);")
rep("DEV Hit traverse2(const Acc& A, uint32_t* stk, const int stride, D3 wo, D3 wd, double t_min, double t_max) {",
    "DEV Hit traverse2(const Acc& A, uint32_t* stk, const int stride, D3 wo, D3 wd, double t_min, double t_max, unsigned long long* ph = nullptr) {")
rep("        while ((cur >> REF_TAG_SHIFT) == 0u) {  // inner node: test both children\n",
    "        while ((cur >> REF_TAG_SHIFT) == 0u) {  // inner node: test both children\n            if (ph) cntev(ph, 14);\n")
rep("        if (cur == REF_DONE) break;\n        if ((cur >> REF_TAG_SHIFT) == 1u) {  // leaf: test its items",
    "        if (cur == REF_DONE) break;\n        unsigned long long tl0 = PH_NOW();\n        if ((cur >> REF_TAG_SHIFT) == 1u) {  // leaf: test its items")
rep("            for (uint32_t i = 0; i < cnt; i++) {\n                uint2 it = A.items2[first + i];",
    "            for (uint32_t i = 0; i < cnt; i++) {\n                if (ph) cntev(ph, 15);\n                uint2 it = A.items2[first + i];")
rep("""        if (sp > 0) {
            sp -= stride;
            cur = stk[sp];
        } else {
            cur = REF_DONE;
        }
    }
    return h;""", """        if (sp > 0) {
            sp -= stride;
            cur = stk[sp];
        } else {
            cur = REF_DONE;
        }
        if (ph) ph[5] += PH_NOW() - tl0;
    }
    return h;""")
rep("DEV bool shade(const Acc& A, const Rec& rec, D3 rdir, Rng& rng, D3& emitted, D3& att, D3& out_dir, bool& diffuse, int* err) {",
    "DEV bool shade(const Acc& A, const Rec& rec, D3 rdir, Rng& rng, D3& emitted, D3& att, D3& out_dir, bool& diffuse, int* err, unsigned long long* ph = nullptr) {")
rep("    if (type != 2) rs = random_in_unit_sphere(rng);",
    "    if (ph && lamb) cntev(ph, 10);\n    if (ph && type == 1) cntev(ph, 11);\n    if (ph && type == 2) cntev(ph, 12);\n    if (type != 2) rs = random_in_unit_sphere(rng, ph ? ph + 9 : nullptr);")
# pt_kernel
rep("""    int tx = 0, ty = 0, s0 = 0, pool = 0, next = 0, cur_slot = 0;
    bool finished = false;
""", """    int tx = 0, ty = 0, s0 = 0, pool = 0, next = 0, cur_slot = 0;
    bool finished = false;
    unsigned long long ph[32];
    for (int i = 0; i < 32; i++) ph[i] = 0;
""")
rep("""        for (;;) {
            // ---- regeneration: dead lanes pull the next (pixel, sample) of the pool ----
            uint64_t dead = __ballot(!alive);""", """        for (;;) {
            unsigned long long t0 = PH_NOW();
            // ---- regeneration: dead lanes pull the next (pixel, sample) of the pool ----
            uint64_t dead = __ballot(!alive);""")
rep("                if (!alive && k < pool) {\n                    int pix = k & (TILE_PIX - 1), s = s0 + (k >> 6);",
    "                cntev(ph, 8);\n                if (!alive && k < pool) {\n                    int pix = k & (TILE_PIX - 1), s = s0 + (k >> 6);")
rep("                        D3 rd = muls(random_in_unit_disk(rng), cam.lens_radius);  // drawn even for aperture 0 (Q4)",
    "                        D3 rd = muls(random_in_unit_disk(rng, ph + 13), cam.lens_radius);")
rep("""            // ---- one path segment: sample_ray's loop body, photon_mapper.rs:335-362 ----
            if (alive) {
                Hit h = (ACCEL == 2) ? traverse2<GENERAL>(A, stk, stk_stride, o, d, rk.t_min, INFINITY)
                                     : traverse<GENERAL, MEDIA>(A, o, d, rk.t_min, INFINITY, &rng);
                bool done = true;""", """            unsigned long long t1 = PH_NOW();
            ph[0] += t1 - t0;
            ph[6] += 1;
            Hit h;
            if (alive) h = traverse2<GENERAL>(A, stk, stk_stride, o, d, rk.t_min, INFINITY, ph);
            unsigned long long t2 = PH_NOW();
            ph[1] += t2 - t1;
            if (alive) {
                bool done = true;""")
rep("""                    Rec rec = materialize<GENERAL>(A, h, o, d, err);
                    D3 emitted, att, ndir;
                    bool diffuse;
                    bool scattered = shade(A, rec, d, rng, emitted, att, ndir, diffuse, err);""", """                    unsigned long long t3 = PH_NOW();
                    Rec rec = materialize<GENERAL>(A, h, o, d, err);
                    D3 emitted, att, ndir;
                    bool diffuse;
                    unsigned long long t4 = PH_NOW();
                    bool scattered = shade(A, rec, d, rng, emitted, att, ndir, diffuse, err, ph);
                    unsigned long long t5 = PH_NOW();
                    ph[2] += t4 - t3;
                    ph[3] += t5 - t4;""")
rep("""                    atomicSub(rmeta + 4 * (out_slot >> 9) + 3, 1u);  // one path less running in that ring slot (UNIT_SPP * 64 = 512 per slot)
                    alive = false;
                }
            }
        }
    }
}
""", """                    atomicSub(rmeta + 4 * (out_slot >> 9) + 3, 1u);
                    alive = false;
                }
            }
            ph[4] += PH_NOW() - t2;
        }
    }
    for (int i = 0; i < 32; i++)
        if (i >= 8 || lane == 0) atomicAdd(&g_phase[i], ph[i]);
}
""")
rep("""    HIP_CHECK(hipStreamSynchronize(stream));
    if (st) {
        double kms = 0;""", """    HIP_CHECK(hipStreamSynchronize(stream));
    {
        unsigned long long hp[32] = {0};
        HIP_CHECK(hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_phase), sizeof(hp)));
        double tot = (double)(hp[0] + hp[1] + hp[4]), it = (double)hp[6];
        fprintf(stderr, "PHASE regen %.3f trav %.3f (leaf-part %.3f) post %.3f [mat %.3f shade %.3f] iters %llu cyc/iter %.0f\\n", hp[0] / tot,
                hp[1] / tot, hp[5] / tot, hp[4] / tot, hp[2] / tot, hp[3] / tot, hp[6], tot / it);
        const char* nm[8] = {"regen", "sphere-iters", "lambert", "metal", "dielectric", "disk-iters", "inner", "leafitems"};
        for (int i = 0; i < 8; i++)
            fprintf(stderr, "EV %-12s wave-exec/iter %.3f lane-exec/iter %.3f util %.3f\\n", nm[i], hp[24 + i] / it, hp[8 + i] / it,
                    hp[8 + i] / (64. * hp[24 + i]));
        unsigned long long z[32] = {0};
        HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)));
    }
    if (st) {
        double kms = 0;""")
open(DST, "w").write(s)
os.makedirs(os.path.join(PKG, "variants"), exist_ok=True)
hipcc = "/opt/rocm/bin/hipcc"
flags = "-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../include -Icsrc --offload-arch=gfx950".split()
try:
    subprocess.check_call([hipcc] + flags + ["-c", "csrc/device/kernels_phase.hip", "-o", "variants/kernels_phase.o"], cwd=PKG)
    host = [os.path.join("csrc", "abi.o")] + sorted(os.path.join("csrc", "host", f) for f in os.listdir(os.path.join(PKG, "csrc", "host")) if f.endswith(".o"))
    subprocess.check_call([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", "variants/librtamd_phase.so"] + host + ["variants/kernels_phase.o"], cwd=PKG)
finally:
    os.remove(DST)
    if os.path.exists(os.path.join(PKG, "variants", "kernels_phase.o")):
        os.remove(os.path.join(PKG, "variants", "kernels_phase.o"))
print("built rust-raytracer_amd/variants/librtamd_phase.so", file=sys.stderr)
