// Issue cost of VALU instructions on gfx950: time and shader cycles per wave64 instruction PER SIMD, at one wave per SIMD (256-thread
// workgroup per CU) and at the occupancy of the path-tracing kernels (1024 threads per CU = 4 waves per SIMD; all waves of a SIMD run the
// same stream of independent instructions).  Round 5: every measurement is a dispatch of >= 20 ms (the iteration count is calibrated
// per instruction; round 4's 0.2-0.7 ms dispatches left the absolute scale open), timed three ways -- HIP events around the dispatch,
// s_memrealtime (the constant 100 MHz counter) and s_memtime inside the kernel -- and the cycles come from GRBM_GUI_ACTIVE / 8 of the
// same dispatches (run the binary under rocprofv3 --pmc GRBM_GUI_ACTIVE; tools/microbench/valu_cost_cycles.py joins the two outputs).
// Build / run: hipcc --offload-arch=gfx950 -O2 tools/microbench/valu_cost.hip -o tools/microbench/valu_cost.bin && tools/microbench/valu_cost.bin
// Basis of the roofline's cycle table (tools/make_pt_model.py).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int OP, int WPS>  // WPS: waves per SIMD = blockDim / 256 (part of the kernel's name for the counter pass)
__global__ void __launch_bounds__(1024) k(unsigned long long* out, int iters, double seed) {
    double a = seed + threadIdx.x, b = seed * 3.0 + 1.0, c0 = 1.0, c1 = 2.0, c2 = 3.0, c3 = 4.0;
    float fa = (float)a, fb = (float)b, f0 = 1.f, f1 = 2.f, f2 = 3.f, f3 = 4.f;
    unsigned int ua = (unsigned)threadIdx.x * 2654435761u + 12345u, u0 = 1, u1 = 2, u2 = 3, u3 = 4;
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { REP64(asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fa), "v"(fb));) }
        if (OP == 1) { REP64(asm volatile("v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %1, %4, %5, %1\n v_fma_f64 %2, %4, %5, %2\n v_fma_f64 %3, %4, %5, %3" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));) }
        if (OP == 2) { REP64(asm volatile("v_min_f32 %0, %4, %0\n v_max_f32 %1, %4, %1\n v_min_f32 %2, %5, %2\n v_max_f32 %3, %5, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fa), "v"(fb));) }
        if (OP == 3) { REP64(asm volatile("v_rcp_f64 %0, %4\n v_rcp_f64 %1, %5\n v_rcp_f64 %2, %4\n v_rcp_f64 %3, %5" : "=v"(c0), "=v"(c1), "=v"(c2), "=v"(c3) : "v"(a), "v"(b));) }
        if (OP == 4) { REP64(asm volatile("v_sqrt_f64 %0, %4\n v_rsq_f64 %1, %5\n v_sqrt_f64 %2, %4\n v_rsq_f64 %3, %5" : "=v"(c0), "=v"(c1), "=v"(c2), "=v"(c3) : "v"(a), "v"(b));) }
        if (OP == 5) { REP64(asm volatile("v_mul_lo_u32 %0, %4, %0\n v_mul_lo_u32 %1, %4, %1\n v_mul_lo_u32 %2, %4, %2\n v_mul_lo_u32 %3, %4, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(ua));) }
        if (OP == 6) { REP64(asm volatile("v_xor_b32 %0, %4, %0\n v_add_u32 %1, %4, %1\n v_lshlrev_b32 %2, 3, %2\n v_alignbit_b32 %3, %3, %3, 5" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(ua));) }
        if (OP == 7) { REP64(asm volatile("v_cvt_f64_u32 %0, %4\n v_cvt_f32_f64 %2, %5\n v_cvt_f64_u32 %1, %4\n v_cvt_f32_u32 %3, %4" : "=v"(c0), "=v"(c1), "=v"(f0), "=v"(f1) : "v"(ua), "v"(a));) }
        if (OP == 8) { REP64(asm volatile("v_rcp_f32 %0, %4\n v_rcp_f32 %1, %5\n v_sqrt_f32 %2, %4\n v_rsq_f32 %3, %5" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(fa), "v"(fb));) }
        if (OP == 9) { REP64(asm volatile("v_add_f64 %0, %4, %0\n v_mul_f64 %1, %4, %1\n v_add_f64 %2, %5, %2\n v_mul_f64 %3, %5, %3" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));) }
        if (OP == 10) { REP64(asm volatile("v_cmp_lt_f32 vcc, %4, %0\n v_cndmask_b32 %1, %4, %1, vcc\n v_cmp_lt_f64 vcc, %6, %7\n v_cndmask_b32 %3, %5, %3, vcc" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fa), "v"(fb), "v"(a), "v"(b) : "vcc");) }
        if (OP == 11) { REP64(asm volatile("v_ldexp_f64 %0, %4, 3\n v_div_fixup_f64 %1, %4, %5, %1\n v_ldexp_f64 %2, %5, 2\n v_div_fixup_f64 %3, %5, %4, %3" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));) }
        // round 4: the members of the model's "other" class that round 3 priced at the class average without measuring them
        if (OP == 12) { REP64(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %4\n v_mov_b32 %3, %5" : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3) : "v"(ua), "v"(fa));) }
        if (OP == 13) { REP64(asm volatile("v_writelane_b32 %0, s20, 3\n v_readlane_b32 s21, %1, 5\n v_writelane_b32 %2, s20, 7\n v_readlane_b32 s22, %3, 9" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : : "s20", "s21", "s22");) }
        if (OP == 14) { REP64(asm volatile("v_min3_f32 %0, %4, %5, %0\n v_max3_f32 %1, %4, %5, %1\n v_min3_f32 %2, %5, %4, %2\n v_max3_f32 %3, %5, %4, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fa), "v"(fb));) }
        if (OP == 15) { REP64(asm volatile("v_and_b32 %0, %4, %0\n v_or_b32 %1, %4, %1\n v_lshl_add_u32 %2, %4, 2, %2\n v_bfe_u32 %3, %4, 3, 9" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(ua));) }
        if (OP == 16) { REP64(asm volatile("v_cndmask_b32_e64 %0, %4, %0, s[20:21]\n v_cndmask_b32_e64 %1, %4, %1, s[22:23]\n v_cndmask_b32_e64 %2, %4, %2, s[20:21]\n v_cndmask_b32_e64 %3, %4, %3, s[22:23]" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(ua) : "s20", "s21", "s22", "s23");) }
        if (OP == 17) { REP64(asm volatile("v_cmp_lt_f32 vcc, %4, %0\n v_cmp_gt_u32 vcc, %5, %1\n v_cmp_nlt_f32 s[20:21], %4, %2\n v_cmp_ngt_f32 s[22:23], %4, %3" : : "v"(f0), "v"(u1), "v"(f2), "v"(f3), "v"(fa), "v"(ua) : "vcc", "s20", "s21", "s22", "s23");) }
        if (OP == 18) { REP64(asm volatile("v_add_co_u32 %0, vcc, %4, %0\n v_addc_co_u32 %1, vcc, %4, %1, vcc\n v_add_co_u32 %2, vcc, %4, %2\n v_addc_co_u32 %3, vcc, %4, %3, vcc" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(ua) : "vcc");) }
        if (OP == 19) { REP64(asm volatile("v_cvt_f32_u32_sdwa %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(ua));) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = t1 - t0;
        out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0;
    }
    if (c0 + c1 + c2 + c3 + f0 + f1 + f2 + f3 + (double)(u0 ^ u1 ^ u2 ^ u3) == 1.2345) out[0] = 1;
}
template <int OP, int WPS>
static void run_one(const char* name) {
    const int blocks = 256, threads = 256 * WPS;
    unsigned long long* d;
    hipMalloc(&d, blocks * 16 * 2 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // calibration (short), then the measurement: as many iterations as make the dispatch last >= 20 ms
    int iters = 200;
    float ms = 0;
    for (int pass = 0; pass < 2; pass++) {
        hipMemset(d, 0, blocks * 16 * 2 * 8);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<OP, WPS>), dim3(blocks), dim3(threads), 0, 0, d, iters, 1.5);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        if (pass == 0) iters = (int)(iters * 25.0 / (ms > 1e-3f ? ms : 1e-3f)) + 1;
    }
    std::vector<unsigned long long> h(blocks * 16 * 2);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double mt = 0, rt = 0;
    const int waves = blocks * threads / 64;
    for (int b = 0; b < blocks; b++)
        for (int w = 0; w < threads / 64; w++) {
            mt += (double)h[(b * 16 + w) * 2];
            rt += (double)h[(b * 16 + w) * 2 + 1];
        }
    mt /= waves;
    rt /= waves;
    const double n_inst = (double)iters * 64 * 4;  // per wave; WPS waves share a SIMD
    // one line per measurement, machine-readable: valu_cost_cycles.py joins it with the GRBM_GUI_ACTIVE of the dispatch named k<OP, WPS>
    printf("VALU_COST op=%d wps=%d name=\"%s\" iters=%d inst_per_wave=%.0f event_ms=%.3f realtime_ns=%.0f memtime_ticks=%.0f ns_per_inst_per_simd_event=%.4f ns_per_inst_per_simd_realtime=%.4f "
           "memtime_ticks_per_ns=%.4f\n",
           OP, WPS, name, iters, n_inst, ms, rt * 10.0, mt, ms * 1e6 / (WPS * n_inst), rt * 10.0 / (WPS * n_inst), mt / (rt * 10.0));
    hipFree(d);
}
template <int OP>
static void run(const char* name) {
    run_one<OP, 1>(name);
    run_one<OP, 4>(name);
}
int main() {
    run<0>("v_fma_f32");
    run<1>("v_fma_f64");
    run<9>("v_add_f64 / v_mul_f64");
    run<2>("v_min_f32 / v_max_f32");
    run<6>("v_xor / v_add_u32 / v_lshlrev / v_alignbit");
    run<10>("v_cmp_lt_f32|f64 + v_cndmask");
    run<5>("v_mul_lo_u32");
    run<7>("v_cvt_f64_u32 / v_cvt_f32_f64 / v_cvt_f32_u32");
    run<11>("v_ldexp_f64 / v_div_fixup_f64");
    run<8>("v_rcp_f32 / v_sqrt_f32 / v_rsq_f32");
    run<3>("v_rcp_f64");
    run<4>("v_sqrt_f64 / v_rsq_f64");
    run<12>("v_mov_b32");
    run<13>("v_writelane_b32 / v_readlane_b32 (SGPR spill moves)");
    run<14>("v_min3_f32 / v_max3_f32");
    run<15>("v_and / v_or / v_lshl_add_u32 / v_bfe_u32");
    run<16>("v_cndmask_b32_e64 (SGPR-pair mask)");
    run<17>("v_cmp f32 / u32 (to vcc and to SGPR pairs)");
    run<18>("v_add_co_u32 / v_addc_co_u32 (64-bit add)");
    run<19>("v_cvt_f32_u32 SDWA (16-bit halves)");
    return 0;
}
