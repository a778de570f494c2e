// valu_rate3.hip -- developer microbenchmark (not part of the product): what one wave64 VALU instruction costs a SIMD of this device, in nanoseconds and in
// shader-clock cycles, measured so that the figure can be audited (round 3; replaces valu_rate.hip / valu_rate2.hip, whose 0.1 - 0.3 ms launches were converted with the
// nominal clock and carried the clock ramp):
//   * every variant runs for >= 50 ms (the iteration count is calibrated per variant), twice; both runs are printed with their spread;
//   * the shader clock is measured during the run: s_memtime (the shader-clock counter) against s_memrealtime (the constant 100 MHz reference) over the kernel's
//     lifetime on one wave, and the ratio is printed per run -- if s_memtime turns out to tick at the reference rate on this part the ratio reads 100 MHz and the
//     cycles column is flagged;
//   * the grid oversubscribes the device (32 x the resident capacity at the chosen occupancy) so that every SIMD holds its W waves for the whole run whatever the
//     dispatcher's placement; occupancy W waves per SIMD is set by the LDS a 256-thread workgroup reserves (160 KB / W);
//   * issue cost = elapsed time x number of SIMDs / wave-instructions issued.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate3 valu_rate3.hip && ./valu_rate3 [min_ms]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kUnroll = 16; // independent dependency chains per lane

enum Op
{
    FMA, MUL, ADD, MOV, AND, CMP_CNDMASK, CNDMASK, MIN, MAX, MED3, MIN3, CVT_I32_F32, CVT_F32_I32, FLOOR, FRACT, LSHL, ADD_U32, LSHL_ADD, MUL_LO, MAD_U24, DIV_FIXUP,
    RCP, SQRT, EXP, LOG, PK_FMA, PK_MUL, FMA_SGPR, MUL_LITERAL, OP_COUNT
};
static const char* kNames[OP_COUNT] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_mov_b32", "v_and_b32", "v_cmp_lt_f32 + v_cndmask_b32 (per pair)", "v_cndmask_b32 (vcc)", "v_min_f32",
                                       "v_max_f32", "v_med3_f32", "v_min3_f32", "v_cvt_i32_f32", "v_cvt_f32_i32", "v_floor_f32", "v_fract_f32", "v_lshlrev_b32", "v_add_u32",
                                       "v_lshl_add_u32", "v_mul_lo_u32", "v_mad_u32_u24", "v_div_fixup_f32", "v_rcp_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32",
                                       "v_pk_fma_f32 (2 fp32 lanes)", "v_pk_mul_f32 (2 fp32 lanes)", "v_fma_f32 with an SGPR operand", "v_mul_f32 with a 32-bit literal"};

typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP> __global__ __launch_bounds__(256) void rate_kernel(float* out, unsigned long long* clocks, float seed, int iters)
{
    extern __shared__ float occupancyPad[]; // reserves LDS: sets the number of resident workgroups per CU
    float a[kUnroll];
    f2    p[kUnroll / 2];
    for (int i = 0; i < kUnroll; ++i) a[i] = seed + float(i) + float(threadIdx.x) * 1e-3f;
    for (int i = 0; i < kUnroll / 2; ++i) p[i] = f2{a[2 * i], a[2 * i + 1]};
    const float m = 0.999f + seed * 1e-9f, c = 1e-3f;
    const f2    m2{m, m}, c2{c, c};
    float sm = m; // uniform: lives in an SGPR
    asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sm) : "v"(m));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int i = 0; i < kUnroll; ++i)
        {
            if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (OP == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == CMP_CNDMASK) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(m), "v"(c) : "vcc");
            if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
            if (OP == MIN) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (OP == MIN3) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (OP == CVT_I32_F32) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
            if (OP == CVT_F32_I32) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]));
            if (OP == FLOOR) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
            if (OP == FRACT) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
            if (OP == LSHL) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
            if (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(m));
            if (OP == MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == MAD_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (OP == DIV_FIXUP) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (OP == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (OP == SQRT) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            if (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (OP == LOG) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
            if (OP == FMA_SGPR) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sm), "v"(c));
            if (OP == MUL_LITERAL) asm volatile("v_mul_f32 %0, 0x3f7fbe77, %0" : "+v"(a[i]));
        }
        if (OP == PK_FMA)
        {
#pragma unroll
            for (int i = 0; i < kUnroll / 2; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
        }
        if (OP == PK_MUL)
        {
#pragma unroll
            for (int i = 0; i < kUnroll / 2; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < kUnroll; ++i) s += a[i];
    for (int i = 0; i < kUnroll / 2; ++i) s += p[i].x + p[i].y;
    out[(blockIdx.x % 4096u) * blockDim.x + threadIdx.x] = s + occupancyPad[0] * 0.0f;
    if (blockIdx.x == gridDim.x / 2u && threadIdx.x == 0u) { clocks[0] = t1 - t0; clocks[1] = r1 - r0; }
}

struct Result { double ms, ns_per_instr, mhz; };

template <int OP> static Result run_once(int wavesPerSimd, int iters, int cus, float* out, unsigned long long* clocks)
{
    const int    instrPerIter = (OP == PK_FMA || OP == PK_MUL) ? kUnroll / 2 : kUnroll; // (CMP_CNDMASK: one pair counted as one)
    const size_t lds          = size_t(160 * 1024 / wavesPerSimd) - 1024;               // W workgroups of 4 waves per CU = W waves per SIMD
    const int    blocks       = cus * wavesPerSimd * 32;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rate_kernel<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    rate_kernel<OP><<<blocks, 256, lds>>>(out, clocks, 1.0f, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2] = {0, 0};
    hipMemcpy(h, clocks, sizeof(h), hipMemcpyDeviceToHost);
    hipEventDestroy(e0); hipEventDestroy(e1);
    const double waveInstr = double(blocks) * 4.0 * double(iters) * instrPerIter;
    Result r;
    r.ms           = ms;
    r.ns_per_instr = double(ms) * 1e6 * (double(cus) * 4.0) / waveInstr;
    r.mhz          = h[1] ? double(h[0]) / double(h[1]) * 100.0 : 0.0;
    return r;
}

template <int OP> static void run(int wavesPerSimd, double minMs, int cus, float* out, unsigned long long* clocks)
{
    int iters = 64;
    Result r  = run_once<OP>(wavesPerSimd, iters, cus, out, clocks); // warm-up + calibration
    r         = run_once<OP>(wavesPerSimd, iters, cus, out, clocks);
    iters     = int(std::min(4.0e6, std::max(64.0, iters * minMs / std::max(r.ms, 1e-3) * 1.1)));
    const Result a = run_once<OP>(wavesPerSimd, iters, cus, out, clocks), b = run_once<OP>(wavesPerSimd, iters, cus, out, clocks);
    const double ns = 0.5 * (a.ns_per_instr + b.ns_per_instr), mhz = 0.5 * (a.mhz + b.mhz);
    printf("%-42s W=%d  %7.1f / %7.1f ms   %6.3f / %6.3f ns per wave-instruction per SIMD (spread %4.1f %%)   clock %6.0f / %6.0f MHz   %5.2f cycles%s\n", kNames[OP],
           wavesPerSimd, a.ms, b.ms, a.ns_per_instr, b.ns_per_instr, 100.0 * std::abs(a.ns_per_instr - b.ns_per_instr) / ns, a.mhz, b.mhz, ns * mhz * 1e-3,
           mhz < 150.0 ? "  (s_memtime ticks at the reference rate here: the cycles column is not shader cycles)" : "");
    fflush(stdout);
}

template <int OP> static void sweep(double minMs, int cus, float* out, unsigned long long* clocks)
{
    if constexpr (OP < OP_COUNT)
    {
        for (int w : {8, 4})
            if (w == 8 || OP == FMA || OP == MUL || OP == RCP || OP == MIN) run<OP>(w, minMs, cus, out, clocks);
        if (OP == FMA) { run<OP>(2, minMs, cus, out, clocks); run<OP>(1, minMs, cus, out, clocks); }
        sweep<OP + 1>(minMs, cus, out, clocks);
    }
}

int main(int argc, char** argv)
{
    const double minMs = argc > 1 ? atof(argv[1]) : 50.0;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    int wallKhz = 0;
    hipDeviceGetAttribute(&wallKhz, hipDeviceAttributeWallClockRate, 0);
    printf("%s: %d CUs (%d SIMDs), nominal shader clock %.0f MHz, s_memrealtime reference %d kHz; every line: two runs of >= %.0f ms\n", prop.name, cus, cus * 4, prop.clockRate * 1e-3,
           wallKhz, minMs);
    float* out;
    unsigned long long* clocks;
    hipMalloc(&out, size_t(4096) * 256 * sizeof(float));
    hipMalloc(&clocks, 2 * sizeof(unsigned long long));
    sweep<0>(minMs, cus, out, clocks);
    return 0;
}
