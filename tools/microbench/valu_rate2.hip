// valu_rate2.hip -- developer microbenchmark (not part of the product): issue cost of the VALU instruction classes the chain's kernels are made of.
// valu_rate.hip showed that on gfx950 a wave64 v_fma_f32 / v_mul_f32 / v_add_u32 issues every ~2.9 cycles, v_pk_fma_f32 every ~5.2 (packed fp32
// buys ~10 %, not 2x), v_max_f32 ~4.4, transcendentals ~8.5.  This one prices the rest (compares + selects, conversions, floor / fract, med3,
// integer multiply, 64-bit address arithmetic, div_fixup, ...) so that the static instruction mix of a kernel (tools/isa_stats.py) can be
// turned into an estimate of its issue time.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate2 valu_rate2.hip && ./valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kIters = 1024, kUnroll = 16;

#define BENCH_LIST(X)                                                                                              \
    X(0, 1, "v_fma_f32 (ref)", "v_fma_f32 %0, %0, %1, %2")                                                          \
    X(1, 1, "v_add_f32", "v_add_f32 %0, %0, %1")                                                                    \
    X(2, 1, "v_sub_f32", "v_sub_f32 %0, %0, %1")                                                                    \
    X(3, 1, "v_mul_f32", "v_mul_f32 %0, %0, %1")                                                                    \
    X(4, 1, "v_mac/fmac_f32", "v_fmac_f32 %0, %1, %2")                                                              \
    X(5, 1, "v_max_f32", "v_max_f32 %0, %0, %1")                                                                    \
    X(6, 1, "v_min_f32", "v_min_f32 %0, %0, %1")                                                                    \
    X(7, 1, "v_med3_f32", "v_med3_f32 %0, %0, %1, %2")                                                              \
    X(8, 1, "v_max3_f32", "v_max3_f32 %0, %0, %1, %2")                                                              \
    X(9, 2, "v_cmp_gt_f32 + v_cndmask (vcc)", "v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")          \
    X(10, 1, "v_cmp_gt_f32 vcc only", "v_cmp_gt_f32 vcc, %0, %1")                                                  \
    X(11, 2, "v_cmp_gt_f32 s[] + v_cndmask_e64", "v_cmp_gt_f32 s[20:21], %0, %1\n v_cndmask_b32 %0, %0, %2, s[20:21]") \
    X(12, 1, "v_cvt_i32_f32", "v_cvt_i32_f32 %0, %0")                                                              \
    X(13, 1, "v_cvt_f32_i32", "v_cvt_f32_i32 %0, %0")                                                              \
    X(14, 1, "v_floor_f32", "v_floor_f32 %0, %0")                                                                  \
    X(15, 1, "v_fract_f32", "v_fract_f32 %0, %0")                                                                  \
    X(16, 1, "v_cvt_flr_i32_f32", "v_cvt_flr_i32_f32 %0, %0")                                                      \
    X(17, 1, "v_med3_i32", "v_med3_i32 %0, %0, %1, %2")                                                            \
    X(18, 1, "v_max_i32", "v_max_i32 %0, %0, %1")                                                                  \
    X(19, 1, "v_min_u32", "v_min_u32 %0, %0, %1")                                                                  \
    X(20, 1, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1")                                                            \
    X(21, 1, "v_mul_u32_u24", "v_mul_u32_u24 %0, %0, %1")                                                          \
    X(22, 1, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %2")                                                      \
    X(23, 1, "v_lshlrev_b32", "v_lshlrev_b32 %0, 2, %0")                                                           \
    X(24, 1, "v_and_b32", "v_and_b32 %0, %0, %1")                                                                  \
    X(25, 1, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 2, %1")                                                     \
    X(26, 1, "v_add3_u32", "v_add3_u32 %0, %0, %1, %2")                                                            \
    X(27, 1, "v_div_fixup_f32", "v_div_fixup_f32 %0, %0, %1, %2")                                                  \
    X(28, 1, "v_rcp_f32", "v_rcp_f32 %0, %0")                                                                      \
    X(29, 1, "v_rsq_f32", "v_rsq_f32 %0, %0")                                                                      \
    X(30, 1, "v_sqrt_f32", "v_sqrt_f32 %0, %0")                                                                    \
    X(31, 1, "v_exp_f32", "v_exp_f32 %0, %0")                                                                      \
    X(32, 1, "v_log_f32", "v_log_f32 %0, %0")                                                                      \
    X(33, 1, "v_sin_f32", "v_sin_f32 %0, %0")                                                                      \
    X(34, 1, "v_mul_f32 literal", "v_mul_f32 %0, 0x3f7fbe77, %0")                                                  \
    X(35, 1, "v_fma_f32 inline const", "v_fma_f32 %0, %0, %1, 1.0")                                              \
    X(36, 1, "v_fma_f32 sgpr", "v_fma_f32 %0, %0, s20, %1")                                                        \
    X(37, 1, "v_mul_f32 e64 (abs)", "v_mul_f32 %0, |%0|, %1")                                                      \
    X(38, 1, "v_mov_b32", "v_mov_b32 %0, %1")                                                                      \
    X(39, 1, "v_mov_b32 dpp row_shr", "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")                 \
    X(40, 1, "v_mad_u64_u32", "v_mad_u64_u32 v[40:41], s[22:23], %0, %1, v[40:41]")                                \
    X(41, 2, "v_add_co_u32 + v_addc_co_u32", "v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %2, vcc")  \
    X(42, 1, "v_cmp_class_f32", "v_cmp_class_f32 vcc, %0, %1")                                                     \
    X(43, 1, "v_ldexp_f32", "v_ldexp_f32 %0, %0, %1")                                                              \
    X(44, 1, "v_bfe_u32", "v_bfe_u32 %0, %0, 3, 5")                                                                \
    X(45, 1, "v_perm_b32", "v_perm_b32 %0, %0, %1, %2")                                                            \
    X(46, 1, "v_cvt_f16_f32", "v_cvt_f16_f32 %0, %0")                                                              \
    X(47, 1, "v_cvt_pkrtz_f16_f32", "v_cvt_pkrtz_f16_f32 %0, %0, %1")                                              \
    X(48, 1, "v_pk_mul_f32", "v_pk_mul_f32 %3, %3, %4")                                                            \
    X(49, 1, "v_pk_add_f32", "v_pk_add_f32 %3, %3, %4")                                                            \
    X(50, 1, "v_dot2c_f32_f16", "v_dot2c_f32_f16 %0, %1, %2")                                                      \
    X(51, 1, "v_fma_f32 dependent chain", "v_fma_f32 %5, %5, %1, %2")                                              \
    X(52, 1, "v_rcp_f32 dependent chain", "v_rcp_f32 %5, %5")

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE> __global__ __launch_bounds__(256) void rate_kernel(float* out, float seed)
{
    float a[kUnroll];
    f2    p[kUnroll];
    for (int i = 0; i < kUnroll; ++i) { a[i] = seed + float(i) + float(threadIdx.x) * 1e-3f; p[i] = f2{a[i], a[i] + 1.f}; }
    float dep = seed;
    const float m = 0.999f + seed * 1e-9f, c = 1e-3f;
    const f2 m2{m, m};
    for (int it = 0; it < kIters; ++it)
    {
#pragma unroll
        for (int i = 0; i < kUnroll; ++i)
        {
#define X(ID, N, NAME, ASM) \
    if (MODE == ID) asm volatile(ASM : "+v"(a[i]) : "v"(m), "v"(c), "v"(p[i]), "v"(m2), "v"(dep) : "vcc", "s20", "s21", "s22", "s23", "v40", "v41");
            BENCH_LIST(X)
#undef X
        }
    }
    float s = dep;
    for (int i = 0; i < kUnroll; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> static double run(int instrPerSlot, int wavesPerSimd, float* out, int cus, double ghz)
{
    const int blocks = cus * wavesPerSimd;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    rate_kernel<MODE><<<blocks, 256>>>(out, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    rate_kernel<MODE><<<blocks, 256>>>(out, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instrPerSimd = double(kIters) * kUnroll * instrPerSlot * wavesPerSimd;
    return ms * 1e-3 * ghz * 1e9 / instrPerSimd;
}

int main()
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    printf("%s: %d CUs, nominal %.2f GHz; cycles per wave64 instruction at nominal clock, 1 / 4 / 8 waves per SIMD\n", prop.name, cus, ghz);
    float* out;
    (void)hipMalloc(&out, size_t(cus) * 8 * 256 * sizeof(float));
#define X(ID, N, NAME, ASM) printf("%-36s %6.2f %6.2f %6.2f\n", NAME, run<ID>(N, 1, out, cus, ghz), run<ID>(N, 4, out, cus, ghz), run<ID>(N, 8, out, cus, ghz));
    BENCH_LIST(X)
#undef X
    return 0;
}
