// valu_rate.hip -- developer microbenchmark (not part of the product): issue rate of plain and packed fp32 VALU instructions on one device.
// Answers: how many cycles does a wave64 v_fma_f32 / v_pk_fma_f32 / v_rcp_f32 / v_max_f32 occupy a SIMD's issue slot?  (The chain's kernels are
// VALU-issue bound: this sets what "two pixels per lane with packed math" can buy.)
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kIters = 2048, kUnroll = 16;

template <int MODE> __global__ __launch_bounds__(256) void rate_kernel(float* out, float seed)
{
    float a[kUnroll];
    f2    p[kUnroll / 2];
    for (int i = 0; i < kUnroll; ++i) a[i] = seed + float(i) + float(threadIdx.x) * 1e-3f;
    for (int i = 0; i < kUnroll / 2; ++i) p[i] = f2{a[2 * i], a[2 * i + 1]};
    const float m = 0.999f + seed * 1e-9f, c = 1e-3f;
    const f2 m2{m, m}, c2{c, c};
    for (int it = 0; it < kIters; ++it)
    {
#pragma unroll
        for (int i = 0; i < kUnroll; ++i)
        {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (MODE == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (MODE == 3) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
            if (MODE == 6) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            if (MODE == 7) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (MODE == 8) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 9) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        }
        if (MODE == 1)
        {
#pragma unroll
            for (int i = 0; i < kUnroll / 2; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
        }
        if (MODE == 10) // half packed, half plain (same instruction count as MODE 1 x 2)
        {
#pragma unroll
            for (int i = 0; i < kUnroll / 4; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
#pragma unroll
            for (int i = 0; i < kUnroll / 2; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        }
    }
    float s = 0.f;
    for (int i = 0; i < kUnroll; ++i) s += a[i];
    for (int i = 0; i < kUnroll / 2; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> static void run(const char* name, int instrPerIter, int wavesPerSimd, float* out, int cus, double ghz)
{
    const int blocks = cus * wavesPerSimd; // 256 threads = 4 waves = one per SIMD; wavesPerSimd blocks per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<MODE><<<blocks, 256>>>(out, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate_kernel<MODE><<<blocks, 256>>>(out, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double instrPerSimd = double(kIters) * instrPerIter * wavesPerSimd; // wave-instructions issued by each SIMD
    const double cycles = ms * 1e-3 * ghz * 1e9;
    printf("%-28s waves/SIMD %d  %8.3f ms  %6.2f cycles per wave-instruction (at %.2f GHz)\n", name, wavesPerSimd, ms, cycles / instrPerSimd, ghz);
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    printf("%s: %d CUs, %.2f GHz\n", prop.name, cus, ghz);
    float* out;
    hipMalloc(&out, size_t(cus) * 8 * 256 * sizeof(float));
    for (int w : {1, 2, 4, 8})
    {
        run<0>("v_fma_f32", kUnroll, w, out, cus, ghz);
        run<1>("v_pk_fma_f32", kUnroll / 2, w, out, cus, ghz);
        run<10>("pk_fma + fma (1:2)", kUnroll / 4 + kUnroll / 2, w, out, cus, ghz);
        run<4>("v_mul_f32", kUnroll, w, out, cus, ghz);
        run<3>("v_max_f32", kUnroll, w, out, cus, ghz);
        run<5>("v_cndmask_b32", kUnroll, w, out, cus, ghz);
        run<8>("v_add_u32", kUnroll, w, out, cus, ghz);
        run<9>("v_mad_u32_u24", kUnroll, w, out, cus, ghz);
        run<2>("v_rcp_f32", kUnroll, w, out, cus, ghz);
        run<6>("v_sqrt_f32", kUnroll, w, out, cus, ghz);
        run<7>("v_exp_f32", kUnroll, w, out, cus, ghz);
    }
    return 0;
}
