// mix_valu_copy.hip -- developer microbenchmark (not part of the product; round 6).  The chain's measurements say "two kinds of work on the GPU at once cost their sum"
// (lanes, CU masks, the role-interleaved A3 + copy grid).  This program asks the question with the two PUREST kinds of work there are, so that the
// answer separates the chip from the chain's kernels:
//     V   a register-only v_fma_f32 loop (no memory instruction at all), W waves per SIMD on every SIMD of the device
//     C   a streaming copy (16 bytes per lane and access, 8 independent loads in flight per lane), grid-stride, FOUR workgroups per CU (16 of a CU's 32 wave slots)
//     G   a gather loop: every lane reads 4 bytes from a pseudo-random line of an L2-resident table (the vector L1's tag look-ups, nothing else), four workgroups per CU
//   (C and G are one wave of workgroups each, so that two of the three kinds fit a CU together: nothing waits for a wave slot of the other)
// Each alone, then V beside C, V beside G and G beside C on two streams -- with the shader clock measured INSIDE the V kernel (s_memtime against the constant 100 MHz s_memrealtime)
// in every case.  If V slows down beside C although they share no pipeline, and its measured clock drops with it, the cause is the power / clock management of the part,
// not a resource of the CU; if its clock holds and it still slows down, the two share an issue resource; if nothing slows down, pure kinds of work DO overlap and the
// chain's kernels lose to each other in the memory pipeline.
//   hipcc --offload-arch=gfx950 -O3 -o mix_valu_copy mix_valu_copy.hip && ./mix_valu_copy [valu_waves_per_simd = 4] [copy_megabytes = 4096]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                  \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); std::exit(1); } \
    } while (0)

constexpr int kChains = 16;

// clocks[2 * wg + 0 / 1]: s_memtime / s_memrealtime deltas of wave 0 of the workgroup (every workgroup: the host takes the median)
__global__ __launch_bounds__(256) void valu_kernel(float* out, unsigned long long* clocks, float seed, int iters)
{
    extern __shared__ float occupancyPad[]; // reserves LDS: bounds the workgroups a CU holds (the copy's waves need slots beside these)
    float a[kChains];
    for (int i = 0; i < kChains; ++i) a[i] = seed + float(i) + float(threadIdx.x) * 1e-3f;
    const float m = 0.999f + seed * 1e-9f, c = 1e-3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int i = 0; i < kChains; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
    for (int i = 0; i < kChains; ++i) s += a[i];
    if (s == 12345.678f) out[threadIdx.x] = s + occupancyPad[0];
    if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kDepth = 8;
__global__ __launch_bounds__(256) void copy_kernel(const f4* __restrict__ src, f4* __restrict__ dst, size_t n) // n: a multiple of 256 * kDepth
{
    const size_t stride = size_t(gridDim.x) * 256u * kDepth;
    for (size_t base = size_t(blockIdx.x) * 256u * kDepth + threadIdx.x; base < n; base += stride)
    {
        f4 v[kDepth];
#pragma unroll
        for (int j = 0; j < kDepth; ++j) v[j] = __builtin_nontemporal_load(src + base + 256u * unsigned(j));
#pragma unroll
        for (int j = 0; j < kDepth; ++j) __builtin_nontemporal_store(v[j], dst + base + 256u * unsigned(j));
    }
}

// every lane of a wave on a line of its own (64 tag look-ups per load), table resident in the L2s after the first pass
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ table, unsigned lines, float* out, int iters)
{
    unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    float    s = 0.0f;
    for (int it = 0; it < iters; ++it)
    {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            h = h * 1664525u + 1013904223u;
            v[j] = table[size_t((h >> 8) % lines) * 32u + (threadIdx.x & 31u)];
        }
        s += v[0] + v[1] + v[2] + v[3];
    }
    if (s == 12345.678f) out[threadIdx.x] = s;
}

struct Clock { double ghz; };
static double median_clock(const std::vector<unsigned long long>& c, int wgs)
{
    std::vector<double> g;
    for (int i = 0; i < wgs; ++i)
        if (c[2 * i + 1] > 0) g.push_back(double(c[2 * i]) / (double(c[2 * i + 1]) / 100e6) * 1e-9); // s_memrealtime: 100 MHz
    if (g.empty()) return 0.0;
    std::sort(g.begin(), g.end());
    return g[g.size() / 2];
}

int main(int argc, char** argv)
{
    const int    W      = argc > 1 ? std::atoi(argv[1]) : 4;
    const size_t copyMB = argc > 2 ? size_t(std::atoll(argv[2])) : 4096;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::printf("%s: %d CUs; V = v_fma_f32 loop at %d waves per SIMD, C = copy of %zu MB (read + write %zu MB), G = 64-line gather loads\n", prop.name, cus, W, copyMB, 2 * copyMB);
    hipStream_t sv, sc;
    CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    // V: one launch fills the device exactly once (W workgroups of 256 threads per CU = W waves per SIMD), so that its duration is iters x 16 FMAs at the clock it gets
    const int      vWgs = cus * W;
    const unsigned pad  = W >= 8 ? 0u : unsigned(160u * 1024u / unsigned(W)) - 1024u;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(valu_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(pad)));
    float* out;
    unsigned long long* clocks;
    CHECK(hipMalloc(&out, 4096));
    CHECK(hipMalloc(&clocks, sizeof(unsigned long long) * 2 * vWgs));
    const size_t n = copyMB * (1u << 20) / 16u / (256u * kDepth) * (256u * kDepth);
    f4 *src, *dst;
    CHECK(hipMalloc(&src, n * 16u));
    CHECK(hipMalloc(&dst, n * 16u));
    CHECK(hipMemset(src, 1, n * 16u));
    CHECK(hipMemset(dst, 0, n * 16u));
    const unsigned lines = 8u * 1024u * 1024u / 128u; // 8 MB table: L2 / Infinity-Cache resident
    float* table;
    CHECK(hipMalloc(&table, size_t(lines) * 128u));
    CHECK(hipMemset(table, 0, size_t(lines) * 128u));
    hipEvent_t e[6];
    for (auto& x : e) CHECK(hipEventCreate(&x));

    auto run_v = [&](int iters) { hipLaunchKernelGGL(valu_kernel, dim3(vWgs), dim3(256), pad, sv, out, clocks, 1.0f, iters); };
    auto run_c = [&](int reps) { for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(copy_kernel, dim3(cus * 4), dim3(256), 0, sc, src, dst, n); };
    auto run_g = [&](int iters) { hipLaunchKernelGGL(gather_kernel, dim3(cus * 4), dim3(256), 0, sc, table, lines, out, iters); };
    auto ms = [&](hipEvent_t a, hipEvent_t b) { float t; CHECK(hipEventElapsedTime(&t, a, b)); return double(t); };
    std::vector<unsigned long long> hc(2 * vWgs);
    auto clock_now = [&]() { CHECK(hipMemcpy(hc.data(), clocks, sizeof(unsigned long long) * 2 * vWgs, hipMemcpyDeviceToHost)); return median_clock(hc, vWgs); };

    // calibrate: V ~ 40 ms alone, C ~ 40 ms alone (reps), G ~ 40 ms alone; warm up the clocks with 0.3 s of V first
    for (int i = 0; i < 8; ++i) run_v(400000);
    CHECK(hipDeviceSynchronize());
    int vIters = 200000;
    {
        CHECK(hipEventRecord(e[0], sv)); run_v(vIters); CHECK(hipEventRecord(e[1], sv)); CHECK(hipDeviceSynchronize());
        vIters = int(double(vIters) * 40.0 / ms(e[0], e[1]));
    }
    int cReps = 4;
    {
        run_c(2); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e[0], sc)); run_c(cReps); CHECK(hipEventRecord(e[1], sc)); CHECK(hipDeviceSynchronize());
        cReps = std::max(1, int(double(cReps) * 40.0 / ms(e[0], e[1])));
    }
    int gIters = 2000;
    {
        run_g(200); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e[0], sc)); run_g(gIters); CHECK(hipEventRecord(e[1], sc)); CHECK(hipDeviceSynchronize());
        gIters = std::max(1, int(double(gIters) * 40.0 / ms(e[0], e[1])));
    }
    const double copyBytes = 2.0 * double(n) * 16.0 * cReps;
    for (int round = 0; round < 3; ++round)
    {
        // alone
        CHECK(hipEventRecord(e[0], sv)); run_v(vIters); CHECK(hipEventRecord(e[1], sv)); CHECK(hipDeviceSynchronize());
        const double vAlone = ms(e[0], e[1]), vClk = clock_now();
        CHECK(hipEventRecord(e[0], sc)); run_c(cReps); CHECK(hipEventRecord(e[1], sc)); CHECK(hipDeviceSynchronize());
        const double cAlone = ms(e[0], e[1]);
        CHECK(hipEventRecord(e[0], sc)); run_g(gIters); CHECK(hipEventRecord(e[1], sc)); CHECK(hipDeviceSynchronize());
        const double gAlone = ms(e[0], e[1]);
        // V beside C
        CHECK(hipEventRecord(e[0], sv)); CHECK(hipEventRecord(e[2], sc));
        run_v(vIters); run_c(cReps);
        CHECK(hipEventRecord(e[1], sv)); CHECK(hipEventRecord(e[3], sc)); CHECK(hipDeviceSynchronize());
        const double vWithC = ms(e[0], e[1]), cWithV = ms(e[2], e[3]), vClkC = clock_now();
        // V beside G
        CHECK(hipEventRecord(e[0], sv)); CHECK(hipEventRecord(e[2], sc));
        run_v(vIters); run_g(gIters);
        CHECK(hipEventRecord(e[1], sv)); CHECK(hipEventRecord(e[3], sc)); CHECK(hipDeviceSynchronize());
        const double vWithG = ms(e[0], e[1]), gWithV = ms(e[2], e[3]), vClkG = clock_now();
        // G beside C: both use the vector memory pipeline (the gather its tag look-ups, the copy its bandwidth) and no arithmetic
        hipStream_t sg = sv;
        CHECK(hipEventRecord(e[0], sg)); CHECK(hipEventRecord(e[2], sc));
        hipLaunchKernelGGL(gather_kernel, dim3(cus * 4), dim3(256), 0, sg, table, lines, out, gIters);
        run_c(cReps);
        CHECK(hipEventRecord(e[1], sg)); CHECK(hipEventRecord(e[3], sc)); CHECK(hipDeviceSynchronize());
        const double gWithC = ms(e[0], e[1]), cWithG = ms(e[2], e[3]);
        std::printf("round %d\n", round);
        std::printf("  alone:      V %7.2f ms at %.3f GHz (%.2f cycles per wave-FMA per SIMD)   C %7.2f ms = %.2f TB/s   G %7.2f ms\n", vAlone, vClk,
                    vAlone * 1e-3 * vClk * 1e9 / (double(vIters) * kChains * W), cAlone, copyBytes / (cAlone * 1e-3) / 1e12, gAlone);
        std::printf("  V beside C: V %7.2f ms at %.3f GHz (x%.2f)   C %7.2f ms = %.2f TB/s (x%.2f)   both done after %.2f ms; alone back to back %.2f ms -> %.0f %% of the shorter hidden\n", vWithC, vClkC,
                    vWithC / vAlone, cWithV, copyBytes / (cWithV * 1e-3) / 1e12, cWithV / cAlone, std::max(vWithC, cWithV), vAlone + cAlone,
                    100.0 * (vAlone + cAlone - std::max(vWithC, cWithV)) / std::min(vAlone, cAlone));
        std::printf("  V beside G: V %7.2f ms at %.3f GHz (x%.2f)   G %7.2f ms (x%.2f)   both done after %.2f ms; alone back to back %.2f ms -> %.0f %% of the shorter hidden\n", vWithG, vClkG, vWithG / vAlone,
                    gWithV, gWithV / gAlone, std::max(vWithG, gWithV), vAlone + gAlone, 100.0 * (vAlone + gAlone - std::max(vWithG, gWithV)) / std::min(vAlone, gAlone));
        std::printf("  G beside C: G %7.2f ms (x%.2f)   C %7.2f ms = %.2f TB/s (x%.2f)   both done after %.2f ms; alone back to back %.2f ms -> %.0f %% of the shorter hidden\n", gWithC, gWithC / gAlone, cWithG,
                    copyBytes / (cWithG * 1e-3) / 1e12, cWithG / cAlone, std::max(gWithC, cWithG), gAlone + cAlone, 100.0 * (gAlone + cAlone - std::max(gWithC, cWithG)) / std::min(gAlone, cAlone));
    }
    return 0;
}
