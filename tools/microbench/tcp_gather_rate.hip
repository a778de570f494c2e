// tcp_gather_rate.hip -- developer microbenchmark (not part of the product): how many clocks does a CU's vector L1 (TCP) spend on one wave64 dword load as a function of
// the number of distinct 128-byte lines its 64 lanes touch?  The chain's gather kernels (SSR's ray march: ~55 lines per depth tap; SSAO's 19 taps: ~38) issue few bytes
// per instruction but many tag look-ups; profiles/r03_pmc_tcp1_v8.txt counts 99 M / 107 M TCP accesses per launch for them, 390 k / 416 k per CU.  If the TCP handles
// about one look-up per clock, that alone is 160 / 173 us of their 212 / 307 us -- and explains why the two kernels, run side by side, take the sum of their times.
// Every lane reads floats from a 16 KB table (L1-resident after the first pass); the lane -> line map makes a wave touch L = 1 .. 64 lines per instruction.
//   hipcc --offload-arch=gfx950 -O3 -o tcp_gather_rate tcp_gather_rate.hip ; ./tcp_gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

namespace mifx
{
constexpr int kLines = 128; // 16 KB table
constexpr int kIter  = 512;
// L lines per wave instruction: lane l reads line (l % L + rotation) and the float (l / L) of it
template <int L> __global__ __launch_bounds__(256) void tcp_gather_kernel(const float* table, float* sink, unsigned long long* clocks)
{
    const unsigned lane = threadIdx.x & 63u;
    const unsigned slot = (lane / L) & 31u;
    float acc = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < kIter; it += 8)
    {
#pragma unroll
        for (int u = 0; u < 8; ++u)
        {
            const unsigned line = (lane % L + unsigned(it + u) * 7u + blockIdx.x) & unsigned(kLines - 1);
            acc += table[line * 32u + slot];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 123456.0f) sink[threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) clocks[0] = t1 - t0;
}
} // namespace mifx

int main()
{
    float* table = nullptr; float* sink = nullptr; unsigned long long* clocks = nullptr;
    hipMalloc(&table, mifx::kLines * 128); hipMalloc(&sink, 4096); hipMalloc(&clocks, 8);
    hipMemset(table, 0, mifx::kLines * 128);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    std::printf("%d CUs; table %d lines of 128 B; every wave issues %d dword loads; 8 waves per SIMD resident\n", cus, mifx::kLines, mifx::kIter);
    std::printf("%8s %12s %22s %24s\n", "lines", "ms", "ns per wave-load per CU", "clocks at 2.4 GHz per load");
    auto run = [&](auto kernel, int lines) {
        const int waves_per_cu = 32, rounds = 8; // 8 waves per SIMD x 4 SIMDs, 8 rounds of workgroups
        const dim3 block(256), grid(cus * waves_per_cu / 4 * rounds);
        hipLaunchKernelGGL(kernel, grid, block, 0, 0, table, sink, clocks); // warm-up
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(kernel, grid, block, 0, 0, table, sink, clocks);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0.0f; hipEventElapsedTime(&ms, a, b);
        ms /= 4.0f;
        const double loads_per_cu = double(waves_per_cu) * rounds * mifx::kIter;
        const double ns = ms * 1e6 / loads_per_cu;
        std::printf("%8d %12.4f %22.3f %24.2f\n", lines, ms, ns, ns * 2.4);
    };
    run(mifx::tcp_gather_kernel<1>, 1);
    run(mifx::tcp_gather_kernel<2>, 2);
    run(mifx::tcp_gather_kernel<4>, 4);
    run(mifx::tcp_gather_kernel<8>, 8);
    run(mifx::tcp_gather_kernel<16>, 16);
    run(mifx::tcp_gather_kernel<32>, 32);
    run(mifx::tcp_gather_kernel<64>, 64);
    return 0;
}
