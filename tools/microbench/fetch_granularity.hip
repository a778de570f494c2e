// fetch_granularity.hip -- developer microbenchmark (not part of the product): in what units does this device's L2 fetch from memory, and how do the memory-side
// counters tally them?  MI355X_MICROARCH.md calibrates FETCH_SIZE on wide coalesced streams only (128-byte requests tallied at 64 bytes: double it); the chain's two
// gather kernels (SSR's ray march, SSAO's taps) read scattered 4-byte texels.  Every lane of kernel S reads ONE float at byte offset lane * S of a 4 GiB buffer
// (first touch: nothing is cached), S = 32, 64, 128, 256:
//   * if the L2 fills whole 128-byte lines, S = 64 issues half the requests per lane of S = 128 (two lanes share a line) and S = 32 a quarter;
//   * if it fills 64-byte sectors, S = 64 and S = 128 issue the same number per lane.
// Run under the counters (separate pass from any trace):
//   rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace -d /tmp/fg -- ./fetch_granularity ; python tools/pmc_stats.py /tmp/fg TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
// (tools/pmc_stats.py lists kernels whose name contains "mifx::", hence the namespace.)   hipcc --offload-arch=gfx950 -O3 -o fetch_granularity fetch_granularity.hip
#include <hip/hip_runtime.h>
#include <cstdio>

namespace mifx
{
template <int STRIDE> __global__ void fetch_stride_kernel(const unsigned char* src, float* sink, unsigned lanes)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lanes) return;
    const float v = *reinterpret_cast<const float*>(src + size_t(i) * STRIDE);
    if (v == 123456.0f) sink[i & 1023u] = v; // (never true: the buffer is zero; keeps the load)
}
} // namespace mifx

int main()
{
    const size_t bytes = size_t(4) << 30;
    unsigned char* buf = nullptr;
    float* sink = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4096) != hipSuccess) return 1;
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    const unsigned lanes = 1u << 24; // 16 Mi lanes: 512 MiB .. 4 GiB of address range
    const dim3 block(256), grid(lanes / 256);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](auto kernel, int stride) {
        hipEventRecord(a);
        hipLaunchKernelGGL(kernel, grid, block, 0, 0, buf, sink, lanes);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0.0f;
        hipEventElapsedTime(&ms, a, b);
        std::printf("stride %3d B: %u lanes, %.3f ms  (%.0f MiB of address range; %.2f TB/s if every 128-byte line touched was fetched whole)\n", stride, lanes, ms,
                    double(lanes) * stride / 1048576.0, double(lanes) * (stride < 128 ? stride : 128) / (ms * 1e-3) / 1e12);
    };
    run(mifx::fetch_stride_kernel<256>, 256);
    run(mifx::fetch_stride_kernel<128>, 128);
    run(mifx::fetch_stride_kernel<64>, 64);
    run(mifx::fetch_stride_kernel<32>, 32);
    return 0;
}
