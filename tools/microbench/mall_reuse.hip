// mall_reuse.hip -- developer microbenchmark (not part of the product; round 6).  How much of a plane a pass has just written does the next pass still find in the
// 256 MB Infinity Cache -- by the direction the consumer walks in, and by the cache policy of the producer's OTHER traffic?
//   P   the producer: out[i] = in0[i] + ... + in4[i] over five input planes and one output plane of `MB` megabytes each (16-byte texels, a 4K float4 plane = 133 MB): the
//       composite's shape -- 0.8 GB through a 256 MB cache.  Its input loads are plain or non-temporal (`nt`), its stores plain or non-temporal.
//   C   the consumer: reads the producer's output plane (and writes a quarter-size plane), rows top-down or bottom-up.
// Reported: the consumer's time after the producer, per combination; and alone after a cache flush (another 1 GB stream) as the no-reuse reference.
// (skew_bytes: plane k starts k * skew bytes into its allocation.  Measured with 4 KB, 68 KB and 1.06 MB: the producer takes 155.5 us either way -- streams of several planes read at
//  the same offsets do not meet in the same channels; the planes of the library need no staggering.)
//   hipcc --offload-arch=gfx950 -O3 -o mall_reuse mall_reuse.hip && ./mall_reuse [plane_megabytes = 133] [skew_bytes = 0]
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

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT_LOAD, bool NT_STORE> __global__ __launch_bounds__(256) void producer(const f4* a, const f4* b, const f4* c, const f4* d, const f4* e, f4* out, size_t n)
{
    const size_t i = size_t(blockIdx.x) * 256u + threadIdx.x;
    if (i >= n) return;
    f4 v;
    if (NT_LOAD) v = __builtin_nontemporal_load(a + i) + __builtin_nontemporal_load(b + i) + __builtin_nontemporal_load(c + i) + __builtin_nontemporal_load(d + i) + __builtin_nontemporal_load(e + i);
    else v = a[i] + b[i] + c[i] + d[i] + e[i];
    if (NT_STORE) __builtin_nontemporal_store(v, out + i);
    else out[i] = v;
}
template <bool UP> __global__ __launch_bounds__(256) void consumer(const f4* in, float* out, size_t n)
{
    const size_t blk = UP ? size_t(gridDim.x) - 1u - blockIdx.x : size_t(blockIdx.x);
    const size_t i   = blk * 256u + threadIdx.x;
    if (i >= n) return;
    const f4 v = in[i];
    out[i] = v.x + v.y + v.z + v.w;
}
__global__ __launch_bounds__(256) void flush(const f4* a, f4* b, size_t n)
{
    const size_t i = size_t(blockIdx.x) * 256u + threadIdx.x;
    if (i < n) b[i] = a[i];
}

int main(int argc, char** argv)
{
    const size_t mb = argc > 1 ? size_t(std::atoll(argv[1])) : 133;
    const size_t n  = mb * 1000u * 1000u / 16u / 256u * 256u;
    f4 *p[5], *out, *fa, *fb;
    float* small;
    // [skew_bytes]: plane k starts k * skew bytes behind its allocation (do the streams of several planes, read at the same offsets, meet in the same channels?)
    const size_t skew = argc > 2 ? size_t(std::atoll(argv[2])) : 0;
    int          kth  = 0;
    for (auto& q : p) { CHECK(hipMalloc(&q, n * 16u + 6u * skew + 256u)); CHECK(hipMemset(q, 0, n * 16u + 6u * skew)); q += size_t(kth++) * skew / 16u; }
    CHECK(hipMalloc(&out, n * 16u + 6u * skew + 256u));
    out += 5u * skew / 16u;
    if (skew) std::printf("plane k offset by k x %zu bytes\n", skew);
    CHECK(hipMalloc(&small, n * 4u));
    const size_t nf = size_t(512) * 1000u * 1000u / 16u;
    CHECK(hipMalloc(&fa, nf * 16u)); CHECK(hipMalloc(&fb, nf * 16u));
    CHECK(hipMemset(fa, 0, nf * 16u)); CHECK(hipMemset(fb, 0, nf * 16u));
    hipEvent_t e0, e1, e2;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
    const dim3 grid(unsigned((n + 255u) / 256u)), block(256);
    std::printf("planes of %zu MB (five read + one written by the producer = %.2f GB); consumer reads the written plane\n", mb, 6.0 * double(n) * 16e-9);
    auto run = [&](int prodKind, bool up, const char* what) {
        std::vector<double> tp, tc;
        for (int rep = 0; rep < 12; ++rep)
        {
            hipLaunchKernelGGL(flush, dim3(unsigned((nf + 255u) / 256u)), block, 0, 0, fa, fb, nf); // 1 GB of unrelated traffic: the cache holds nothing of the planes
            CHECK(hipEventRecord(e0, 0));
            if (prodKind == 0) hipLaunchKernelGGL((producer<false, false>), grid, block, 0, 0, p[0], p[1], p[2], p[3], p[4], out, n);
            if (prodKind == 1) hipLaunchKernelGGL((producer<true, false>), grid, block, 0, 0, p[0], p[1], p[2], p[3], p[4], out, n);
            if (prodKind == 2) hipLaunchKernelGGL((producer<true, true>), grid, block, 0, 0, p[0], p[1], p[2], p[3], p[4], out, n);
            if (prodKind == 3) hipLaunchKernelGGL((producer<false, true>), grid, block, 0, 0, p[0], p[1], p[2], p[3], p[4], out, n);
            CHECK(hipEventRecord(e1, 0));
            if (up) hipLaunchKernelGGL(consumer<true>, grid, block, 0, 0, out, small, n);
            else hipLaunchKernelGGL(consumer<false>, grid, block, 0, 0, out, small, n);
            CHECK(hipEventRecord(e2, 0));
            CHECK(hipDeviceSynchronize());
            float a, b;
            CHECK(hipEventElapsedTime(&a, e0, e1)); CHECK(hipEventElapsedTime(&b, e1, e2));
            if (rep >= 2) { tp.push_back(a * 1e3); tc.push_back(b * 1e3); }
        }
        std::sort(tp.begin(), tp.end()); std::sort(tc.begin(), tc.end());
        std::printf("  %-58s producer %7.1f us   consumer %6.1f us  (%.2f TB/s over its %.0f MB)\n", what, tp[tp.size() / 2], tc[tc.size() / 2],
                    (double(n) * 20.0) / (tc[tc.size() / 2] * 1e-6) / 1e12, double(n) * 20e-6);
    };
    // the no-reuse reference: the consumer behind the flush alone
    {
        std::vector<double> tc;
        for (int rep = 0; rep < 12; ++rep)
        {
            hipLaunchKernelGGL(flush, dim3(unsigned((nf + 255u) / 256u)), block, 0, 0, fa, fb, nf);
            CHECK(hipEventRecord(e1, 0));
            hipLaunchKernelGGL(consumer<false>, grid, block, 0, 0, out, small, n);
            CHECK(hipEventRecord(e2, 0));
            CHECK(hipDeviceSynchronize());
            float b;
            CHECK(hipEventElapsedTime(&b, e1, e2));
            if (rep >= 2) tc.push_back(b * 1e3);
        }
        std::sort(tc.begin(), tc.end());
        std::printf("  %-58s %28s consumer %6.1f us\n", "consumer alone behind 1 GB of unrelated traffic", "", tc[tc.size() / 2]);
    }
    run(0, false, "plain producer, consumer top-down");
    run(0, true, "plain producer, consumer bottom-up");
    run(1, false, "producer with non-temporal LOADS, consumer top-down");
    run(1, true, "producer with non-temporal LOADS, consumer bottom-up");
    run(2, false, "producer with non-temporal loads AND stores, top-down");
    run(2, true, "producer with non-temporal loads AND stores, bottom-up");
    run(3, false, "producer with non-temporal STORES only, top-down");
    run(3, true, "producer with non-temporal STORES only, bottom-up");
    return 0;
}
