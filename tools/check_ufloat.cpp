// Exhaustive host check of quantize_ufloat<M> against ufloat_to_float<M>(float_to_ufloat<M>(x)), and of encode_quantized<M>(quantize_ufloat<M>(x)) against float_to_ufloat<M>(x), (diligentfx_amd/csrc/mifx_ufloat.h) for all 2^32 bit patterns:
//   g++ -O2 -fopenmp -std=c++17 -I diligentfx_amd/csrc tools/check_ufloat.cpp -o /tmp/check_ufloat && /tmp/check_ufloat
#include <cstdint>
#include <cstdio>
#include <cstring>
#define MIFX_UF static inline
#include "mifx_ufloat.h"

template <int M> static unsigned long long check()
{
    unsigned long long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (long long i = 0; i < (1ll << 32); ++i)
    {
        const uint32_t u = uint32_t(i);
        float x;
        std::memcpy(&x, &u, 4);
        const float a = ufloat_to_float<M>(float_to_ufloat<M>(x)), b = quantize_ufloat<M>(x);
        uint32_t ua, ub;
        std::memcpy(&ua, &a, 4);
        std::memcpy(&ub, &b, 4);
        const bool nanA = a != a, nanB = b != b;
        const bool code_ok = float_to_ufloat<M>(x) == encode_quantized<M>(b);
        if (nanA != nanB || (!nanA && ua != ub) || !code_ok)
        {
            if (bad < 5) std::printf("M=%d x=%08x want %08x got %08x\n", M, u, ua, ub);
            ++bad;
        }
    }
    return bad;
}
int main()
{
    const unsigned long long b6 = check<6>(), b5 = check<5>();
    std::printf("quantize_ufloat<6>: %llu mismatches, quantize_ufloat<5>: %llu mismatches of 2^32 inputs each\n", b6, b5);
    return (b6 || b5) ? 1 : 0;
}
