// mifx_ufloat.h -- the unsigned small floats of R11G11B10_FLOAT (5 exponent bits, M = 6 or 5 mantissa bits, no sign): encode, decode, and the two in one step.
// Plain integer / float code, no dependency: included by mifx_device.h (inside namespace mifx, MIFX_UF = MIFX_D) and by tools/check_ufloat.cpp, which compares
// quantize_ufloat with decode(encode(x)) for every one of the 2^32 float bit patterns on the host.
#ifndef MIFX_UF
#define MIFX_UF MIFX_D
#endif
// unsigned small float with 5 exponent bits and M mantissa bits (float11: M = 6, float10: M = 5), integer arithmetic only
template <int M> MIFX_UF unsigned float_to_ufloat(float x)
{
    const unsigned f = __builtin_bit_cast(unsigned, x);
    const unsigned e = (f >> 23) & 0xffu, m = f & 0x7fffffu;
    if (e == 255u) return m ? ((31u << M) | (1u << (M - 1))) : ((f >> 31) ? 0u : (31u << M)); // NaN stays NaN; -INF -> 0, +INF stays
    if (f >> 31) return 0u;                                                                   // negative values clamp to 0
    const int E = int(e) - 127 + 15;
    if (E >= 31) return 31u << M; // overflow -> +INF
    unsigned mant, shift;
    if (E <= 0)
    {
        if (E < -M) return 0u;             // below half of the smallest subnormal (ties at E == -M round to even = 0 or up below)
        mant  = m | 0x800000u;             // implicit one
        shift = unsigned(23 - M + 1 - E);  // 18 .. 24 + M
    }
    else
    {
        mant  = (unsigned(E) << 23) | m;   // exponent and mantissa as one integer: a mantissa carry increments the exponent
        shift = unsigned(23 - M);
    }
    const unsigned q = mant >> shift, rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1);
    return q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u);
}
template <int M> MIFX_UF float ufloat_to_float(unsigned v)
{
    const unsigned e = v >> M, m = v & ((1u << M) - 1u);
    if (e == 31u) return __builtin_bit_cast(float, 0x7f800000u | (m << (23 - M)));
    if (e == 0u) return float(m) * (1.0f / float(1u << (14 + M))); // subnormal: m * 2^-14 / 2^M
    return __builtin_bit_cast(float, ((e + 112u) << 23) | (m << (23 - M)));
}

// decode(encode(x)) without building the code: what a store followed by a load does to a value.
//   negative (and -INF) -> 0, NaN stays NaN; at or above 2^-14 round the mantissa to M bits (nearest even; a carry moves into the exponent) and map everything from
//   2^16 up to +INF; below 2^-14 the format is a fixed-point grid of 2^-(14 + M): (x + C) - C with C = 1.5 * 2^(9 - M) rounds to it (nearest even) in fp32.
template <int M> MIFX_UF float quantize_ufloat(float x)
{
    x = x < 0.0f ? 0.0f : x;
    const unsigned u = __builtin_bit_cast(unsigned, x);
    constexpr unsigned shift = 23u - unsigned(M), mask = (1u << shift) - 1u;
    const unsigned r = (u + (mask >> 1) + ((u >> shift) & 1u)) & ~mask;
    float n = __builtin_bit_cast(float, r);
    n = n >= 65536.0f ? __builtin_bit_cast(float, 0x7f800000u) : n;
    constexpr float C = 1.5f * float(1u << (9 - M));
    const float s = (x + C) - C;
    return x != x ? x : (x < 6.103515625e-5f ? s : n); // (2^-14; a NaN whose payload sits in the low mantissa bits would otherwise be rounded to INF)
}
// the code of a value that quantize_ufloat<M> produced (a member of the format's value set, or NaN): float_to_ufloat<M>(x) == encode_quantized<M>(quantize_ufloat<M>(x))
// for every float (tools/check_ufloat.cpp) in a third of the instructions.
template <int M> MIFX_UF unsigned encode_quantized(float q)
{
    const unsigned u      = __builtin_bit_cast(unsigned, q);
    const unsigned normal = (u >> (23 - M)) - (112u << M);                 // exponent re-biased (127 -> 15), mantissa truncated (exact: q is on the grid)
    const unsigned sub    = unsigned(q * float(1u << (14 + M)));           // below 2^-14: a multiple of 2^-(14 + M)
    unsigned code = q < 6.103515625e-5f ? sub : normal;
    code = code > (31u << M) ? (31u << M) : code;                           // +INF (re-biased exponent 143)
    return q != q ? ((31u << M) | (1u << (M - 1))) : code;                  // NaN stays NaN
}
