// Text of a float64 as str(numpy.float64) / repr(float) print it, and pandas' reading of that text back -- for the DEVICE
// (round 5: file -> rows; rows.hip has the host form on std::to_chars, which the CPU tests compare this one with).
//   helper_file.py:1403-1478 (save_list: str() of every value), :860-905 (get_data: pandas.read_csv), :1366-1400 (save_df_to_csv)
// Shortest digits that round-trip, by Steele & White / Burger & Dybvig's free-format algorithm in 128-bit fixed point: enough for
// 2^-20 <= |v| < 2^24 (a track's coordinates, sizes and angles; everything else -- and NaN, infinities -- is left to the host:
// `in_range`).  No tables, no division: a digit is at most nine subtractions.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__) || defined(__CUDACC__)
#define YSMR_FMT_HD __host__ __device__ __forceinline__
#else
#define YSMR_FMT_HD inline
#endif

namespace ysmr_fmt {

typedef unsigned __int128 u128;

struct Digits {
    char d[20];     // '0' .. '9', the first not '0'
    int n;          // how many (1 .. 17)
    int exp10;      // the value is d[0].d[1]d[2]... x 10^exp10
};

YSMR_FMT_HD uint64_t bits_of(double v) { union { double d; uint64_t u; } c; c.d = v; return c.u; }
YSMR_FMT_HD double double_of(uint64_t u) { union { double d; uint64_t u; } c; c.u = u; return c.d; }

// finite, non-zero, 2^-20 <= |v| < 2^24
YSMR_FMT_HD bool in_range(double v)
{
    const int e = (int)((bits_of(v) >> 52) & 0x7FFu);
    return e >= 1023 - 20 && e < 1023 + 24;
}

// v > 0, in_range(v)
YSMR_FMT_HD void shortest(double v, Digits &out)
{
    const uint64_t b = bits_of(v);
    const uint64_t frac = b & ((1ull << 52) - 1ull);
    const int e2 = (int)((b >> 52) & 0x7FFu) - 1075;        // v = m 2^e2, m = 2^52 + frac
    const uint64_t m = (1ull << 52) | frac;
    constexpr int F = 84;                                    // fixed point: 1.0 = 2^F
    const int sh = e2 + F;                                   // 12 .. 55
    u128 R = (u128)m << sh;
    u128 Mp = (u128)1 << (sh - 1);                           // half the gap to the next double up
    u128 Mm = frac == 0 ? (u128)1 << (sh - 2) : Mp;          // ... down (half as wide below a power of two)
    const bool even = (m & 1ull) == 0;                       // round-to-nearest-even: the boundaries themselves read back as v
    u128 S = (u128)1 << F;
    int k = 0;
    // scale: (R + Mp) / S in [1/10, 1)
    while (even ? R + Mp >= S : R + Mp > S) { S *= 10; ++k; }
    while (!(even ? (R + Mp) * 10 >= S : (R + Mp) * 10 > S)) { R *= 10; Mp *= 10; Mm *= 10; --k; }
    int n = 0;
    while (true) {
        R *= 10; Mp *= 10; Mm *= 10;
        int d = 0;
        while (R >= S) { R -= S; ++d; }
        const bool low = even ? R <= Mm : R < Mm;
        const bool high = even ? R + Mp >= S : R + Mp > S;
        if (!low && !high) { out.d[n++] = (char)('0' + d); continue; }
        if (low && high) {                                   // both neighbours read back as v: the nearer one, a tie to the even digit
            const u128 twice = R * 2;                        // (ties are real: a float32's double has an exact expansion ending in 5)
            d += twice < S ? 0 : (twice > S ? 1 : (d & 1));
        }
        else if (high) d += 1;
        out.d[n++] = (char)('0' + d);
        break;
    }
    out.n = n;
    out.exp10 = k - 1;
}

// CPython's layout of those digits (format_float_short, 'r'): exponent form iff the decimal exponent is < -4 or >= 16,
// otherwise positional with at least ".0".  Returns the end.
YSMR_FMT_HD char *layout(char *p, const Digits &g)
{
    const int nd = g.n, exp10 = g.exp10;
    if (exp10 < -4 || exp10 >= 16) {
        *p++ = g.d[0];
        if (nd > 1) { *p++ = '.'; for (int i = 1; i < nd; ++i) *p++ = g.d[i]; }
        *p++ = 'e';
        *p++ = exp10 < 0 ? '-' : '+';
        int a = exp10 < 0 ? -exp10 : exp10;
        if (a >= 100) { *p++ = (char)('0' + a / 100); a %= 100; }
        *p++ = (char)('0' + a / 10); *p++ = (char)('0' + a % 10);
        return p;
    }
    const int decpt = exp10 + 1;   // digits before the decimal point
    if (decpt <= 0) {
        *p++ = '0'; *p++ = '.';
        for (int i = 0; i < -decpt; ++i) *p++ = '0';
        for (int i = 0; i < nd; ++i) *p++ = g.d[i];
    } else if (decpt >= nd) {
        for (int i = 0; i < nd; ++i) *p++ = g.d[i];
        for (int i = nd; i < decpt; ++i) *p++ = '0';
        *p++ = '.'; *p++ = '0';
    } else {
        for (int i = 0; i < decpt; ++i) *p++ = g.d[i];
        *p++ = '.';
        for (int i = decpt; i < nd; ++i) *p++ = g.d[i];
    }
    return p;
}

// 10^e, 0 <= e <= 44, as pandas' table holds it: the literal 1e<e>, i.e. the correctly rounded double.  10^0 .. 10^22 are
// exact products; the others are spelled out.
YSMR_FMT_HD double pow10_table(int e)
{
    switch (e) {
        case 0: return 1e0; case 1: return 1e1; case 2: return 1e2; case 3: return 1e3; case 4: return 1e4; case 5: return 1e5;
        case 6: return 1e6; case 7: return 1e7; case 8: return 1e8; case 9: return 1e9; case 10: return 1e10; case 11: return 1e11;
        case 12: return 1e12; case 13: return 1e13; case 14: return 1e14; case 15: return 1e15; case 16: return 1e16;
        case 17: return 1e17; case 18: return 1e18; case 19: return 1e19; case 20: return 1e20; case 21: return 1e21;
        case 22: return 1e22; case 23: return 1e23; case 24: return 1e24; case 25: return 1e25; case 26: return 1e26;
        case 27: return 1e27; case 28: return 1e28; case 29: return 1e29; case 30: return 1e30; case 31: return 1e31;
        case 32: return 1e32; case 33: return 1e33; case 34: return 1e34; case 35: return 1e35; case 36: return 1e36;
        case 37: return 1e37; case 38: return 1e38; case 39: return 1e39; case 40: return 1e40; case 41: return 1e41;
        case 42: return 1e42; case 43: return 1e43; default: return 1e44;
    }
}

// pandas' float converter on the text [p, end) of a POSITIVE number laid out by layout() (rows.hip: pandas_parse, the
// restatement of precise_xstrtod in pandas/_libs/src/parser/tokenizer.c): at most 17 digits (leading zeros included) are
// accumulated in a double, the rest only shifts the exponent, and the result is scaled by one multiplication or division with a
// table power of ten.  |exponent| stays below 45 for the numbers shortest() serves.
YSMR_FMT_HD double pandas_parse_positive(const char *p, const char *end)
{
    double number = 0.0;
    int exponent = 0, nd = 0, ndec = 0;
    const int max_digits = 17;
    while (p < end && *p >= '0' && *p <= '9') {
        if (nd < max_digits) { number = number * 10.0 + (double)(*p - '0'); ++nd; }
        else ++exponent;
        ++p;
    }
    if (p < end && *p == '.') {
        ++p;
        while (nd < max_digits && p < end && *p >= '0' && *p <= '9') { number = number * 10.0 + (double)(*p - '0'); ++p; ++nd; ++ndec; }
        if (nd >= max_digits)
            while (p < end && *p >= '0' && *p <= '9') ++p;
        exponent -= ndec;
    }
    if (p < end && (*p == 'e' || *p == 'E')) {
        ++p;
        bool eneg = false;
        if (p < end && *p == '-') { eneg = true; ++p; }
        else if (p < end && *p == '+') ++p;
        int n = 0;
        while (p < end && *p >= '0' && *p <= '9') { n = n * 10 + (*p - '0'); ++p; }
        exponent += eneg ? -n : n;
    }
    if (exponent > 0) number *= pow10_table(exponent);
    else if (exponent < 0) number /= pow10_table(-exponent);
    return number;
}

// One value of a row: its text at p (the end is returned) -- with via_pandas the text of the value pandas reads back from v's
// text, and *v becomes that value.  ok = false: not served here (the caller hands the rows to the host); nothing is written.
YSMR_FMT_HD char *put_value(char *p, double *v, bool via_pandas, bool &ok)
{
    double a = *v;
    const bool neg = (bits_of(a) >> 63) != 0;
    if (neg) a = -a;
    if (a == 0.0) { if (neg) *p++ = '-'; *p++ = '0'; *p++ = '.'; *p++ = '0'; return p; }     // ("-0.0" reads back as -0.0: unchanged)
    if (!in_range(a)) { ok = false; return p; }
    Digits g;
    shortest(a, g);
    char *q = p;
    if (neg) *q++ = '-';
    char *end = layout(q, g);
    if (!via_pandas) return end;
    const double back = pandas_parse_positive(q, end);
    if (back == a) return end;
    if (!in_range(back)) { ok = false; return p; }
    *v = neg ? -back : back;
    shortest(back, g);
    return layout(q, g);
}

YSMR_FMT_HD char *put_u32(char *p, uint32_t v)
{
    char t[10];
    int n = 0;
    do { t[n++] = (char)('0' + v % 10u); v /= 10u; } while (v);
    while (n) *p++ = t[--n];
    return p;
}

}  // namespace ysmr_fmt
