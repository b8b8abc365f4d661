// "%.18e" text of doubles, as numpy.savetxt writes the reference's dataset
// files (utils/dataloader.py:55,59,108 read them back), without going through
// printf for the common magnitudes.
//
// A finite double is m * 2^e with a 53-bit integer m.  Its 19 significant
// decimal digits are N = round_half_even(m * 2^e * 10^k), k chosen so that
// 10^18 <= N < 10^19.  For 0 <= k <= 27, 5^k fits 64 bits, so m * 5^k is an
// exact 128-bit integer and the remaining power of two is an exact shift:
// the rounding is done on the true value, as glibc's printf does.  That covers
// 1e-9 <= |v| < 1e19; everything else (zero aside) takes snprintf.  Host code.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

namespace auvfmt {

struct Pow5Table {
    uint64_t t[28];
    Pow5Table() {
        uint64_t p = 1;
        for (int i = 0; i < 28; ++i) {
            t[i] = p;
            p *= 5;
        }
    }
};

// 5^0 .. 5^27; a function-local static of class type is initialised exactly once even when the
// formatter's worker threads arrive together
inline const uint64_t* pow5_table() {
    static const Pow5Table table;
    return table.t;
}

// digits of N * 2^-s... see header.  Returns false when v is outside the fast range.
inline bool digits19(double av, uint64_t* n_out, int* e10_out) {
    int be;
    const double fr = std::frexp(av, &be);                  // av = fr * 2^be, fr in [0.5, 1)
    const uint64_t m = (uint64_t)std::ldexp(fr, 53);        // exact: 53-bit integer
    const int e2 = be - 53;                                 // av = m * 2^e2
    int e10 = (int)std::floor((be - 1) * 0.30102999566398120);  // floor(log10 av), maybe one too small
    const uint64_t lo = 1000000000000000000ull;             // 10^18
    const uint64_t* p5 = pow5_table();
    for (int attempt = 0; attempt < 3; ++attempt) {
        const int k = 18 - e10;
        if (k < 0 || k > 27) return false;
        const unsigned __int128 prod = (unsigned __int128)m * p5[k];   // < 2^116
        const int sh = e2 + k;                                        // N = prod * 2^sh
        unsigned __int128 n;
        if (sh >= 0) {
            if (sh > 11) return false;  // would not fit a 19-digit value anyway
            n = prod << sh;
        } else {
            const int s = -sh;
            if (s >= 127) return false;
            n = prod >> s;
            const unsigned __int128 rem = prod - (n << s);
            const unsigned __int128 half = (unsigned __int128)1 << (s - 1);
            if (rem > half || (rem == half && (n & 1))) n += 1;
        }
        if (n < lo) {  // exponent guess one too high (cannot happen with this guess, kept for safety)
            e10 -= 1;
            continue;
        }
        if (n >= (unsigned __int128)lo * 10) {
            // either the guess was one too small, or rounding carried to 10^19
            e10 += 1;
            continue;
        }
        *n_out = (uint64_t)n;
        *e10_out = e10;
        return true;
    }
    return false;
}

// writes the text of v at out (at most 26 bytes, no terminator); returns its length
inline int format_e18(double v, char* out) {
    if (!std::isfinite(v)) return snprintf(out, 30, "%.18e", v);
    char* p = out;
    if (std::signbit(v)) *p++ = '-';
    const double av = std::fabs(v);
    uint64_t n = 0;
    int e10 = 0;
    if (av == 0.0) {
        n = 0;
    } else if (!digits19(av, &n, &e10)) {
        return snprintf(out, 30, "%.18e", v);
    }
    char dig[19];
    for (int i = 18; i >= 0; --i) {
        dig[i] = (char)('0' + (n % 10));
        n /= 10;
    }
    *p++ = dig[0];
    *p++ = '.';
    memcpy(p, dig + 1, 18);
    p += 18;
    *p++ = 'e';
    *p++ = e10 < 0 ? '-' : '+';
    const int ae = e10 < 0 ? -e10 : e10;
    if (ae >= 100) *p++ = (char)('0' + ae / 100);
    *p++ = (char)('0' + (ae / 10) % 10);
    *p++ = (char)('0' + ae % 10);
    return (int)(p - out);
}

// rows [r0, r1) into `out` (sized by the caller); returns bytes written
inline int64_t format_block(const double* values, const int32_t* row_len, int64_t stride, int64_t r0, int64_t r1,
                            char* out) {
    int64_t pos = 0;
    for (int64_t i = r0; i < r1; ++i) {
        const double* row = values + i * stride;
        for (int32_t j = 0; j < row_len[i]; ++j) {
            if (j) out[pos++] = ' ';
            pos += format_e18(row[j], out + pos);
        }
        out[pos++] = '\n';
    }
    return pos;
}

}  // namespace auvfmt
