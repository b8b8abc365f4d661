// "%.18e" text of doubles formed ON the device: the rows of the reference's
// dataset files (numpy.savetxt's format; utils/dataloader.py:55,59,108 read
// them back) leave the GPU as text, so that regenerating the files at scale
// moves only the bytes that go to disk and keeps the host out of the digits.
//
// Same arithmetic as the host formatter (auv_format.hpp): a finite double is
// m * 2^e with a 53-bit integer m; its 19 significant decimal digits are
// N = round_half_even(m * 2^e * 10^k), 10^18 <= N < 10^19.  Here 5^k is a
// 128-bit constant (k <= 55), m * 5^k an exact 192-bit integer and the power of
// two an exact shift, which covers 1e-36 <= |v| < 1e18 and zero: every value a
// dataset row can hold.  Anything else (huge, tiny, infinite, NaN) is COUNTED,
// not approximated: the caller then formats that block on the host.  In the
// covered range the exponent has two digits, so a number takes 24 characters
// plus its sign, and the place of every number in its row is known from the
// signs before it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace auv {

constexpr int kFmtPow5 = 56;  // 5^55 < 2^128
struct FmtPow5 {
    uint64_t lo[kFmtPow5], hi[kFmtPow5];
};
constexpr FmtPow5 make_fmt_pow5() {
    FmtPow5 t{};
    unsigned __int128 p = 1;
    for (int i = 0; i < kFmtPow5; ++i) {
        t.lo[i] = (uint64_t)p;
        t.hi[i] = (uint64_t)(p >> 64);
        p *= 5;
    }
    return t;
}
__constant__ FmtPow5 kFmtPow5Table = make_fmt_pow5();

constexpr uint64_t kFmtTen18 = 1000000000000000000ull;
constexpr uint64_t kFmtTen19 = 10000000000000000000ull;

// what the device formats itself: zero, and 1e-36 <= |v| < 1e18
__device__ __forceinline__ bool fmt_in_range(double v) {
    const double av = __builtin_fabs(v);
    return av == 0.0 || (av >= 1e-36 && av < 1e18);  // false for NaN and infinities
}

// 19 digits N and decimal exponent of av (> 0, in range); false if it does not come out (never, in range)
__device__ __forceinline__ bool fmt_digits19(double av, uint64_t& n_out, int& e10_out) {
    const uint64_t bits = (uint64_t)__double_as_longlong(av);
    const int E = (int)(bits >> 52) & 0x7ff;
    if (E == 0 || E == 0x7ff) return false;  // subnormal, infinite, NaN
    const uint64_t m = (bits & 0xfffffffffffffull) | (1ull << 52);
    const int e2 = E - 1075;   // av = m * 2^e2
    const int be = e2 + 53;    // av = fr * 2^be, fr in [0.5, 1)
    int e10 = (int)__builtin_floor((double)(be - 1) * 0.30102999566398120);  // floor(log10 av), maybe one too small
    for (int attempt = 0; attempt < 3; ++attempt) {
        const int k = 18 - e10;
        if (k < 0 || k >= kFmtPow5) return false;
        // prod = m * 5^k, 192 bits
        const uint64_t pl = kFmtPow5Table.lo[k], ph = kFmtPow5Table.hi[k];
        const uint64_t w0 = m * pl, c0 = __umul64hi(m, pl);
        const uint64_t p1 = m * ph, c1 = __umul64hi(m, ph);
        const uint64_t w1 = c0 + p1;
        const uint64_t w2 = c1 + (w1 < c0 ? 1ull : 0ull);
        const int sh = e2 + k;  // N = prod * 2^sh
        uint64_t n0, n1;
        if (sh >= 0) {
            if (sh > 11 || w1 != 0 || w2 != 0 || (sh > 0 && (w0 >> (64 - sh)) != 0)) {
                // does not fit 64 bits: more than 19 digits, the exponent guess was too small
                n0 = 0;
                n1 = 1;
            } else {
                n0 = w0 << sh;
                n1 = 0;
            }
        } else {
            // X = prod >> (s - 1), remembering whether a dropped bit was set; then the rounding bit is X's lowest
            const int t = -sh - 1;
            if (t >= 192) return false;
            const int ws = t >> 6, bs = t & 63;
            uint64_t sticky = (ws >= 1 ? w0 : 0ull) | (ws >= 2 ? w1 : 0ull);
            const uint64_t a0 = ws == 0 ? w0 : ws == 1 ? w1 : w2;
            const uint64_t a1 = ws == 0 ? w1 : ws == 1 ? w2 : 0ull;
            const uint64_t a2 = ws == 0 ? w2 : 0ull;
            uint64_t x0 = a0, x1 = a1, x2 = a2;
            if (bs) {
                sticky |= a0 & ((1ull << bs) - 1ull);
                x0 = (a0 >> bs) | (a1 << (64 - bs));
                x1 = (a1 >> bs) | (a2 << (64 - bs));
                x2 = a2 >> bs;
            }
            const uint64_t round = x0 & 1ull;
            n0 = (x0 >> 1) | (x1 << 63);
            n1 = (x1 >> 1) | (x2 << 63);
            if ((x2 >> 1) != 0) return false;
            if (round && (sticky != 0 || (n0 & 1ull))) {  // above half, or half and odd
                n0 += 1ull;
                n1 += n0 == 0 ? 1ull : 0ull;
            }
        }
        if (n1 == 0 && n0 < kFmtTen18) {  // exponent guess one too high (cannot happen with this guess, kept for safety)
            e10 -= 1;
            continue;
        }
        if (n1 != 0 || n0 >= kFmtTen19) {  // the guess was one too small, or rounding carried to 10^19
            e10 += 1;
            continue;
        }
        n_out = n0;
        e10_out = e10;
        return true;
    }
    return false;
}

// the 24 characters of |v| at p (d.dddddddddddddddddde+XX); false if v is outside the device's range
__device__ __forceinline__ bool fmt_write_abs(double av, char* p) {
    uint64_t n = 0;
    int e10 = 0;
    if (av != 0.0 && !fmt_digits19(av, n, e10)) return false;
    if (e10 > 99 || e10 < -99) return false;
    // 19 digits as 1 + 9 + 9
    const uint64_t q = n / 1000000000ull;
    unsigned int r_lo = (unsigned int)(n - q * 1000000000ull);
    const unsigned int top = (unsigned int)(q / 1000000000ull);
    unsigned int r_mid = (unsigned int)(q - (uint64_t)top * 1000000000ull);
    p[0] = (char)('0' + top);
    p[1] = '.';
#pragma unroll
    for (int i = 8; i >= 0; --i) {
        p[2 + i] = (char)('0' + r_mid % 10u);
        r_mid /= 10u;
        p[11 + i] = (char)('0' + r_lo % 10u);
        r_lo /= 10u;
    }
    p[20] = 'e';
    p[21] = e10 < 0 ? '-' : '+';
    const unsigned int ae = (unsigned int)(e10 < 0 ? -e10 : e10);
    p[22] = (char)('0' + ae / 10u);
    p[23] = (char)('0' + ae % 10u);
    return true;
}

struct FmtArgs {
    const double* values;    // [n_rows, stride]
    const int32_t* row_len;  // [n_rows] numbers in use per row, or nullptr: all of them
    long long n_rows, stride;
    char* text;
    long long text_cap;
    long long* row_off;  // [n_rows + 1]: bytes per row, then (after the scan) where each row starts; total last
    int32_t* status;     // [0] values outside the device's range, [1] rows that did not fit text_cap
};

// numbers in use in a row, held to what the row can hold
__device__ __forceinline__ long long fmt_row_len(const FmtArgs& a, long long row) {
    if (!a.row_len) return a.stride;
    const long long n = a.row_len[row];
    return n < 0 ? 0 : (n > a.stride ? a.stride : n);
}

// bytes of every row: a wave per row counts the signs (a number is 24 characters, its sign if negative, and the
// blank or newline behind it) and the values the device cannot format
__global__ __launch_bounds__(256) void k_fmt_row_bytes(FmtArgs a) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= a.n_rows) return;
    const long long len = fmt_row_len(a, row);
    const double* v = a.values + row * a.stride;
    int negs = 0, bad = 0;
    for (long long j = lane; j < len; j += 64) {
        const double x = v[j];
        negs += __builtin_signbit(x) ? 1 : 0;
        bad += fmt_in_range(x) ? 0 : 1;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        negs += __shfl_xor(negs, d, 64);
        bad += __shfl_xor(bad, d, 64);
    }
    if (lane == 0) {
        a.row_off[row] = len > 0 ? 25 * len + negs : 1;  // an empty row is its newline
        if (bad) atomicAdd(&a.status[0], bad);
    }
}

// exclusive running sum of row_off[0 .. n) in place, the total at row_off[n]; one workgroup
__global__ __launch_bounds__(1024) void k_fmt_scan(long long* row_off, long long n) {
    __shared__ long long part[1024];
    const int tid = threadIdx.x;
    const long long per = (n + 1023) / 1024;
    const long long lo = tid * per, hi = lo + per < n ? lo + per : n;
    long long s = 0;
    for (long long i = lo; i < hi; ++i) s += row_off[i];
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {  // inclusive scan of the partial sums
        const long long add = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    long long run = tid ? part[tid - 1] : 0;
    for (long long i = lo; i < hi; ++i) {
        const long long b = row_off[i];
        row_off[i] = run;
        run += b;
    }
    if (tid == 1023) row_off[n] = part[1023];
}

// the text of one row per workgroup: number j of the row starts 25 j + (negative numbers before it) bytes in
__global__ __launch_bounds__(256) void k_fmt_write(FmtArgs a) {
    __shared__ int wave_negs[4];
    const long long row = blockIdx.x;
    const long long len = fmt_row_len(a, row);
    const long long start = a.row_off[row], end = a.row_off[row + 1];
    if (end > a.text_cap) {
        if (threadIdx.x == 0) atomicAdd(&a.status[1], 1);
        return;
    }
    char* out = a.text + start;
    if (len <= 0) {
        if (threadIdx.x == 0) out[0] = '\n';
        return;
    }
    const double* v = a.values + row * a.stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int negs_before = 0;  // in the chunks already written
    for (long long j0 = 0; j0 < len; j0 += 256) {
        const long long j = j0 + threadIdx.x;
        const bool on = j < len;
        const double x = on ? v[j] : 0.0;
        const bool neg = on && __builtin_signbit(x);
        const unsigned long long mask = __ballot(neg);
        const int below = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wave_negs[wave] = __popcll(mask);
        __syncthreads();
        int before = negs_before + below;
        for (int w = 0; w < wave; ++w) before += wave_negs[w];
        const int chunk = wave_negs[0] + wave_negs[1] + wave_negs[2] + wave_negs[3];
        if (on) {
            char* p = out + 25 * j + before;
            if (neg) *p++ = '-';
            if (!fmt_write_abs(__builtin_fabs(x), p)) {
                for (int i = 0; i < 24; ++i) p[i] = '?';  // counted by k_fmt_row_bytes: the caller drops this text
            }
            p[24] = j + 1 == len ? '\n' : ' ';
        }
        negs_before += chunk;
        __syncthreads();
    }
}

}  // namespace auv
