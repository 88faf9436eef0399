// ptnn_text.hpp -- the text side of the result-file layout (host only): np.savetxt's bytes without printf.
//
// The reference dumps every chain's traces with np.savetxt -- pos_w at '%.18e', the scores at '%1.8f' / '%1.2f', the likelihood
// at '%1.4f' (multicore-pt-regression/pt_timeseries_regression.py:454-481, 864-868) -- and show_results reads them back with
// np.loadtxt (REG:795-831).  At the device's sampling rate that text IS the run: 64 chains x 10 000 samples x 31 weights are
// 558 MB of digits, 0.22 s through snprintf on 16 threads against 0.04 s of sampling.  Two observations make it cheap:
//
//   * every value is a float32 (the device's traces) or a short double, and np.savetxt's output for it is the CORRECTLY ROUNDED
//     decimal expansion (glibc rounds the exact binary value, ties to even).  |v| = M 2^E exactly, so round(|v| 10^p) is one
//     128-bit multiplication, one shift and a look at the remainder: exact, no tables of powers of five beyond 10^38, ~20 ns.
//     Whatever does not fit (more than 127 bits, exotic flags or widths) goes through snprintf as before -- same bytes either way;
//   * a rejected MH step repeats the previous row (pos_w[i+1] = pos_w[i], REG:417): 80 - 98 % of a chain's rows.  A row equal
//     to the one before it reuses that row's text.
//
// text_round (the value np.loadtxt reads back from '%1.Nf') needs no string at all: it is round(|v| 10^N) / 10^N, one correctly
// rounded division of two exactly representable doubles -- what strtod returns for those digits.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

namespace ptnn_text {

typedef unsigned __int128 u128;

struct Format {
    bool ok = false;        // one floating conversion printf understands (validated by the caller's float_format_ok)
    bool fast = false;      // %[1][.N](f|e) without flags: handled here; otherwise every value goes through snprintf
    char conv = 0;          // 'f' or 'e'
    int prec = 6;
    char text[16] = {0};
};

inline Format parse_format(const char* fmt) {
    Format f;
    const size_t n = std::strlen(fmt);
    if (n < 2 || n >= sizeof f.text || fmt[0] != '%') return f;
    std::memcpy(f.text, fmt, n + 1);
    f.ok = true;
    f.conv = fmt[n - 1];
    size_t k = 1;
    long width = 0;
    bool digits_only = true;
    while (k + 1 < n && fmt[k] >= '0' && fmt[k] <= '9') width = width * 10 + (fmt[k++] - '0');
    if (fmt[1] == '0' && k > 1) digits_only = false;            // a leading 0 is the zero-pad flag
    int prec = 6;
    if (k + 1 < n && fmt[k] == '.') {
        ++k;
        prec = 0;
        while (k + 1 < n && fmt[k] >= '0' && fmt[k] <= '9') prec = prec * 10 + (fmt[k++] - '0');
    }
    f.prec = prec;
    // anything left before the conversion character is a flag this file does not handle; width 0 / 1 never pads
    f.fast = digits_only && k + 1 == n && width <= 1 && (f.conv == 'f' || f.conv == 'e') && prec <= 18;
    return f;
}

inline const u128* pow10_table() {
    static u128 t[39];
    static bool init = false;
    if (!init) { t[0] = 1; for (int i = 1; i < 39; ++i) t[i] = t[i - 1] * 10; init = true; }
    return t;
}
inline int bitlen(u128 v) {
    const uint64_t hi = (uint64_t)(v >> 64), lo = (uint64_t)v;
    return hi ? 128 - __builtin_clzll(hi) : (lo ? 64 - __builtin_clzll(lo) : 0);
}

// |v| = M 2^E with M odd (or 0): the shortest integer mantissa
inline void decompose(double av, uint64_t& M, int& E) {
    uint64_t bits;
    std::memcpy(&bits, &av, 8);
    const int be = (int)((bits >> 52) & 0x7ff);
    M = bits & ((1ull << 52) - 1);
    if (be) { M |= 1ull << 52; E = be - 1075; } else E = -1074;
    if (M) { const int z = __builtin_ctzll(M); M >>= z; E += z; }
}

// q = round_half_even(M 2^E 10^p), p >= 0; false when the intermediate does not fit 127 bits or q does not fit 64
inline bool scaled_round(uint64_t M, int E, int p, uint64_t& q) {
    if (M == 0) { q = 0; return true; }
    if (p > 38) return false;
    const u128* P10 = pow10_table();
    const int mb = 64 - __builtin_clzll(M), pb = bitlen(P10[p]);
    if (mb + pb > 127) return false;
    u128 N = (u128)M * P10[p];
    if (E >= 0) {
        if (bitlen(N) + E > 64) return false;
        q = (uint64_t)(N << E);
        return true;
    }
    const int s = -E;
    if (s >= 128) { q = 0; return true; }                  // N < 2^127 <= 2^(s-1): below one half
    const u128 quo = N >> s, rem = N & ((((u128)1) << s) - 1), half = ((u128)1) << (s - 1);
    if (quo >> 64) return false;
    q = (uint64_t)quo;
    if (rem > half || (rem == half && (q & 1))) { if (++q == 0) return false; }
    return true;
}

inline char* put_u64(char* out, uint64_t v) {               // decimal, no padding
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *out++ = tmp[--n];
    return out;
}
inline char* put_u64_padded(char* out, uint64_t v, int digits) {   // exactly `digits` characters, zero padded on the left
    for (int k = digits - 1; k >= 0; --k) { out[k] = (char)('0' + v % 10); v /= 10; }
    return out + digits;
}

// printf("%.{prec}f", v) into out (no terminator); returns the end, or nullptr when the fast path does not apply
inline char* fixed(char* out, double v, int prec) {
    if (!(v == v) || std::isinf(v)) return nullptr;
    uint64_t M; int E;
    decompose(std::fabs(v), M, E);
    uint64_t q;
    if (!scaled_round(M, E, prec, q)) return nullptr;
    if (std::signbit(v)) *out++ = '-';
    const uint64_t p10 = (uint64_t)pow10_table()[prec];
    out = put_u64(out, q / p10);
    if (prec) { *out++ = '.'; out = put_u64_padded(out, q % p10, prec); }
    return out;
}

// printf("%.{prec}e", v) (prec <= 18); nullptr when the fast path does not apply
inline char* scientific(char* out, double v, int prec) {
    if (!(v == v) || std::isinf(v)) return nullptr;
    const double av = std::fabs(v);
    uint64_t M; int E;
    decompose(av, M, E);
    int k = 0;
    uint64_t q = 0;
    const uint64_t lo = (uint64_t)pow10_table()[prec], hi = lo * 10;      // q must land in [10^prec, 10^(prec+1))
    if (M) {
        // floor(log10 av) from the binary exponent, then at most one correction either way
        const int e2 = (64 - __builtin_clzll(M)) + E - 1;                 // av in [2^e2, 2^(e2+1))
        k = (int)std::floor(e2 * 0.30102999566398120);
        for (int tries = 0;; ++tries) {
            const int p = prec - k;
            if (p < 0 || tries > 3) return nullptr;
            if (!scaled_round(M, E, p, q)) return nullptr;
            if (q >= hi) {
                // either the estimate was one short, or rounding carried 9.99..95 up to 10.0: both read "one decade up"
                if (q == hi) { const int p1 = p - 1; uint64_t q1; if (p1 >= 0 && scaled_round(M, E, p1, q1) && q1 >= lo && q1 < hi) { q = q1; ++k; break; } }
                ++k;
                continue;
            }
            if (q < lo) { --k; continue; }
            break;
        }
    }
    if (std::signbit(v)) *out++ = '-';
    if (prec) {
        *out++ = (char)('0' + q / lo);
        *out++ = '.';
        out = put_u64_padded(out, q % lo, prec);
    } else *out++ = (char)('0' + q);
    *out++ = 'e';
    *out++ = k < 0 ? '-' : '+';
    const unsigned ak = (unsigned)(k < 0 ? -k : k);
    if (ak < 10) { *out++ = '0'; *out++ = (char)('0' + ak); } else out = put_u64(out, ak);
    return out;
}

// one value in np.savetxt's format; always succeeds (snprintf when the fast path declines); returns the end
inline char* put_value(char* out, double v, const Format& f) {
    if (f.fast) {
        char* e = f.conv == 'f' ? fixed(out, v, f.prec) : scientific(out, v, f.prec);
        if (e) return e;
    }
    const int w = std::snprintf(out, 400, f.text, v);
    return out + (w < 0 ? 0 : (w >= 400 ? 399 : w));
}

// the value np.loadtxt reads back after np.savetxt(fmt): strtod(printf(fmt, v)) without the string where that is exact
inline double round_trip(double v, const Format& f) {
    if (f.fast && v == v && !std::isinf(v)) {
        if (f.conv == 'f' && f.prec <= 18) {
            uint64_t M; int E, q_ok;
            uint64_t q;
            decompose(std::fabs(v), M, E);
            q_ok = scaled_round(M, E, f.prec, q) && q < (1ull << 53);
            if (q_ok) {
                // q and 10^prec are exact doubles (prec <= 18 < 22): IEEE division rounds the true quotient correctly, which is
                // what strtod does with the same digits
                const double r = (double)q / (double)(uint64_t)pow10_table()[f.prec];
                return std::signbit(v) ? -r : r;
            }
        }
        if (f.conv == 'e' && f.prec >= 16) return v;        // 17 significant digits identify a double
    }
    char buf[512];
    std::snprintf(buf, sizeof buf, f.text, v);
    return std::strtod(buf, nullptr);
}

}  // namespace ptnn_text
