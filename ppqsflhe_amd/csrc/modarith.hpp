// modarith.hpp -- 64-bit modular arithmetic for gfx950 device code (and host mirrors).
//
// Stands in for OpenFHE's NativeIntegerT::{ModAddFast, ModSubFast, ModMulFastConst,
// ModMul} ([upstream] core/include/math/hal/intnat/ubintnat.h), reached from every
// DCRTPoly operation on the hot path.  Results are canonical
// residues in [0,q) wherever a value is stored as an output; inside transforms the
// Harvey lazy ranges [0,2q)/[0,4q) are used (q < 2^62).
//
// gfx950 has no 64-bit integer multiplier: a 64x64->128 product is four
// v_mad_u64_u32; a low-64 product is one v_mad_u64_u32 + two v_mul_lo_u32.  The
// helpers below are written so hipcc emits exactly those.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MK_HD __host__ __device__ __forceinline__
#define MK_D __device__ __forceinline__
#else
#define MK_HD inline
#define MK_D inline
#endif

// Non-temporal policy for streamed operands (measured on MI355X, same-box A/B at C3: 0 -> 18.40 k ct/s,
// 1 -> 18.60 k, 2 -> 18.90 k): the once-read / once-written tiles stop evicting the twiddle and eval-key tiles that
// other workgroups of the same XCD re-read from L2.
#ifndef MK_NT
#define MK_NT 2
#endif

namespace mk {

typedef uint64_t u64;
typedef unsigned __int128 u128;

// per-limb constants kept in a device table (one entry per modulus of QP)
struct LimbConst {
    u64 q;        // modulus
    u64 q2;       // 2q
    u64 mu;       // floor(2^(62+k) / q), k = bit length of q   (Barrett)
    u64 c64;      // 2^64 mod q                                  (128-bit fold)
    u64 ninv;     // N^-1 mod q
    u64 ninv_sh;  // Shoup companion of ninv
    uint32_t k;   // bit length of q
    uint32_t sh;  // k - 2: window shift for Barrett
    // fp64 path (limbs with q < 1.25 * 2^50 when the engine enables it): transforms of such a limb run on
    // v_fma_f64 (53-bit exact products) instead of the 32-bit integer multiplier
    double qd;        // (double) q
    double qinv;      // 1 / q
    double ninv_d;    // (double) N^-1 mod q
    double ninv_qd;   // ninv_d / q
    uint32_t fp;      // 1: this limb's NTT tables hold doubles (w, w/q) and its inter-pass data are doubles
    // pseudo-Mersenne path (integer limbs with q = 2^k - c, c <= 2^(k-34) -- every 60-bit prime OpenFHE's generator
    // picks is one: it walks down from 2^60 in steps of 2N): the NTT tables' companions hold w * 2^32 mod q and the
    // butterflies reduce by folding at bit k (pm_lazy / pm_fold below) instead of Shoup's quotient estimate
    uint32_t pm;
    uint32_t pm_c;    // c = 2^k - q
    uint32_t pad_;
};

// whether q qualifies for the pseudo-Mersenne butterflies (host side; the engine additionally wants an integer limb)
MK_HD bool pm_eligible(u64 q) {
    uint32_t k = 0;
    while (k < 64 && (q >> k)) ++k;
    if (k < 40 || k > 60) return false;
    const u64 c = ((u64)1 << k) - q;
    return c <= ((u64)1 << (k - 34));
}

MK_HD u64 mulhi64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (u64)(((u128)a * b) >> 64);
#endif
}

MK_HD void mul128(u64 a, u64 b, u64 &hi, u64 &lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    lo = a * b;
    hi = __umul64hi(a, b);
#else
    u128 p = (u128)a * b;
    lo = (u64)p;
    hi = (u64)(p >> 64);
#endif
}

// x - m when x >= m, else x, for x, m < 2^63: the difference is formed with one 64-bit add of the
// (wave-uniform) negated constant and its sign selects -- no carry chain, no 64-bit compare.
MK_HD u64 csub(u64 x, u64 m) {
    const u64 t = x + (0 - m);
    return (int64_t)t < 0 ? x : t;
}

MK_HD u64 add_mod(u64 a, u64 b, u64 q) {
    u64 s = a + b;
    return s >= q ? s - q : s;
}
MK_HD u64 sub_mod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }

// Shoup multiplication by a constant w with companion wp = floor(w * 2^64 / q).
// Valid for ANY 64-bit a; lazy result in [0, 2q).
MK_HD u64 shoup_lazy(u64 a, u64 w, u64 wp, u64 q) {
    u64 h = mulhi64(a, wp);
    return a * w - h * q;
}
// canonical result in [0, q)
MK_HD u64 shoup_mul(u64 a, u64 w, u64 wp, u64 q) { return csub(shoup_lazy(a, w, wp, q), q); }

// ---- pseudo-Mersenne limbs: q = 2^k - c, c <= 2^(k-34), U := 2^k --------------------------------------------------
// gfx950 issues v_mad_u64_u32 at the rate of a 32-bit add (tools/ubench_intmul.hip), so what counts is the NUMBER of
// instructions.  Shoup's product is 3 multiplies for the quotient's high word + 2 x 3 for the two low products plus
// their glue; with 2^k = c (mod q) a product is reduced by folding its bits above k back in with one small multiply:
// 5 v_mad_u64_u32 + the 32-bit word shuffling between them.
struct PmK {           // wave-uniform shifts / masks / constants of one limb
    uint32_t c, c2;    // c, 2c
    uint32_t t;        // 63 - k: the NTT tables of such a limb hold w << t and (w * 2^32 mod q) << t
    uint32_t s_f, m_f;    // split of a 64-bit word at bit k: shift k-32, mask 2^(k-32) - 1
    u64 q3;            // 3q >= any pm_lazy result (offset of the butterflies' subtractions)
    uint32_t c64;      // 2^64 mod q = c 2^(64-k) <= 2^30 (pm_reduce128)
    uint32_t e60;      // 2^60 mod q = c 2^(60-k) <= 2^26 (pm_reduce_cols)
};
MK_HD PmK pm_consts(const LimbConst &L) {
    PmK p;
    p.c = L.pm_c;
    p.c2 = 2 * L.pm_c;
    p.t = 63 - L.k;
    p.s_f = L.k - 32;
    p.m_f = (1u << p.s_f) - 1u;
    p.q3 = 3 * L.q;
    p.c64 = L.pm_c << (64 - L.k);
    p.e60 = L.pm_c << (60 - L.k);
    return p;
}
// table entries of a pseudo-Mersenne limb for the twiddle w (host side)
MK_HD u64 pm_tw(u64 w, const LimbConst &L) { return w << (63 - L.k); }
inline u64 pm_tw_companion(u64 w, const LimbConst &L) { return (u64)(((u128)w << 32) % L.q) << (63 - L.k); }
// x (any 64-bit word) -> x mod q in [0, U + 2^30):  (x mod 2^k) + (x >> k) c, and (x >> k) c < 2^(64-k) 2^(k-34).
MK_HD u64 pm_fold(u64 x, const PmK &P) {
    const uint32_t xh = (uint32_t)(x >> 32);
    const u64 lo = ((u64)(xh & P.m_f) << 32) | (uint32_t)x;
    return (u64)(xh >> P.s_f) * P.c + lo;
}
// {high word of y, 0} as a register pair in ONE instruction (the compiler would build it from two moves)
MK_HD u64 hi32_pair(u64 y) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 r;
    asm("v_lshrrev_b64 %0, 32, %1" : "=v"(r) : "v"(y));
    return r;
#else
    return y >> 32;
#endif
}
// a * w mod q, lazy, for a < 8U = 2^(k+3), from the table entries wt = w << t and wxt = (w * 2^32 mod q) << t (t = 63 - k):
//   S' = a_lo wt + a_hi wxt = 2^t S,  S = a_lo w + a_hi (w 2^32 mod q) = a w (mod q),  S < 2^32 U + 2^(k-29) U <= 1.5 * 2^32 U
//   the fold position k+1 of S is bit 64 of S' (96 bits: z : low word of y1):  hi = S >> (k+1) = high word of z < 1.5 * 2^31,
//   lo = S mod 2^(k+1) = (S' mod 2^64) >> t,  and 2^(k+1) = 2c (mod q)  ->  lo + hi 2c < 2U + 0.375U.
// Result in [0, 2.375U).  No partial sum wraps: a_lo wt_lo < 2^64; a_lo wt_hi + 2^32 < 2^64 (wt < 2^63); a_hi < 2^31 keeps
// a_hi wxt_lo + 2^32 below 2^64; z ends as floor(S' / 2^32) < 1.5 * 2^63.
MK_HD u64 pm_lazy(u64 a, u64 wt, u64 wxt, const PmK &P) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32);
    const u64 y0 = (u64)a0 * (uint32_t)wt;
    u64 z = (u64)a0 * (uint32_t)(wt >> 32) + hi32_pair(y0);
    const u64 y1 = (u64)a1 * (uint32_t)wxt + (u64)(uint32_t)y0;
    z = (u64)a1 * (uint32_t)(wxt >> 32) + z;
    z += hi32_pair(y1);
    const u64 lo = (((u64)(uint32_t)z << 32) | (uint32_t)y1) >> P.t;
    return (u64)(uint32_t)(z >> 32) * P.c2 + lo;
}

// X = hi:lo -> X mod q in [0, q), for X < 2^(2k+6) (a sum of up to 6 products of a lazy word < 8U and a residue): the high
// word is folded with 2^64 = c64 (mod q) into Y = m1 2^32 + (low word of m0) < 2^(2k-28) + 2^64, Y is folded at bit k
// (Y >> k < 2^32) into r < U + 2^32 c <= 1.25U, one conditional subtraction finishes.  11 instructions where Barrett's
// quotient estimate (high product, low product, two corrections) takes about 25.
MK_HD u64 pm_reduce128(u64 hi, u64 lo, const PmK &P, u64 q) {
    const u64 m0 = (u64)(uint32_t)hi * P.c64 + (u64)(uint32_t)lo;
    u64 m1 = (u64)(uint32_t)(hi >> 32) * P.c64 + hi32_pair(lo);
    m1 += hi32_pair(m0);
    const uint32_t yh = (uint32_t)(m1 >> P.s_f);
    const u64 ylo = ((u64)((uint32_t)m1 & P.m_f) << 32) | (uint32_t)m0;
    return csub((u64)yh * P.c + ylo, q);
}

// Barrett reduction of a 128-bit x = hi:lo with x < 2^(k+62) (k = bitlen q) to [0,q).
// qhat = mulhi64(x >> (k-2), mu), mu = floor(2^(62+k)/q): qhat in {Q-2,Q-1,Q}.
MK_HD u64 barrett_reduce128(u64 hi, u64 lo, const LimbConst &L) {
    u64 y = (hi << (64 - L.sh)) | (lo >> L.sh);
    u64 qh = mulhi64(y, L.mu);
    u64 r = lo - qh * L.q;
    return csub(csub(r, L.q2), L.q);
}

// general a*b mod q for a,b < q (product < q^2 < 2^(2k) <= 2^(k+62))
MK_HD u64 mul_mod(u64 a, u64 b, const LimbConst &L) {
    u64 hi, lo;
    mul128(a, b, hi, lo);
    return barrett_reduce128(hi, lo, L);
}

// A sum of at most 4 products a_i*b_i with a_i < 2^60 and b_i < q is below 2^(k+62), so it fits the
// Barrett window without the fold of reduce_wide (base conversions with <= 4 source limbs, key-switch
// inner products with <= 4 digits).
MK_HD u64 reduce_sum4(u64 hi, u64 lo, const LimbConst &L) { return barrett_reduce128(hi, lo, L); }

// reduce an arbitrary 128-bit accumulator (< 2^124) mod q: fold the high word with
// 2^64 mod q first so that the Barrett window fits (x' < 2^(k+60) + 2^64).
MK_HD u64 reduce_wide(u64 hi, u64 lo, const LimbConst &L) {
    u64 fh, fl;
    mul128(hi, L.c64, fh, fl);
    u64 nlo = fl + lo;
    u64 nhi = fh + (nlo < fl ? 1 : 0);
    return barrett_reduce128(nhi, nlo, L);
}

// reduce one 64-bit word mod q (q may be much smaller than 2^64)
MK_HD u64 reduce_word(u64 x, const LimbConst &L) { return barrett_reduce128(0, x, L); }

// ---- exact modular arithmetic on fp64 FMAs (q < 1.25 * 2^50) --------------------------------------
// Residues are integers held in doubles.  y*w = h + l exactly (h = RN(y*w), l = fma(y,w,-h)); b = rint(y * (w/q))
// is the quotient within 1.5 for |y| <= 2^52; h - b*q is an integer below 2^53, so fma(-b,q,h) is exact and
// v = (h - b*q) + l = y*w - b*q exactly, |v| <= q * (0.5 + 1.0001 |y| / 2^52).   (tools/ubench_fpmod.hip: 0
// mismatches in 6.5e9 trials against integer arithmetic, incl. q just below 2^51.)
#if defined(__HIPCC__)
MK_D double fp_mulmod(double y, double w, double wq, double q) {
    const double h = __dmul_rn(y, w);
    const double l = __fma_rn(y, w, -h);
    const double b = rint(__dmul_rn(y, wq));
    const double c = __fma_rn(-b, q, h);
    return __dadd_rn(c, l);
}
// x - q*rint(x/q): |result| <= 0.51 q for |x| < 2^53 (exact: the fma's true value is an integer below 2^53)
MK_D double fp_reduce(double x, double q, double qinv) { return __fma_rn(-rint(__dmul_rn(x, qinv)), q, x); }
MK_D u64 fp_to_canonical(double x, double q, double qinv) {
    double r = fp_reduce(x, q, qinv);
    r = r < 0.0 ? r + q : r;
    return (u64)r;
}
// y*e mod q for an arbitrary (not precomputed) factor e in [0,q), |y| <= q: the quotient is estimated from the
// rounded product (relative error <= 1.5*2^-52, i.e. <= 0.47 units at |y| = q < 1.25*2^50), so the result is the
// exact integer y*e - b*q with |result| <= 0.97 q (<= 0.75 q when |y| <= 0.51 q); the fma's true value is an
// integer below 2^52 and therefore exact
MK_D double fp_mulmod_any(double y, double e, double q, double qinv) {
    const double h = __dmul_rn(y, e);
    const double l = __fma_rn(y, e, -h);
    const double b = rint(__dmul_rn(h, qinv));
    const double c = __fma_rn(-b, q, h);
    return __dadd_rn(c, l);
}
// exact conversion of an integer below 2^52 (splice it into the mantissa of 2^52, subtract 2^52)
MK_D double u52_to_double(u64 v) {
    return __longlong_as_double((long long)(v | 0x4330000000000000ull)) - 4503599627370496.0;
}
MK_D u64 dbits(double d) { return (u64)__double_as_longlong(d); }
MK_D double bitsd(u64 b) { return __longlong_as_double((long long)b); }
#endif

// ---- column accumulation for sums of products of 60-bit numbers ----------------------------------
// a = a1*2^30 + a0, b = b1*2^30 + b0 (all four halves < 2^30):  sum_i a_i*b_i = C0 + C1*2^30 + C2*2^60 with
//   C0 = sum a0*b0, C1 = sum (a0*b1 + a1*b0), C2 = sum a1*b1.
// Every partial product is < 2^60, so for <= 4 terms the three columns fit 64 bits and each is ONE chain of
// v_mad_u64_u32 (d = a*b + d) -- no carries, no 128-bit adds.  reduce_cols() reduces the triple mod q.
struct Cols {
    u64 c0, c1, c2;
};
MK_HD void split30(u64 v, uint32_t &lo, uint32_t &hi) {
    lo = (uint32_t)v & 0x3FFFFFFFu;
    hi = (uint32_t)(v >> 30);
}
MK_HD void mac_cols(Cols &c, uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1) {
    c.c0 = (u64)a0 * b0 + c.c0;
    c.c1 = (u64)a0 * b1 + c.c1;
    c.c1 = (u64)a1 * b0 + c.c1;
    c.c2 = (u64)a1 * b1 + c.c2;
}
// S = c0 + c1*2^30 + c2*2^60 < 2^(k+62)  ->  S mod q.  The Barrett window floor(S / 2^sh) is assembled from
// the three columns (under-estimate by <= 2), so qhat in {Q-3..Q} and the remainder is below 4q.
MK_HD u64 reduce_cols_lazy(const Cols &c, const LimbConst &L) {  // result in [0, 4q)
    const u64 lo = c.c0 + (c.c1 << 30) + (c.c2 << 60);
    const int e1 = 30 - (int)L.sh;  // sh in [18,58]
    const u64 y = (c.c0 >> L.sh) + (e1 >= 0 ? (c.c1 << e1) : (c.c1 >> (-e1))) + (c.c2 << (60 - L.sh));
    const u64 qh = mulhi64(y, L.mu);
    return lo - qh * L.q;
}
// the same triple on a pseudo-Mersenne limb (columns of <= 4 products a_i b_i, a_i < 2^60, b_i < q), lazy result below
// 2.1U (the first butterfly stage takes < 8U):
//   S = A + t 2^60,  A = c0 + (c1 mod 2^30) 2^30 < 2^62 + 2^60,  t = c2 + (c1 >> 30) < 2^(k+2) + 2^33,  2^60 = e60 (mod q)
//   T = t e60 < 2^(k+2) c 2^(60-k) + ... < 2^(k+29) is folded at bit k: (T mod 2^k) + (T >> k) c < U + 2^29 c < 1.01U;
//   A is folded below 1.001U.
MK_HD u64 pm_reduce_cols(const Cols &c, const PmK &P) {
    const u64 t = c.c2 + (c.c1 >> 30);
    const u64 m0 = (u64)(uint32_t)t * P.e60;
    const u64 m1 = (u64)(uint32_t)(t >> 32) * P.e60 + hi32_pair(m0);  // T = m1 2^32 + low word of m0
    const uint32_t yh = (uint32_t)(m1 >> P.s_f);
    const u64 ylo = ((u64)((uint32_t)m1 & P.m_f) << 32) | (uint32_t)m0;
    const u64 a = c.c0 + ((c.c1 & 0x3FFFFFFFull) << 30);
    // (a wave-uniform "skip the fold when k = 60" costs 36 more registers in k_conv_col than the three instructions save)
    return (u64)yh * P.c + ylo + pm_fold(a, P);
}
MK_HD u64 reduce_cols(const Cols &c, const LimbConst &L) {
    return csub(csub(reduce_cols_lazy(c, L), L.q2), L.q);
}
// 5..8 terms: the middle column is kept as two sums (each <= 8 * 2^60 fits), and the total (< 2^123) goes through
// the folding reduction.
struct Cols4 {
    u64 c0, c1a, c1b, c2;
};
MK_HD void mac_cols4(Cols4 &c, uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1) {
    c.c0 = (u64)a0 * b0 + c.c0;
    c.c1a = (u64)a0 * b1 + c.c1a;
    c.c1b = (u64)a1 * b0 + c.c1b;
    c.c2 = (u64)a1 * b1 + c.c2;
}
MK_HD void add128(u64 &hi, u64 &lo, u64 xhi, u64 xlo) {
    lo += xlo;
    hi += xhi + (lo < xlo ? 1 : 0);
}
MK_HD u64 reduce_cols4(const Cols4 &c, const LimbConst &L) {  // canonical
    u64 hi = 0, lo = c.c0;
    add128(hi, lo, c.c1a >> 34, c.c1a << 30);
    add128(hi, lo, c.c1b >> 34, c.c1b << 30);
    add128(hi, lo, c.c2 >> 4, c.c2 << 60);
    return reduce_wide(hi, lo, L);
}

// a 60-bit word stored as its two 30-bit halves in the two 32-bit halves of a u64 (what split30 would produce)
MK_HD u64 pack30(u64 v) { return (v & 0x3FFFFFFFull) | ((v >> 30) << 32); }
MK_HD u64 unpack30(u64 p) { return (p & 0xFFFFFFFFull) | ((p >> 32) << 30); }

// 128-bit accumulate acc += a*b
MK_HD void mac128(u64 &hi, u64 &lo, u64 a, u64 b) {
    u64 ph, pl;
    mul128(a, b, ph, pl);
    lo += pl;
    hi += ph + (lo < pl ? 1 : 0);
}

}  // namespace mk
