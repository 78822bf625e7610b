// ntt_radix.hpp -- register-resident radix-H NTT rounds for gfx950 (H = 4, 8, 16).
//
// A sub-transform of R = H*H points is done in two rounds of log2(H) butterfly stages each.
// In a round every thread holds H elements in VGPRs and all its twiddles, so the only
// cross-thread traffic is ONE LDS exchange between the rounds (the v0 kernel in
// ntt_kernels.hpp did log2(R) LDS round trips with a dependent twiddle load per stage).
//
//   forward (Cooley-Tukey, large strides first):
//     round A: thread j holds x[j + H k], k < H      stages 0..log H-1     base_eff = base
//     round B: thread j holds x[H j + k], k < H      stages log H..2log H-1 base_eff = base*H + j
//   inverse (Gentleman-Sande) runs round B then round A with the stages reversed.
//   Twiddle of in-register stage s, group g (g < 2^s): table[(base_eff << s) + g].
//
// Every kernel exists in three arithmetic instances (template parameter AR): fp64 (AR_FP: exact FMA products, see
// modarith.hpp) for limbs below 1.25 * 2^50, and for the 60-bit limbs either the pseudo-Mersenne butterflies (AR_PM:
// q = 2^k - c, reduction by folding at bit k; what OpenFHE's 60-bit primes allow) or Shoup/Harvey lazy butterflies
// (AR_INT, any modulus); the host launches the fp64 and the integer instance over the limbs of their class, and picks
// ONE integer arithmetic per context (NttTables::int_pm: every integer limb qualifies for AR_PM).  512-point rows (N = 2^17) use three rounds of
// radix 8 (k_ntt_row3).  Fused kernels: k_conv_col (approximate base conversion + forward column pass), k_icol_sum +
// k_conv_col_psum (ApproxModDown's conversion for a whole group of clients), k_switch_col (rescale), k_row_inner_fp /
// k_row3_inner_fp (forward row pass of the converted digits + eval-key inner product, fp64 limbs; single re-encryptions),
// k_row3_inner_int (the same for the integer limbs, continuing into the inverse row pass on the P limbs),
// k_row3_tail_once and, in qsum_kernels.hpp, k_qsum3_fp (the Q-limb half of the n-client step).  In every row kernel a
// wavefront owns complete rows, so LDS hand-offs are wave-level (wave_lds_sync) and the kernels contain no workgroup
// barrier after the twiddle staging.
#pragma once
#include "modarith.hpp"
#include "ntt_kernels.hpp"

namespace mk {

// streaming (read-once / write-once) accesses: non-temporal (MK_NT in modarith.hpp: 1 = fused row kernels, 2 = every
// transform pass; data re-read from L2 by other workgroups -- twiddles, eval-key tiles, conversion sources -- stay default)
#ifndef MK_NT8
#define MK_NT8 0  // 8-byte (strided) loads stay on the default path: as non-temporal they cost 1.4 % (same-box A/B)
#endif
MK_D u64 ld_stream(const u64 *p) { return (MK_NT && MK_NT8) ? __builtin_nontemporal_load(p) : *p; }
MK_D ulong2 ld_stream2(const ulong2 *p) {
    if (MK_NT) {
        ulong2 v;
        v.x = __builtin_nontemporal_load(&p->x);
        v.y = __builtin_nontemporal_load(&p->y);
        return v;
    }
    return *p;
}
MK_D u64 ld_pass(const u64 *p) { return (MK_NT >= 2 && MK_NT8) ? __builtin_nontemporal_load(p) : *p; }
#ifndef MK_NT8S
#define MK_NT8S 1  // 8-byte strided STORES do profit from the non-temporal policy (+2.5 % against plain stores)
#endif
MK_D void st_pass(u64 *p, u64 v) {
    if (MK_NT >= 2 && MK_NT8S) __builtin_nontemporal_store(v, p);
    else *p = v;
}
MK_D ulong2 ld_pass2(const ulong2 *p) {
    if (MK_NT >= 2) {
        ulong2 v;
        v.x = __builtin_nontemporal_load(&p->x);
        v.y = __builtin_nontemporal_load(&p->y);
        return v;
    }
    return *p;
}
MK_D void st_pass2(ulong2 *p, ulong2 v) {
    if (MK_NT >= 2) {
        __builtin_nontemporal_store(v.x, &p->x);
        __builtin_nontemporal_store(v.y, &p->y);
    } else {
        *p = v;
    }
}
MK_D void st_stream2(ulong2 *p, ulong2 v) {
    if (MK_NT) {
        __builtin_nontemporal_store(v.x, &p->x);
        __builtin_nontemporal_store(v.y, &p->y);
    } else {
        *p = v;
    }
}

// In-kernel phase stamps (diagnostic build -DMK_STAMP=1 only; the product build compiles every call away).  A wave stamps
// the shader clock (s_memtime) at phase boundaries and leaves the phase lengths in NttTables::stamps, region `REGION` of
// STAMP_REGION entries: wave w of workgroup b writes 8 words at ((b * 4 + w) % (STAMP_REGION / 8)) * 8.  tools/stamps.py
// reads them back through mkckks_debug_stamps.
#ifndef MK_STAMP
#define MK_STAMP 0
#endif
constexpr unsigned STAMP_REGION = 1u << 20, STAMP_REGIONS = 4;
struct Stamper {
    unsigned long long t[8];
    MK_D Stamper() {
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = 0;
    }
    template <int K>
    MK_D void mark() {
        if (MK_STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            t[K] = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // t[K] += now - since (for phases inside loops); returns now
    template <int K>
    MK_D unsigned long long add(unsigned long long since) {
        if (MK_STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
            t[K] += now - since;
            return now;
        }
        return 0;
    }
    MK_D unsigned long long now() {
        if (MK_STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long v = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
            return v;
        }
        return 0;
    }
    template <int REGION>
    MK_D void flush(unsigned long long *buf, bool differences) {
        if (MK_STAMP && buf && (threadIdx.x & 63) == 0) {
            const unsigned b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            unsigned long long *p = buf + (size_t)REGION * STAMP_REGION + ((b * 4 + threadIdx.x / 64) % (STAMP_REGION / 8)) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] = differences ? (i + 1 < 8 && t[i + 1] ? t[i + 1] - t[i] : 0) : t[i];
        }
    }
};

// 1-D grids over (group, member): the members of a group share operand tiles (the source tiles of a conversion, the
// twiddle / eval-key tiles of a row tile).  Workgroup b runs on XCD b % 8 (round-robin dispatch), so with 8 | groups a
// group's members are made consecutive in ONE XCD's queue: its tiles are fetched over the fabric once and then hit in
// that L2.  CU-affine form (NttTables::cu_affine): positions qidx, qidx + 32, qidx + 64, ... of an XCD's queue land on
// the same CU, W at a time, and later workgroups inherit the slots of the ones they replace (tools/probe_dispatch.hip:
// first-generation co-residents are exactly 256 blocks apart) -- a group's members are laid out along that direction,
// so the workgroups that share tiles also share a CU and W - 1 of W tile loads can hit in its L1 instead of L2.
// Placement is a speed matter only: any mapping that is a bijection gives the same results.
MK_D void group_member(uint32_t b, uint32_t groups, uint32_t members, uint32_t cu_affine, uint32_t &grp, uint32_t &mem) {
    if (groups % 8 == 0) {
        const uint32_t xcd = b % 8, qidx = b / 8;
        if (cu_affine && groups % 256 == 0) {
            const uint32_t per = 32 * members, blk = qidx / per, r = qidx % per;
            grp = (blk * 32 + r % 32) * 8 + xcd;
            mem = r / 32;
        } else {
            grp = (qidx / members) * 8 + xcd;
            mem = qidx % members;
        }
    } else {
        grp = b / members;
        mem = b % members;
    }
}

// (item, column tile) of source-tile group `grp` in the conversion kernels.  grp % 8 is the XCD; with 8 | items every XCD
// takes the items congruent to it and walks their column tiles consecutively, so the ~18 source tiles an XCD works on at
// any moment cover all 16 column phases (128-byte offsets inside a 2-KiB row) instead of one or two -- the plain
// item-major split hands XCD x only the tiles x and x + 8 (+1 % on the step, +2.3 % together with the CU-affine
// placement, same-box A/B).
MK_D void conv_item_tile(uint32_t grp, uint32_t groups, uint32_t items, uint32_t tiles, uint32_t &item, uint32_t &tile) {
    if (groups % 8 == 0 && items % 8 == 0) {
        const uint32_t u = grp / 8;
        tile = u % tiles;
        item = (u / tiles) * 8 + grp % 8;
    } else {
        item = grp / tiles;
        tile = grp % tiles;
    }
}

// twiddles of one round: w[(1<<s) - 1 + g] = table[(base_eff << s) + g]
template <int LOG_H>
MK_D void load_round_twiddles(const u64 *__restrict__ tw, const u64 *__restrict__ tw_sh, uint32_t base_eff,
                              u64 (&w)[(1 << LOG_H) - 1], u64 (&wp)[(1 << LOG_H) - 1]) {
#pragma unroll
    for (int s = 0; s < LOG_H; ++s) {
        const uint32_t first = base_eff << s;
#pragma unroll
        for (int g = 0; g < (1 << s); ++g) {
            w[(1 << s) - 1 + g] = tw[first + g];
            wp[(1 << s) - 1 + g] = tw_sh[first + g];
        }
    }
}

// round-B twiddles of a row kernel thread (row of the limb, position j in the row) from the packed table
// NttTables::twb / itwb: chunk i of the row is H consecutive 16-byte pairs, one per thread
template <int LOG_H>
MK_D void load_rowb_twiddles(const u64 *__restrict__ twb_limb, uint32_t row, int j, u64 (&w)[(1 << LOG_H) - 1],
                             u64 (&wp)[(1 << LOG_H) - 1]) {
    constexpr int H = 1 << LOG_H;
    const ulong2 *p = reinterpret_cast<const ulong2 *>(twb_limb) + (size_t)row * H * H + j;
#pragma unroll
    for (int i = 0; i < H - 1; ++i) {
        const ulong2 t = p[i * H];
        w[i] = t.x;
        wp[i] = t.y;
    }
}

// log2(H) forward stages on H registers.  Inputs < 8q, outputs < 8q: even stages bring x back below 4q
// before the butterfly (outputs < 6q), odd stages skip the correction (outputs < 8q) -- half the
// conditional subtractions of the classic Harvey schedule; 8q < 2^63 since q < 2^60.
template <int LOG_H>
MK_D void radix_forward(u64 (&x)[1 << LOG_H], const u64 (&w)[(1 << LOG_H) - 1], const u64 (&wp)[(1 << LOG_H) - 1],
                        u64 q, u64 q2) {
    constexpr int H = 1 << LOG_H;
    const u64 q4 = q2 + q2;
#pragma unroll
    for (int s = 0; s < LOG_H; ++s) {
        const int dist = H >> (s + 1);
#pragma unroll
        for (int p = 0; p < H / 2; ++p) {
            const int g = p / dist, k0 = g * 2 * dist + (p % dist);
            if (s % 2 == 0) ct_butterfly_c4(x[k0], x[k0 + dist], w[(1 << s) - 1 + g], wp[(1 << s) - 1 + g], q, q2, q4);
            else ct_butterfly_nc(x[k0], x[k0 + dist], w[(1 << s) - 1 + g], wp[(1 << s) - 1 + g], q, q2);
        }
    }
}

// log2(H) inverse stages on H registers (stage order reversed); values stay in [0,2q)
template <int LOG_H>
MK_D void radix_inverse(u64 (&x)[1 << LOG_H], const u64 (&w)[(1 << LOG_H) - 1], const u64 (&wp)[(1 << LOG_H) - 1],
                        u64 q, u64 q2) {
    constexpr int H = 1 << LOG_H;
#pragma unroll
    for (int s = LOG_H - 1; s >= 0; --s) {
        const int dist = H >> (s + 1);
#pragma unroll
        for (int p = 0; p < H / 2; ++p) {
            const int g = p / dist, k0 = g * 2 * dist + (p % dist);
            gs_butterfly(x[k0], x[k0 + dist], w[(1 << s) - 1 + g], wp[(1 << s) - 1 + g], q, q2);
        }
    }
}

// the same rounds with the pseudo-Mersenne butterflies (LimbConst::pm; bounds next to ct_butterfly_pm_f): inputs < 8U,
// outputs < 7.001U forward; inputs and outputs < 2.375U inverse
template <int LOG_H>
MK_D void radix_forward_pm(u64 (&x)[1 << LOG_H], const u64 (&w)[(1 << LOG_H) - 1], const u64 (&wx)[(1 << LOG_H) - 1],
                           const PmK &P) {
    constexpr int H = 1 << LOG_H;
#pragma unroll
    for (int s = 0; s < LOG_H; ++s) {
        const int dist = H >> (s + 1);
#pragma unroll
        for (int p = 0; p < H / 2; ++p) {
            const int g = p / dist, k0 = g * 2 * dist + (p % dist);
            if (s % 2 == 0) ct_butterfly_pm_f(x[k0], x[k0 + dist], w[(1 << s) - 1 + g], wx[(1 << s) - 1 + g], P);
            else ct_butterfly_pm_n(x[k0], x[k0 + dist], w[(1 << s) - 1 + g], wx[(1 << s) - 1 + g], P);
        }
    }
}
template <int LOG_H>
MK_D void radix_inverse_pm(u64 (&x)[1 << LOG_H], const u64 (&w)[(1 << LOG_H) - 1], const u64 (&wx)[(1 << LOG_H) - 1],
                           const PmK &P) {
    constexpr int H = 1 << LOG_H;
#pragma unroll
    for (int s = LOG_H - 1; s >= 0; --s) {
        const int dist = H >> (s + 1);
#pragma unroll
        for (int p = 0; p < H / 2; ++p) {
            const int g = p / dist, k0 = g * 2 * dist + (p % dist);
            gs_butterfly_pm(x[k0], x[k0 + dist], w[(1 << s) - 1 + g], wx[(1 << s) - 1 + g], P);
        }
    }
}

// ---- fp64 rounds (limbs with LimbConst::fp): x[], w[], wq[] hold DOUBLE bit patterns -----------------
// forward: v = y*w mod q (|v| <= q(0.5 + 1.0001|y|/2^52)); x is re-centred (u = reduce(x), |u| <= 0.51q) on the even
// stages of a round only; with q < 1.25*2^50 every value stays below 3.1q.  inverse: s = x + y is reduced, d = x - y goes straight into the
// product (|d| <= 2.66q < 2^52.1, product output <= 1.33q).  Everything stays an exact integer below 2^53.
MK_D void ct_butterfly_fp(u64 &xb, u64 &yb, u64 wb, u64 wqb, double q, double qinv) {
    const double v = fp_mulmod(bitsd(yb), bitsd(wb), bitsd(wqb), q);
    const double u = fp_reduce(bitsd(xb), q, qinv);
    xb = dbits(u + v);
    yb = dbits(u - v);
}
// forward butterfly without re-centring x (odd stages): with q < 1.25*2^50 the bound X -> 1.31 X + 0.5q after such a
// stage and X -> 1.01q + 0.31 X after a re-centring one settle below 3.1q < 2^52, sums stay exact integers < 2^53
MK_D void ct_butterfly_fp_nr(u64 &xb, u64 &yb, u64 wb, u64 wqb, double q) {
    const double v = fp_mulmod(bitsd(yb), bitsd(wb), bitsd(wqb), q);
    const double u = bitsd(xb);
    xb = dbits(u + v);
    yb = dbits(u - v);
}
MK_D void gs_butterfly_fp(u64 &xb, u64 &yb, u64 wb, u64 wqb, double q, double qinv) {
    const double x = bitsd(xb), y = bitsd(yb);
    xb = dbits(fp_reduce(x + y, q, qinv));
    yb = dbits(fp_mulmod(x - y, bitsd(wb), bitsd(wqb), q));
}
template <int LOG_H>
MK_D void radix_forward_fp(u64 (&x)[1 << LOG_H], const u64 (&w)[(1 << LOG_H) - 1], const u64 (&wq)[(1 << LOG_H) - 1],
                           double q, double qinv) {
    constexpr int H = 1 << LOG_H;
#pragma unroll
    for (int s = 0; s < LOG_H; ++s) {
        const int dist = H >> (s + 1);
#pragma unroll
        for (int p = 0; p < H / 2; ++p) {
            const int g = p / dist, k0 = g * 2 * dist + (p % dist);
            if (s % 2 == 0) ct_butterfly_fp(x[k0], x[k0 + dist], w[(1 << s) - 1 + g], wq[(1 << s) - 1 + g], q, qinv);
            else ct_butterfly_fp_nr(x[k0], x[k0 + dist], w[(1 << s) - 1 + g], wq[(1 << s) - 1 + g], q);
        }
    }
}
template <int LOG_H>
MK_D void radix_inverse_fp(u64 (&x)[1 << LOG_H], const u64 (&w)[(1 << LOG_H) - 1], const u64 (&wq)[(1 << LOG_H) - 1],
                           double q, double qinv) {
    constexpr int H = 1 << LOG_H;
#pragma unroll
    for (int s = LOG_H - 1; s >= 0; --s) {
        const int dist = H >> (s + 1);
#pragma unroll
        for (int p = 0; p < H / 2; ++p) {
            const int g = p / dist, k0 = g * 2 * dist + (p % dist);
            gs_butterfly_fp(x[k0], x[k0 + dist], w[(1 << s) - 1 + g], wq[(1 << s) - 1 + g], q, qinv);
        }
    }
}
// dispatch on the arithmetic of the kernel instance (FP is a template parameter of every radix kernel: the two
// arithmetics get their own register allocation; a launch of one instance skips the limbs of the other class)
template <int LOG_H, int AR>
MK_D void radix_forward_any(u64 (&x)[1 << LOG_H], const u64 (&w)[(1 << LOG_H) - 1], const u64 (&wp)[(1 << LOG_H) - 1],
                            const LimbConst &lc) {
    if (AR == AR_FP) radix_forward_fp<LOG_H>(x, w, wp, lc.qd, lc.qinv);
    else if (AR == AR_PM) radix_forward_pm<LOG_H>(x, w, wp, pm_consts(lc));
    else radix_forward<LOG_H>(x, w, wp, lc.q, lc.q2);
}
template <int LOG_H, int AR>
MK_D void radix_inverse_any(u64 (&x)[1 << LOG_H], const u64 (&w)[(1 << LOG_H) - 1], const u64 (&wp)[(1 << LOG_H) - 1],
                            const LimbConst &lc) {
    if (AR == AR_FP) radix_inverse_fp<LOG_H>(x, w, wp, lc.qd, lc.qinv);
    else if (AR == AR_PM) radix_inverse_pm<LOG_H>(x, w, wp, pm_consts(lc));
    else radix_inverse<LOG_H>(x, w, wp, lc.q, lc.q2);
}

// The same rounds with the twiddles of a stage fetched right before the stage instead of all 2 (H - 1) words up front:
// `pair_at(i)` returns twiddle i of the round (w, companion) -- an LDS read in the column kernels (ColTwB), whose latency
// is short enough to be taken stage by stage; the 60 registers of a whole round's twiddles shrink to the 32 of its last stage.
template <int AR, bool EVEN>
MK_D void butterfly_forward_any(u64 &x, u64 &y, u64 w, u64 wp, const LimbConst &lc, const PmK &P) {
    if (AR == AR_FP) {
        if (EVEN) ct_butterfly_fp(x, y, w, wp, lc.qd, lc.qinv);
        else ct_butterfly_fp_nr(x, y, w, wp, lc.qd);
    } else if (AR == AR_PM) {
        if (EVEN) ct_butterfly_pm_f(x, y, w, wp, P);
        else ct_butterfly_pm_n(x, y, w, wp, P);
    } else {
        if (EVEN) ct_butterfly_c4(x, y, w, wp, lc.q, lc.q2, lc.q2 + lc.q2);
        else ct_butterfly_nc(x, y, w, wp, lc.q, lc.q2);
    }
}
template <int AR>
MK_D void butterfly_inverse_any(u64 &x, u64 &y, u64 w, u64 wp, const LimbConst &lc, const PmK &P) {
    if (AR == AR_FP) gs_butterfly_fp(x, y, w, wp, lc.qd, lc.qinv);
    else if (AR == AR_PM) gs_butterfly_pm(x, y, w, wp, P);
    else gs_butterfly(x, y, w, wp, lc.q, lc.q2);
}
template <int LOG_H, int AR, int S_, typename F>
MK_D void radix_forward_stage(u64 (&x)[1 << LOG_H], F &pair_at, const LimbConst &lc, const PmK &P) {
    constexpr int H = 1 << LOG_H, dist = H >> (S_ + 1);
    u64 w[1 << S_], wp[1 << S_];
    __builtin_amdgcn_sched_barrier(0);  // this stage's twiddles are fetched here, not at the top of the round
#pragma unroll
    for (int g = 0; g < (1 << S_); ++g) {
        const ulong2 t = pair_at((1 << S_) - 1 + g);
        w[g] = t.x;
        wp[g] = t.y;
    }
#pragma unroll
    for (int p = 0; p < H / 2; ++p) {
        const int g = p / dist, k0 = g * 2 * dist + (p % dist);
        butterfly_forward_any<AR, S_ % 2 == 0>(x[k0], x[k0 + dist], w[g], wp[g], lc, P);
    }
}
template <int LOG_H, int AR, int S_, typename F>
MK_D void radix_inverse_stage(u64 (&x)[1 << LOG_H], F &pair_at, const LimbConst &lc, const PmK &P) {
    constexpr int H = 1 << LOG_H, dist = H >> (S_ + 1);
    u64 w[1 << S_], wp[1 << S_];
    __builtin_amdgcn_sched_barrier(0);  // this stage's twiddles are fetched here, not at the top of the round
#pragma unroll
    for (int g = 0; g < (1 << S_); ++g) {
        const ulong2 t = pair_at((1 << S_) - 1 + g);
        w[g] = t.x;
        wp[g] = t.y;
    }
#pragma unroll
    for (int p = 0; p < H / 2; ++p) {
        const int g = p / dist, k0 = g * 2 * dist + (p % dist);
        butterfly_inverse_any<AR>(x[k0], x[k0 + dist], w[g], wp[g], lc, P);
    }
}
template <int LOG_H, int AR, typename F>
MK_D void radix_forward_staged(u64 (&x)[1 << LOG_H], F &&pair_at, const LimbConst &lc) {
    const PmK P = AR == AR_PM ? pm_consts(lc) : PmK{};
    radix_forward_stage<LOG_H, AR, 0>(x, pair_at, lc, P);
    if constexpr (LOG_H > 1) radix_forward_stage<LOG_H, AR, 1>(x, pair_at, lc, P);
    if constexpr (LOG_H > 2) radix_forward_stage<LOG_H, AR, 2>(x, pair_at, lc, P);
    if constexpr (LOG_H > 3) radix_forward_stage<LOG_H, AR, 3>(x, pair_at, lc, P);
}
template <int LOG_H, int AR, typename F>
MK_D void radix_inverse_staged(u64 (&x)[1 << LOG_H], F &&pair_at, const LimbConst &lc) {
    const PmK P = AR == AR_PM ? pm_consts(lc) : PmK{};
    if constexpr (LOG_H > 3) radix_inverse_stage<LOG_H, AR, 3>(x, pair_at, lc, P);
    if constexpr (LOG_H > 2) radix_inverse_stage<LOG_H, AR, 2>(x, pair_at, lc, P);
    if constexpr (LOG_H > 1) radix_inverse_stage<LOG_H, AR, 1>(x, pair_at, lc, P);
    radix_inverse_stage<LOG_H, AR, 0>(x, pair_at, lc, P);
}

// ---- LDS layouts -------------------------------------------------------------------
// column tile: R rows x S columns (S = 256/H), stored as H blocks of H rows; blocks are padded by
// 16 words when S == 16 so that two neighbouring blocks land in different halves of a bank row.
template <int LOG_H>
struct ColTile {
    static constexpr int H = 1 << LOG_H, S = 256 / H, BLK = H * S + (S == 16 ? 16 : 0);
    static constexpr int WORDS = H * BLK;
    // row = blk*H + kk
    static MK_D int at(int blk, int kk, int c) { return blk * BLK + kk * S + c; }
};
// row tile: S rows of R contiguous words; one pad word per H words, row stride R + H
template <int LOG_H>
struct RowTile {
    static constexpr int H = 1 << LOG_H, S = 256 / H, R = H * H, RS = R + H;
    static constexpr int WORDS = S * RS;
    static MK_D int at(int g, int x) { return g * RS + x + (x >> LOG_H); }
};

// Column-pass round-B twiddles through LDS (round 3).  In round B thread (j, c) needs the H - 1 pairs of base H + j -- the
// same for the S columns c, so as per-thread global loads each wave instruction fetches 64 / S distinct words, and the 2 (H - 1)
// loads per thread are a quarter to a half of everything the column kernels send through the CU's texture addresser, the
// unit the PMC counters show saturated in them (TA busy 0.77-0.86, profiles/r03_mempipe_counters.txt).  A wave owns
// JW = 64 / S values of j; their stage-s entries are ONE contiguous run of JW 2^s table words starting at (H + j0) << s,
// so the wave loads its JW (H - 1) pairs once, coalesced, into a wave-private LDS strip (no workgroup barrier) and every
// thread takes its pairs with broadcast LDS reads.
template <int LOG_H>
struct ColTwB {
    static constexpr int H = 1 << LOG_H, S = 256 / H, JW = 64 / S, PAIRS = JW * (H - 1);  // round B, per wave
    static constexpr int STRIP = PAIRS + (H - 1);  // + the H - 1 pairs of round A (base 1: the same for the whole limb)
    static constexpr int WORDS = (NTT_THREADS / 64) * STRIP * 2;
    static_assert(PAIRS <= 64 && H - 1 <= 64, "one pair per lane and round");
    static MK_D void stage(u64 *strip, const u64 *tw, const u64 *tw_sh) {  // strip: this workgroup's WORDS words
        const int lane = threadIdx.x % 64, wv = threadIdx.x / 64;
        if (lane < PAIRS) {
            const int s = 31 - __clz(lane / JW + 1), off = lane - JW * ((1 << s) - 1);
            const uint32_t idx = ((uint32_t)(H + JW * wv) << s) + (uint32_t)off;
            *reinterpret_cast<ulong2 *>(strip + (wv * STRIP + lane) * 2) = ulong2{tw[idx], tw_sh[idx]};
        }
        if (lane < H - 1)  // round A: table entries 1 .. H - 1 in order (entry (1 << s) + g is twiddle (1 << s) - 1 + g of the round)
            *reinterpret_cast<ulong2 *>(strip + (wv * STRIP + PAIRS + lane) * 2) = ulong2{tw[1 + lane], tw_sh[1 + lane]};
    }
    static MK_D void fetch(const u64 *strip, int j, u64 (&w)[H - 1], u64 (&wp)[H - 1]) {
        const int wv = threadIdx.x / 64, jj = j - JW * wv;
#pragma unroll
        for (int s = 0; s < LOG_H; ++s)
#pragma unroll
            for (int g = 0; g < (1 << s); ++g) {
                const ulong2 t = *reinterpret_cast<const ulong2 *>(strip + (wv * STRIP + JW * ((1 << s) - 1) + (jj << s) + g) * 2);
                w[(1 << s) - 1 + g] = t.x;
                wp[(1 << s) - 1 + g] = t.y;
            }
    }
    // twiddle i of round B / round A of this thread, for radix_*_staged
    static MK_D ulong2 pair_b(const u64 *strip, int j, int i) {
        const int wv = threadIdx.x / 64, jj = j - JW * wv;
        const int s = 31 - __clz(i + 1), g = i - ((1 << s) - 1);  // constants after unrolling
        return *reinterpret_cast<const ulong2 *>(strip + (wv * STRIP + JW * ((1 << s) - 1) + (jj << s) + g) * 2);
    }
    static MK_D ulong2 pair_a(const u64 *strip, int i) {
        return *reinterpret_cast<const ulong2 *>(strip + ((threadIdx.x / 64) * STRIP + PAIRS + i) * 2);
    }
    static MK_D void fetch_a(const u64 *strip, u64 (&w)[H - 1], u64 (&wp)[H - 1]) {
        const int wv = threadIdx.x / 64;
#pragma unroll
        for (int i = 0; i < H - 1; ++i) {
            const ulong2 t = *reinterpret_cast<const ulong2 *>(strip + (wv * STRIP + PAIRS + i) * 2);
            w[i] = t.x;
            wp[i] = t.y;
        }
    }
};

// Row-pass round A twiddles through LDS.  In round A the H threads of a row all need the SAME H-1 twiddles
// (index ((r1 + row) << s) + g), so per-thread global loads fetch 4 distinct words per wave-instruction.  For the
// S consecutive rows of a tile the stage-s entries are ONE contiguous run of S*2^s words starting at
// (r1 + row0) << s: the workgroup loads the S*(H-1) words of each table once, coalesced, and every thread then
// takes its H-1 pairs with broadcast LDS reads.
template <int LOG_H>
struct RowTwA {
    static constexpr int H = 1 << LOG_H, S = 256 / H, WORDS = S * (H - 1);  // per table
    static MK_D void stage(u64 *ldsw, u64 *ldswp, const u64 *tw, const u64 *tw_sh, uint32_t base0) {
        for (int e = threadIdx.x; e < WORDS; e += NTT_THREADS) {
            const int s = 31 - __clz(e / S + 1);
            const int off = e - S * ((1 << s) - 1);
            const uint32_t idx = (base0 << s) + (uint32_t)off;
            ldsw[e] = tw[idx];
            ldswp[e] = tw_sh[idx];
        }
    }
    static MK_D void fetch(const u64 *ldsw, const u64 *ldswp, int g, u64 (&w)[H - 1], u64 (&wp)[H - 1]) {
#pragma unroll
        for (int s = 0; s < LOG_H; ++s)
#pragma unroll
            for (int gg = 0; gg < (1 << s); ++gg) {
                const int e = S * ((1 << s) - 1) + (g << s) + gg;
                w[(1 << s) - 1 + gg] = ldsw[e];
                wp[(1 << s) - 1 + gg] = ldswp[e];
            }
    }
};

// Wave-local row tiles.  A row of R = H*H words is handled by H threads, i.e. a wavefront owns 64/H complete rows in
// BOTH rounds (the exchange between the rounds stays inside a row).  When the copy-in / copy-out phases use the same
// ownership (wave_pair below: one wave instruction = 1 KiB contiguous), no LDS word is ever touched by two waves and
// the workgroup barriers become wave-level ordering points: a wave's own ds_write / ds_read execute in order.
MK_D void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// The ownership invariants that make the wave-level hand-off legal, checked at compile time for every tile geometry the
// row kernels instantiate.  A row of R = H*H words is transformed by H threads with consecutive thread ids, so
//   (1) a wavefront holds whole rows in BOTH rounds:            64 % H == 0, rows per wave RW = 64 / H >= 1;
//   (2) the copy-in / copy-out phases use the SAME ownership:   wave w touches pairs [w * PER_WAVE, (w+1) * PER_WAVE) with
//       PER_WAVE = RW * R / 2 = exactly the pairs of rows [w * RW, (w+1) * RW), and PAIRS * 64 == PER_WAVE, i.e. the
//       PAIRS accesses of wave_pair() cover the wave's rows once and nothing else;
//   (3) the workgroup is a whole number of wavefronts and of rows: NTT_THREADS % 64 == 0, S * H == NTT_THREADS.
// Hence no LDS word of the tile is read or written by two different waves, and within one wave DS operations execute in
// program order (an ISA property of CDNA, not of the HIP language): wave_lds_sync() only has to stop the COMPILER from
// moving LDS accesses across it.  The round-A twiddle area is the one shared region: every wave stages and reads only the
// entries of its own rows (stage_twiddles_wave), same argument.
template <int LOG_H>
struct WaveOwnership {
    static constexpr int H = 1 << LOG_H, R = H * H, S = 256 / H, RW = 64 / H;
    static constexpr int PER_WAVE = RW * R / 2, PAIRS = S * R / 2 / NTT_THREADS;
    static_assert(64 % H == 0 && RW >= 1, "a wavefront must hold whole rows");
    static_assert(NTT_THREADS % 64 == 0 && S * H == NTT_THREADS, "workgroup = whole wavefronts = whole rows");
    static_assert(PAIRS * 64 == PER_WAVE, "wave_pair() must cover exactly the wave's own rows");
    static_assert((NTT_THREADS / 64) * RW == S, "the waves' row ranges partition the tile");
};
// i-th 16-byte access (pair of words, index into the tile's pairs) of this lane inside its wave's rows
template <int LOG_H>
MK_D int wave_pair(int i) {
    constexpr int PER_WAVE = WaveOwnership<LOG_H>::PER_WAVE;
    return (int)(threadIdx.x / 64) * PER_WAVE + (int)(threadIdx.x % 64) + 64 * i;
}
// round-A twiddles of this wave's rows only (same LDS layout as RowTwA::stage; RW = 64/H rows per wave)
template <int LOG_H>
MK_D void stage_twiddles_wave(u64 *ldsw, u64 *ldswp, const u64 *tw, const u64 *tw_sh, uint32_t base0) {
    constexpr int H = 1 << LOG_H, S = 256 / H, RW = 64 / H;
    const int wv = threadIdx.x / 64;
    for (int e = threadIdx.x % 64; e < RW * (H - 1); e += 64) {
        const int s = 31 - __clz(e / RW + 1);
        const int off = e - RW * ((1 << s) - 1);
        const uint32_t idx = ((base0 + (uint32_t)(RW * wv)) << s) + (uint32_t)off;
        const int dst = S * ((1 << s) - 1) + ((RW * wv) << s) + off;
        ldsw[dst] = tw[idx];
        ldswp[dst] = tw_sh[idx];
    }
}

// ---- kernels -----------------------------------------------------------------------

// Forward column pass, everything after the H input words of this thread (rows j + H k, column c) are
// in x[]: round A, LDS exchange, round B, store rows H j + k (lazy [0,8q): the row pass finishes).
template <int LOG_H, int AR>
MK_D void col_forward_finish(u64 (&x)[1 << LOG_H], u64 *lds, const u64 *tw, const u64 *tw_sh, const LimbConst &lc,
                             int j, int c, u64 *dst_col, uint32_t r2, Stamper *st = nullptr) {
    using TL = ColTile<LOG_H>;
    constexpr int H = TL::H;
    const u64 *strip = lds + TL::WORDS;
    ColTwB<LOG_H>::stage(lds + TL::WORDS, tw, tw_sh);  // both rounds' twiddles of this wave's rows, read stage by stage
    wave_lds_sync();
    radix_forward_staged<LOG_H, AR>(x, [&](int i) { return ColTwB<LOG_H>::pair_a(strip, i); }, lc);
    if (MK_STAMP && st) st->template mark<2>();
#pragma unroll
    for (int k = 0; k < H; ++k) lds[TL::at(k, j, c)] = x[k];  // row j + H k
    __syncthreads();
    if (MK_STAMP && st) st->template mark<3>();
#pragma unroll
    for (int k = 0; k < H; ++k) x[k] = lds[TL::at(j, k, c)];  // row H j + k
    radix_forward_staged<LOG_H, AR>(x, [&](int i) { return ColTwB<LOG_H>::pair_b(strip, j, i); }, lc);
    if (MK_STAMP && st) st->template mark<4>();
#pragma unroll
    for (int k = 0; k < H; ++k) st_pass(dst_col + (size_t)(H * j + k) * r2, x[k]);  // lazy u64, or doubles on an fp limb
    if (MK_STAMP && st) st->template mark<5>();
}

// Column pass over R1 = H*H rows: one workgroup = S = 256/H adjacent columns.  Global accesses are
// S x 8-B row segments (128 B at H = 16); one LDS exchange between the two rounds.
template <int LOG_H, bool INV, int AR>
__global__ __launch_bounds__(NTT_THREADS, 4) void k_ntt_col_r(NttIo io, NttTables T, const u64 *scale,
                                                           const u64 *scale_sh, int pack) {
    using TL = ColTile<LOG_H>;
    constexpr int H = TL::H, S = TL::S;
    __shared__ u64 lds[TL::WORDS + ColTwB<LOG_H>::WORDS];
    const uint32_t poly = blockIdx.y / io.nsel, sl = nth_set_bit(io.slot_mask, blockIdx.y % io.nsel);
    if (ntt_slot_skipped(io, poly, io.vslot0 + sl)) return;  // block-uniform
    const uint32_t id = limb_id_of(io.vslot0 + sl, io.nl, T.L);
    const LimbConst lc = T.limb[id];
    if ((lc.fp != 0) != (AR == AR_FP)) return;  // never: the host selects the slots of this instance's class
    const uint32_t n = 1u << T.log_n, r2 = 1u << T.log_r2;
    const int c = threadIdx.x % S, j = threadIdx.x / S;
    const u64 *src = io.in + (size_t)poly * io.in_stride + (size_t)(io.in_slot0 + sl) * n + blockIdx.x * S + c;
    u64 *dst = io.out + (size_t)poly * io.out_stride + (size_t)(io.out_slot0 + sl) * n + blockIdx.x * S + c;
    const u64 *tw = (INV ? T.itw : T.tw) + (size_t)id * n;
    const u64 *tw_sh = (INV ? T.itw_sh : T.tw_sh) + (size_t)id * n;
    u64 x[H];
    if (!INV) {
#pragma unroll
        for (int k = 0; k < H; ++k) x[k] = ld_pass(src + (size_t)(j + H * k) * r2);
        if (AR == AR_FP) {  // canonical residues -> doubles (exact, q < 2^51)
#pragma unroll
            for (int k = 0; k < H; ++k) x[k] = dbits((double)x[k]);
        }
        col_forward_finish<LOG_H, AR>(x, lds, tw, tw_sh, lc, j, c, dst, r2);
    } else {
        ColTwB<LOG_H>::stage(lds + TL::WORDS, tw, tw_sh);
#pragma unroll
        for (int k = 0; k < H; ++k) x[k] = ld_pass(src + (size_t)(H * j + k) * r2);  // from the row pass: doubles on an fp limb
        wave_lds_sync();  // the strip is this wave's own
        const u64 *strip = lds + TL::WORDS;
        radix_inverse_staged<LOG_H, AR>(x, [&](int i) { return ColTwB<LOG_H>::pair_b(strip, j, i); }, lc);
#pragma unroll
        for (int k = 0; k < H; ++k) lds[TL::at(j, k, c)] = x[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < H; ++k) x[k] = lds[TL::at(k, j, c)];
        radix_inverse_staged<LOG_H, AR>(x, [&](int i) { return ColTwB<LOG_H>::pair_a(strip, i); }, lc);
        // scale by N^-1 (x folded constant): table entries are (u64, Shoup) or, on an fp limb, (double, double/q)
        const u64 sc = scale ? scale[id] : ((AR == AR_FP) ? dbits(lc.ninv_d) : lc.ninv);
        const u64 sc_sh = scale ? scale_sh[id] : ((AR == AR_FP) ? dbits(lc.ninv_qd) : lc.ninv_sh);
        if ((AR == AR_FP) && pack == 2) {  // (the output form is decided once, not per word: 16 branches inside the loop otherwise)
#pragma unroll
            for (int k = 0; k < H; ++k) {
                // canonical residue as a double for k_conv_col's fp64 products: |x| <= 1.33 q -> |s| <= 0.92 q, so one
                // conditional add lands in [0, q) (the same value fp_to_canonical returns)
                double sd = fp_mulmod(bitsd(x[k]), bitsd(sc), bitsd(sc_sh), lc.qd);
                sd = sd < 0.0 ? sd + lc.qd : sd;
                st_pass(dst + (size_t)(j + H * k) * r2, dbits(sd));
            }
        } else if (pack) {
#pragma unroll
            for (int k = 0; k < H; ++k) {
                const u64 v = (AR == AR_FP) ? fp_to_canonical(fp_mulmod(bitsd(x[k]), bitsd(sc), bitsd(sc_sh), lc.qd), lc.qd, lc.qinv)
                                 : shoup_mul(x[k], sc, sc_sh, lc.q);
                st_pass(dst + (size_t)(j + H * k) * r2, pack30(v));  // packed halves feed k_conv_col directly
            }
        } else {
#pragma unroll
            for (int k = 0; k < H; ++k) {
                const u64 v = (AR == AR_FP) ? fp_to_canonical(fp_mulmod(bitsd(x[k]), bitsd(sc), bitsd(sc_sh), lc.qd), lc.qd, lc.qinv)
                                 : shoup_mul(x[k], sc, sc_sh, lc.q);
                st_pass(dst + (size_t)(j + H * k) * r2, v);
            }
        }
    }
}

// Approximate base conversion (ApproxSwitchCRTBasis) fused into the forward column pass of the
// converted limb (sources arrive as packed 30-bit halves, see pack30): the H input words of a thread are computed as  sum_i x_i * [S/s_i]_t  from the N_IN
// source limbs instead of being loaded.  The sources are COEFFICIENT-format limbs that the preceding
// inverse transform already multiplied by [(S/s_i)^-1]_{s_i} (folded into its N^-1 scaling).
// Grid: 1-D, (item, target limb, column tile); the n_out workgroups that share one source tile are made
// neighbours inside one XCD's queue so the 4 source tiles are fetched from HBM once and then hit in L2.
struct ConvIo {
    const u64 *in;      // [items][in_slots][N]
    u64 *out;           // [items][out_slots][N]
    size_t in_stride, out_stride;
    uint32_t items;
    unsigned long long target_mask;  // targets (indices into cv.dst_*) of this instance's arithmetic class
    uint32_t nsel;                   // popcount(target_mask)
};
#ifndef MK_CONV_DEPTH
#define MK_CONV_DEPTH 6  // source slices (outputs) the conversion's loads run ahead
#endif
// timing probes of the conversion's source loads (diagnostic builds; results are wrong, the instruction stream is the same):
// MK_PROBE = 1: every load of a thread reads row j of its source (16 distinct lines per wave instead of 256: L1-resident);
// MK_PROBE = 2: no loads at all (the sources are register garbage)
#ifndef MK_PROBE
#define MK_PROBE 0
#endif
#if MK_PROBE == 1
#define MK_PROBE_SRC(src, limb_off, row_off) (src)[(limb_off) + (size_t)j * r2 + 0 * (row_off)]
#elif MK_PROBE == 2
#define MK_PROBE_SRC(src, limb_off, row_off) ((u64)threadIdx.x * 0x9E3779B97F4A7C15ull + (limb_off) + (row_off))
#else
#define MK_PROBE_SRC(src, limb_off, row_off) (src)[(limb_off) + (row_off)]
#endif

// canonical integer below 2^52 held in a double -> its 30-bit halves (what split30 gives for the u64)
MK_D void split30_d(u64 dbl_bits, uint32_t &lo, uint32_t &hi) {
    const u64 b = dbits(bitsd(dbl_bits) + 4503599627370496.0) & 0xFFFFFFFFFFFFFull;  // mantissa of 2^52 + v is v
    lo = (uint32_t)b & 0x3FFFFFFFu;
    hi = (uint32_t)(b >> 30);
}
// form in which source i reaches k_conv_col under SRCMODE: 0 = every source as packed 30-bit halves, 1 = every source
// a canonical double (all sources of the digit are fp64-class limbs), 2 = source 0 packed (60-bit q_0), the rest doubles
template <int SRCMODE>
MK_D constexpr bool src_is_double(int i) {
    return SRCMODE == 1 || (SRCMODE == 2 && i > 0);
}
// per-target constants of a conversion with at most 4 sources (scalar registers: the target is workgroup-uniform)
template <int N_IN>
struct ConvConst {
    uint32_t h0[N_IN], h1[N_IN];  // [S/s_i]_t as 30-bit halves (integer path)
    double hd[N_IN], hq[N_IN];    // [S/s_i]_t and its quotient by t as doubles (fp64 path)
};
template <int N_IN, int AR, int SRCMODE, typename CONV>
MK_D void conv_consts(const CONV &cv, uint32_t jt, ConvConst<N_IN> &k) {
#pragma unroll
    for (int i = 0; i < N_IN; ++i) {
        if ((AR == AR_FP) && src_is_double<SRCMODE>(i)) {
            k.hd[i] = cv.hat_d[i * cv.n_out + jt];
            k.hq[i] = cv.hatq_d[i * cv.n_out + jt];
        } else {
            split30(cv.hat[i * cv.n_out + jt], k.h0[i], k.h1[i]);
        }
    }
}
// 30-bit halves of the sources that enter the integer column accumulation (packed sources as they are, double sources
// split back); shared by every target computed from the same sources
template <int N_IN, int SRCMODE>
MK_D void conv_split_sources(const u64 (&p)[N_IN], uint32_t (&a0)[N_IN], uint32_t (&a1)[N_IN]) {
#pragma unroll
    for (int i = 0; i < N_IN; ++i) {
        if (src_is_double<SRCMODE>(i)) {
            split30_d(p[i], a0[i], a1[i]);
        } else {
            a0[i] = (uint32_t)p[i];
            a1[i] = (uint32_t)(p[i] >> 32);
        }
    }
}
// one output  sum_i x_i [S/s_i]_t  in the range the first butterfly round of the target's arithmetic accepts.
// Packed sources and [S/s_i]_t are below 2^60: 30-bit column accumulation, pure v_mad_u64_u32 chains.  Sources that
// arrive as doubles (fp64-class limbs, below 1.25 * 2^50): an fp64-class target takes their products mod t on the FMA unit
// (6 operations instead of 4 mads + a share of the Barrett step); an integer-class target uses the split halves.
template <int N_IN, int AR, int SRCMODE>
MK_D u64 conv_output(const u64 (&p)[N_IN], const uint32_t (&a0)[N_IN], const uint32_t (&a1)[N_IN], const ConvConst<N_IN> &k,
                     const LimbConst &lc) {
    if ((AR == AR_FP) && SRCMODE != 0) {
        double acc = 0.0;
        if (SRCMODE == 2) {  // the packed source(s): column accumulation, below 4q < 2^53
            Cols ia{0, 0, 0};
#pragma unroll
            for (int i = 0; i < N_IN; ++i)
                if (!src_is_double<SRCMODE>(i)) mac_cols(ia, (uint32_t)p[i], (uint32_t)(p[i] >> 32), k.h0[i], k.h1[i]);
            acc = (double)reduce_cols_lazy(ia, lc);
        }
#pragma unroll
        for (int i = 0; i < N_IN; ++i)
            if (src_is_double<SRCMODE>(i)) acc += fp_mulmod(bitsd(p[i]), k.hd[i], k.hq[i], lc.qd);
        // |acc| <= 4q + 4 * 0.8q < 2^53: exact; into the rounds' range
        return dbits(fp_reduce(acc, lc.qd, lc.qinv));
    }
    Cols acc{0, 0, 0};
#pragma unroll
    for (int i = 0; i < N_IN; ++i) mac_cols(acc, a0[i], a1[i], k.h0[i], k.h1[i]);
    // < 4q (< 2.1U on a pseudo-Mersenne limb): the first butterfly stage accepts < 8q
    const u64 v = AR == AR_PM ? pm_reduce_cols(acc, pm_consts(lc)) : reduce_cols_lazy(acc, lc);
    return AR == AR_FP ? dbits(fp_reduce((double)v, lc.qd, lc.qinv)) : v;  // < 4q < 2^53: exact in a double
}

template <int LOG_H, int N_IN, int AR, typename CONV, int SRCMODE = 0>
__global__ __launch_bounds__(NTT_THREADS, 4) void k_conv_col(ConvIo io, NttTables T, CONV cv) {
    using TL = ColTile<LOG_H>;
    constexpr int H = TL::H, S = TL::S;
    static_assert(SRCMODE == 0 || N_IN <= 4, "double sources: at most 4 per digit");
    __shared__ u64 lds[TL::WORDS + ColTwB<LOG_H>::WORDS];
    const uint32_t n = 1u << T.log_n, r2 = 1u << T.log_r2, tiles = r2 / S;
    const uint32_t groups = io.items * tiles;  // source tiles
    uint32_t grp, jt;
    group_member(blockIdx.x, groups, io.nsel, T.cu_affine, grp, jt);
    jt = nth_set_bit(io.target_mask, jt);
    uint32_t item, tile;
    conv_item_tile(grp, groups, io.items, tiles, item, tile);
    const uint32_t id = cv.dst_id[jt];
    const LimbConst lc = T.limb[id];
    if ((lc.fp != 0) != (AR == AR_FP)) return;  // block-uniform
    const int c = threadIdx.x % S, j = threadIdx.x / S;
    const u64 *src = io.in + (size_t)item * io.in_stride + tile * S + c;
    u64 *dst = io.out + (size_t)item * io.out_stride + (size_t)cv.dst_slot[jt] * n + tile * S + c;
    u64 x[H];
    Stamper stm;
    stm.mark<0>();
    if constexpr (N_IN <= 4) {
        ConvConst<N_IN> kc;
        conv_consts<N_IN, AR, SRCMODE>(cv, jt, kc);
        // Source loads run MK_CONV_DEPTH outputs ahead of their use.  Left to itself the compiler keeps about 8 loads in
        // flight per wave (s_waitcnt vmcnt(7) before every product); a ring of DEPTH source slices held in registers makes
        // it DEPTH * N_IN loads (+1 % on the step: the conversion phase -- three quarters of a wave's lifetime by the
        // in-kernel stamps, profiles/r03_stamps_conv_qsum.txt -- is bound by what the CU's L1 can pull from L2, not by the
        // number of loads in flight)
        constexpr int DEPTH = MK_CONV_DEPTH < H ? MK_CONV_DEPTH : H;
        u64 ring[DEPTH][N_IN];
#pragma unroll
        for (int k = 0; k < DEPTH; ++k)
#pragma unroll
            for (int i = 0; i < N_IN; ++i) ring[k][i] = MK_PROBE_SRC(src, (size_t)cv.src_slot[i] * n, (size_t)(j + H * k) * r2);
#pragma unroll
        for (int k = 0; k < H; ++k) {
            u64 p[N_IN];
#pragma unroll
            for (int i = 0; i < N_IN; ++i) p[i] = ring[k % DEPTH][i];
            if (k + DEPTH < H) {  // refill the slice just taken
#pragma unroll
                for (int i = 0; i < N_IN; ++i)
                    ring[k % DEPTH][i] = MK_PROBE_SRC(src, (size_t)cv.src_slot[i] * n, (size_t)(j + H * (k + DEPTH)) * r2);
            }
            uint32_t a0[N_IN], a1[N_IN];
            if (!((AR == AR_FP) && SRCMODE != 0)) conv_split_sources<N_IN, SRCMODE>(p, a0, a1);
            x[k] = conv_output<N_IN, AR, SRCMODE>(p, a0, a1, kc, lc);
            // the output is materialised here and the refills stay where they are: otherwise the arithmetic sinks to its first
            // use (the butterflies) while all 16 slices' loads stay at the top, and what does not fit is parked in scratch
            asm volatile("" : "+v"(x[k]));
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // 5..8 sources (e.g. alpha = K = 7 at L = 20): same 30-bit columns with the middle one split in two
        uint32_t h0[N_IN], h1[N_IN];
#pragma unroll
        for (int i = 0; i < N_IN; ++i) split30(cv.hat[i * cv.n_out + jt], h0[i], h1[i]);
#pragma unroll
        for (int k = 0; k < H; ++k) {
            Cols4 acc{0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < N_IN; ++i) {
                const u64 p = src[(size_t)cv.src_slot[i] * n + (size_t)(j + H * k) * r2];
                mac_cols4(acc, (uint32_t)p, (uint32_t)(p >> 32), h0[i], h1[i]);
            }
            x[k] = reduce_cols4(acc, lc);
            if (AR == AR_FP) x[k] = dbits(fp_reduce((double)x[k], lc.qd, lc.qinv));
            asm volatile("" : "+v"(x[k]));  // as above: keep the arithmetic beside its loads (111 spilled registers otherwise)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    stm.mark<1>();
    col_forward_finish<LOG_H, AR>(x, lds, T.tw + (size_t)id * n, T.tw_sh + (size_t)id * n, lc, j, c, dst, r2, &stm);
    if (AR == AR_FP) stm.flush<0>(T.stamps, true);
    else stm.flush<1>(T.stamps, true);
}

// k_conv_col for TWO targets of one arithmetic class per workgroup (at most 4 sources).  Every target limb's workgroup
// pulls the same 4 source tiles (128 KiB) through its CU's L1 to produce 32 KiB, and that pull -- not the arithmetic, not
// the number of loads in flight -- is what the conversion phase waits for (in-kernel stamps: 20-24 k of a wave's 31 k
// cycles).  Two conversions from one pass over the sources halve it; the second target's 16 words wait in registers
// while the first goes through its column pass.  Same-box A/B on the step: +1.5 % (integer-class targets), +1.7 %
// (fp64-class), +3.2 % both (29.64 -> 30.58 k ct/s).  Round 2 and this round's first attempt measured the same idea at
// -10...-22 %: their kernels spilled.  The last workgroup of an odd target count carries one live target.
#ifndef MK_CONV2_DEPTH
#define MK_CONV2_DEPTH 4
#endif
template <int LOG_H, int N_IN, int AR, typename CONV, int SRCMODE = 0>
__global__ __launch_bounds__(NTT_THREADS, 3) void k_conv_col2(ConvIo io, NttTables T, CONV cv) {
    using TL = ColTile<LOG_H>;
    constexpr int H = TL::H, S = TL::S;
    static_assert(SRCMODE == 0 || N_IN <= 4, "double sources: at most 4 per digit");
    __shared__ u64 lds[TL::WORDS + ColTwB<LOG_H>::WORDS];
    const uint32_t n = 1u << T.log_n, r2 = 1u << T.log_r2, tiles = r2 / S;
    const uint32_t groups = io.items * tiles, ntg = (io.nsel + 1) / 2;
    uint32_t grp, jg;
    group_member(blockIdx.x, groups, ntg, T.cu_affine, grp, jg);
    const bool two = jg * 2 + 1 < io.nsel;  // workgroup-uniform
    const uint32_t jta = nth_set_bit(io.target_mask, jg * 2), jtb = two ? nth_set_bit(io.target_mask, jg * 2 + 1) : jta;
    uint32_t item, tile;
    conv_item_tile(grp, groups, io.items, tiles, item, tile);
    const uint32_t ida = cv.dst_id[jta], idb = cv.dst_id[jtb];
    const LimbConst la = T.limb[ida], lb = T.limb[idb];
    if ((la.fp != 0) != (AR == AR_FP) || (lb.fp != 0) != (AR == AR_FP)) return;  // never: the host pairs targets of one class
    const int c = threadIdx.x % S, j = threadIdx.x / S;
    const u64 *src = io.in + (size_t)item * io.in_stride + tile * S + c;
    u64 xa[H], xb[H];
    if constexpr (N_IN > 4) {
        // 5..8 sources (e.g. alpha = K = 7 at L = 20): 30-bit columns with the middle one split in two, as in k_conv_col
        uint32_t h0a[N_IN], h1a[N_IN], h0b[N_IN], h1b[N_IN];
#pragma unroll
        for (int i = 0; i < N_IN; ++i) {
            split30(cv.hat[i * cv.n_out + jta], h0a[i], h1a[i]);
            split30(cv.hat[i * cv.n_out + jtb], h0b[i], h1b[i]);
        }
#pragma unroll
        for (int k = 0; k < H; ++k) {
            Cols4 acca{0, 0, 0, 0}, accb{0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < N_IN; ++i) {
                const u64 p = src[(size_t)cv.src_slot[i] * n + (size_t)(j + H * k) * r2];
                mac_cols4(acca, (uint32_t)p, (uint32_t)(p >> 32), h0a[i], h1a[i]);
                mac_cols4(accb, (uint32_t)p, (uint32_t)(p >> 32), h0b[i], h1b[i]);
            }
            xa[k] = reduce_cols4(acca, la);
            xb[k] = reduce_cols4(accb, lb);
            if (AR == AR_FP) {
                xa[k] = dbits(fp_reduce((double)xa[k], la.qd, la.qinv));
                xb[k] = dbits(fp_reduce((double)xb[k], lb.qd, lb.qinv));
            }
            asm volatile("" : "+v"(xa[k]), "+v"(xb[k]));
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
    ConvConst<N_IN> ka, kb;
    conv_consts<N_IN, AR, SRCMODE>(cv, jta, ka);
    conv_consts<N_IN, AR, SRCMODE>(cv, jtb, kb);
    constexpr int DEPTH = MK_CONV2_DEPTH < H ? MK_CONV2_DEPTH : H;  // 4 slices ahead (the one-target kernel: 6)
    u64 ring[DEPTH][N_IN];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k)
#pragma unroll
        for (int i = 0; i < N_IN; ++i) ring[k][i] = src[(size_t)cv.src_slot[i] * n + (size_t)(j + H * k) * r2];
#pragma unroll
    for (int k = 0; k < H; ++k) {
        u64 p[N_IN];
#pragma unroll
        for (int i = 0; i < N_IN; ++i) p[i] = ring[k % DEPTH][i];
        if (k + DEPTH < H) {
#pragma unroll
            for (int i = 0; i < N_IN; ++i)
                ring[k % DEPTH][i] = src[(size_t)cv.src_slot[i] * n + (size_t)(j + H * (k + DEPTH)) * r2];
        }
        uint32_t a0[N_IN], a1[N_IN];
        if (!((AR == AR_FP) && SRCMODE != 0)) conv_split_sources<N_IN, SRCMODE>(p, a0, a1);  // once for both targets
        xa[k] = conv_output<N_IN, AR, SRCMODE>(p, a0, a1, ka, la);
        xb[k] = conv_output<N_IN, AR, SRCMODE>(p, a0, a1, kb, lb);
        // materialised here (see k_conv_col): this is what makes two targets fit -- 115-117 VGPRs and no scratch, 4 waves
        // per SIMD like the one-target kernel; without it the compiler parked 34-150 registers in scratch
        asm volatile("" : "+v"(xa[k]), "+v"(xb[k]));
        __builtin_amdgcn_sched_barrier(0);
    }
    }
    u64 *dst = io.out + (size_t)item * io.out_stride + (size_t)cv.dst_slot[jta] * n + tile * S + c;
    col_forward_finish<LOG_H, AR>(xa, lds, T.tw + (size_t)ida * n, T.tw_sh + (size_t)ida * n, la, j, c, dst, r2);
    if (!two) return;
    __syncthreads();  // the first target's exchange is read out
    dst = io.out + (size_t)item * io.out_stride + (size_t)cv.dst_slot[jtb] * n + tile * S + c;
    // the second pass's twiddle loads must not be hoisted above the first pass (60 more live registers): its table
    // pointers are opaque until here
    const u64 *twb = T.tw + (size_t)idb * n, *twb_sh = T.tw_sh + (size_t)idb * n;
    asm volatile("" : "+s"(twb), "+s"(twb_sh));
    col_forward_finish<LOG_H, AR>(xb, lds, twb, twb_sh, lb, j, c, dst, r2);
}

// ---- ApproxModDown's conversion P -> Q_l for a whole group of clients at once ------------------------------------------
// The reference converts every client's key-switch result on its own (ApproxModDown inside each ReEncrypt) and adds the
// re-encryptions afterwards (EvalAdd).  With x_{c,k} the canonical residue mod p_k of client c's coefficient (the inverse
// transform of its P limb k, scaled by N^-1 [(P/p_k)^-1]_{p_k}), the conversion into the target q_t is
//     conv_c[t] = ( sum_k x_{c,k} [P/p_k]_t ) mod q_t ,
// and what the aggregate needs is  sum_c conv_c[t] mod q_t = ( sum_k X_k [P/p_k]_t ) mod q_t  with  X_k = sum_c x_{c,k}
// taken as an INTEGER (no reduction mod p_k: the approximate conversion is not additive in residues mod p_k, it is
// additive in their integer representatives).  n <= 15 residues below 2^60 fit a 64-bit word, so the group's clients
// are summed where their coefficients are produced (k_icol_sum, inside the inverse column pass) and converted ONCE
// (k_conv_col_psum): n - 1 of the n conversions per (index, component, target limb) and n - 1 of the n writes of the
// P-limb coefficients disappear, and the result is the same residue mod q_t, bit for bit.
//
// k_icol_sum: inverse COLUMN pass of the P limbs (second pass of ApproxModDown's SetFormat(COEFFICIENT)) of every client
// of the group, scaled to canonical residues and summed over the clients as 64-bit integers.
//   pc:   [client][poly][K][N]  outputs of the inverse row pass (lazy), clients in_cstride words apart
//   psum: [poly][K][N]          X_k = sum over clients, < n_clients * p_k < 2^64
// grid (column tile, poly * K + k); the host caps the group so that the sum cannot wrap.
template <int LOG_H, int AR>
__global__ __launch_bounds__(NTT_THREADS, 3) void k_icol_sum(const u64 *pc, u64 *psum, NttTables T, const u64 *scale,
                                                             const u64 *scale_sh, uint32_t K, uint32_t n_clients,
                                                             size_t in_cstride) {
    using TL = ColTile<LOG_H>;
    constexpr int H = TL::H, S = TL::S;
    static_assert(AR != AR_FP, "P limbs are integer-class");
    __shared__ u64 lds[TL::WORDS + ColTwB<LOG_H>::WORDS];
    const uint32_t poly = blockIdx.y / K, k = blockIdx.y % K;
    const uint32_t id = T.L + k;
    const LimbConst lc = T.limb[id];
    const uint32_t n = 1u << T.log_n, r2 = 1u << T.log_r2;
    const int c = threadIdx.x % S, j = threadIdx.x / S;
    const size_t off = ((size_t)poly * K + k) * n + blockIdx.x * S + c;
    const u64 *itw0 = T.itw + (size_t)id * n, *itw_sh0 = T.itw_sh + (size_t)id * n;
    const u64 sc = scale[id], sc_sh = scale_sh[id];
    u64 acc[H];
#pragma unroll
    for (int kk = 0; kk < H; ++kk) acc[kk] = 0;
    ColTwB<LOG_H>::stage(lds + TL::WORDS, itw0, itw_sh0);  // both rounds' twiddles: the same for every client, staged once
    wave_lds_sync();
    const u64 *strip = lds + TL::WORDS;
#pragma unroll 1
    for (uint32_t cl = 0; cl < n_clients; ++cl) {
        const u64 *src = pc + (size_t)cl * in_cstride + off;
        u64 x[H];
        // nothing but the sums is meant to live across the client loop: left alone the compiler computes the H load
        // offsets and loads the 2 (H - 1) round-B twiddles (they do not depend on the client) once in front of the loop
        // and keeps ~90 registers alive across it (52 spilled at 3 waves per SIMD).  Values that are opaque per
        // iteration are recomputed / re-read where they are used (L1 / L2 hits).
        uint32_t r2v = r2;
        asm volatile("" : "+s"(r2v));
#pragma unroll
        for (int kk = 0; kk < H; ++kk) x[kk] = ld_pass(src + (uint32_t)(H * j + kk) * r2v);
        radix_inverse_staged<LOG_H, AR>(x, [&](int i) { return ColTwB<LOG_H>::pair_b(strip, j, i); }, lc);
        if (cl) __syncthreads();  // the previous client's exchange is read out
#pragma unroll
        for (int kk = 0; kk < H; ++kk) lds[TL::at(j, kk, c)] = x[kk];
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < H; ++kk) x[kk] = lds[TL::at(kk, j, c)];
        radix_inverse_staged<LOG_H, AR>(x, [&](int i) { return ColTwB<LOG_H>::pair_a(strip, i); }, lc);
#pragma unroll
        for (int kk = 0; kk < H; ++kk) acc[kk] += shoup_mul(x[kk], sc, sc_sh, lc.q);  // canonical, as the per-client pass stores it
    }
    u64 *dst = psum + off;
#pragma unroll
    for (int kk = 0; kk < H; ++kk) st_pass(dst + (size_t)(j + H * kk) * r2, acc[kk]);
}

// k_conv_col_psum: the ONE conversion of a group, sum_k X_k [P/p_k]_t mod q_t, fused into the forward column pass of the
// target limb (k_conv_col's structure; grid over (poly, column tile, target), the targets of a source tile neighbours in
// one XCD's queue).  Sources: the K integer sums X_k (any 64-bit word); each is first reduced mod q_t -- an fp64-class
// target splits it into 32-bit halves, (X_hi [2^32]_t + X_lo) [P/p_k]_t as two exact FMA products; an integer-class
// target (q_0) takes one Barrett step and the 30-bit column accumulation of k_conv_col.
template <int LOG_H, int N_IN, int AR, typename CONV>
__global__ __launch_bounds__(NTT_THREADS, 4) void k_conv_col_psum(ConvIo io, NttTables T, CONV cv) {
    using TL = ColTile<LOG_H>;
    constexpr int H = TL::H, S = TL::S;
    __shared__ u64 lds[TL::WORDS + ColTwB<LOG_H>::WORDS];
    const uint32_t n = 1u << T.log_n, r2 = 1u << T.log_r2, tiles = r2 / S;
    const uint32_t groups = io.items * tiles;
    uint32_t grp, jt;
    group_member(blockIdx.x, groups, io.nsel, T.cu_affine, grp, jt);
    jt = nth_set_bit(io.target_mask, jt);
    uint32_t item, tile;
    conv_item_tile(grp, groups, io.items, tiles, item, tile);
    const uint32_t id = cv.dst_id[jt];
    const LimbConst lc = T.limb[id];
    if ((lc.fp != 0) != (AR == AR_FP)) return;  // never: the host selects the targets of this instance's class
    const int c = threadIdx.x % S, j = threadIdx.x / S;
    const u64 *src = io.in + (size_t)item * io.in_stride + tile * S + c;
    u64 *dst = io.out + (size_t)item * io.out_stride + (size_t)cv.dst_slot[jt] * n + tile * S + c;
    u64 x[H];
    if (AR == AR_FP) {
        double hd[N_IN], hq[N_IN];
#pragma unroll
        for (int i = 0; i < N_IN; ++i) {
            hd[i] = cv.hat_d[i * cv.n_out + jt];
            hq[i] = cv.hatq_d[i * cv.n_out + jt];
        }
        // 2^32 mod q_t; its quotient by q_t only steers fp_mulmod's choice of representative (the product is exact
        // for any integer quotient near the true one), so the double rounding of c32 * qinv is harmless
        const double c32 = (double)reduce_word(1ull << 32, lc), c32q = c32 * lc.qinv;
#pragma unroll
        for (int kk = 0; kk < H; ++kk) {
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < N_IN; ++i) {
                const u64 p = src[(size_t)cv.src_slot[i] * n + (size_t)(j + H * kk) * r2];
                // X = hi 2^32 + lo:  r = hi [2^32]_t + lo,  |r| <= 0.51 q + 2^32 < 2^52
                const double r = fp_mulmod((double)(uint32_t)(p >> 32), c32, c32q, lc.qd) + (double)(uint32_t)p;
                acc += fp_mulmod(r, hd[i], hq[i], lc.qd);  // |.| <= 0.76 q each: < 2^53 for 8 sources
            }
            x[kk] = dbits(fp_reduce(acc, lc.qd, lc.qinv));
            asm volatile("" : "+v"(x[kk]));  // materialised here (see k_conv_col)
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        uint32_t h0[N_IN], h1[N_IN];
#pragma unroll
        for (int i = 0; i < N_IN; ++i) split30(cv.hat[i * cv.n_out + jt], h0[i], h1[i]);
#pragma unroll
        for (int kk = 0; kk < H; ++kk) {
            if (N_IN <= 4) {
                Cols acc{0, 0, 0};
#pragma unroll
                for (int i = 0; i < N_IN; ++i) {
                    const u64 r = reduce_word(src[(size_t)cv.src_slot[i] * n + (size_t)(j + H * kk) * r2], lc);
                    uint32_t a0, a1;
                    split30(r, a0, a1);
                    mac_cols(acc, a0, a1, h0[i], h1[i]);
                }
                x[kk] = AR == AR_PM ? pm_reduce_cols(acc, pm_consts(lc)) : reduce_cols_lazy(acc, lc);  // < 4q
                asm volatile("" : "+v"(x[kk]));
                __builtin_amdgcn_sched_barrier(0);
            } else {
                Cols4 acc{0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < N_IN; ++i) {
                    const u64 r = reduce_word(src[(size_t)cv.src_slot[i] * n + (size_t)(j + H * kk) * r2], lc);
                    uint32_t a0, a1;
                    split30(r, a0, a1);
                    mac_cols4(acc, a0, a1, h0[i], h1[i]);
                }
                x[kk] = reduce_cols4(acc, lc);
                asm volatile("" : "+v"(x[kk]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    col_forward_finish<LOG_H, AR>(x, lds, T.tw + (size_t)id * n, T.tw_sh + (size_t)id * n, lc, j, c, dst, r2);
}

// k_conv_col_psum for TWO fp64-class targets per workgroup (see k_conv_col2: one pass over the K source tiles feeds both
// conversions; the last workgroup of an odd target count carries one).  This is the instance the fp64-class targets run;
// k_conv_col_psum itself serves the integer-class ones (q_0: a single target).
template <int LOG_H, int N_IN, typename CONV>
__global__ __launch_bounds__(NTT_THREADS, 3) void k_conv_col_psum2(ConvIo io, NttTables T, CONV cv) {
    using TL = ColTile<LOG_H>;
    constexpr int H = TL::H, S = TL::S;
    __shared__ u64 lds[TL::WORDS + ColTwB<LOG_H>::WORDS];
    const uint32_t n = 1u << T.log_n, r2 = 1u << T.log_r2, tiles = r2 / S;
    const uint32_t groups = io.items * tiles, ntg = (io.nsel + 1) / 2;
    uint32_t grp, jg;
    group_member(blockIdx.x, groups, ntg, T.cu_affine, grp, jg);
    const bool two = jg * 2 + 1 < io.nsel;  // workgroup-uniform
    const uint32_t jta = nth_set_bit(io.target_mask, jg * 2), jtb = two ? nth_set_bit(io.target_mask, jg * 2 + 1) : jta;
    uint32_t item, tile;
    conv_item_tile(grp, groups, io.items, tiles, item, tile);
    const uint32_t ida = cv.dst_id[jta], idb = cv.dst_id[jtb];
    const LimbConst la = T.limb[ida], lb = T.limb[idb];
    if (la.fp == 0 || lb.fp == 0) return;  // never: the host pairs fp64-class targets
    const int c = threadIdx.x % S, j = threadIdx.x / S;
    const u64 *src = io.in + (size_t)item * io.in_stride + tile * S + c;
    double hda[N_IN], hqa[N_IN], hdb[N_IN], hqb[N_IN];
#pragma unroll
    for (int i = 0; i < N_IN; ++i) {
        hda[i] = cv.hat_d[i * cv.n_out + jta];
        hqa[i] = cv.hatq_d[i * cv.n_out + jta];
        hdb[i] = cv.hat_d[i * cv.n_out + jtb];
        hqb[i] = cv.hatq_d[i * cv.n_out + jtb];
    }
    const double c32a = (double)reduce_word(1ull << 32, la), c32qa = c32a * la.qinv;  // see k_conv_col_psum
    const double c32b = (double)reduce_word(1ull << 32, lb), c32qb = c32b * lb.qinv;
    u64 xa[H], xb[H];
#pragma unroll
    for (int kk = 0; kk < H; ++kk) {
        double acca = 0.0, accb = 0.0;
#pragma unroll
        for (int i = 0; i < N_IN; ++i) {
            const u64 p = src[(size_t)cv.src_slot[i] * n + (size_t)(j + H * kk) * r2];
            const double hi = (double)(uint32_t)(p >> 32), lo = (double)(uint32_t)p;
            acca += fp_mulmod(fp_mulmod(hi, c32a, c32qa, la.qd) + lo, hda[i], hqa[i], la.qd);
            accb += fp_mulmod(fp_mulmod(hi, c32b, c32qb, lb.qd) + lo, hdb[i], hqb[i], lb.qd);
        }
        xa[kk] = dbits(fp_reduce(acca, la.qd, la.qinv));
        xb[kk] = dbits(fp_reduce(accb, lb.qd, lb.qinv));
        asm volatile("" : "+v"(xa[kk]), "+v"(xb[kk]));  // materialised here (see k_conv_col)
        __builtin_amdgcn_sched_barrier(0);
    }
    u64 *dst = io.out + (size_t)item * io.out_stride + (size_t)cv.dst_slot[jta] * n + tile * S + c;
    col_forward_finish<LOG_H, AR_FP>(xa, lds, T.tw + (size_t)ida * n, T.tw_sh + (size_t)ida * n, la, j, c, dst, r2);
    if (!two) return;
    __syncthreads();  // the first target's exchange is read out
    dst = io.out + (size_t)item * io.out_stride + (size_t)cv.dst_slot[jtb] * n + tile * S + c;
    const u64 *twb = T.tw + (size_t)idb * n, *twb_sh = T.tw_sh + (size_t)idb * n;
    asm volatile("" : "+s"(twb), "+s"(twb_sh));  // not hoisted above the first pass
    col_forward_finish<LOG_H, AR_FP>(xb, lds, twb, twb_sh, lb, j, c, dst, r2);
}

// DropLastElementAndScale, first half fused: NativeVectorT::SwitchModulus of the dropped limb (COEFFICIENT format,
// canonical; centred lift: v > floor(q_last / 2) is negative) into a remaining limb + the forward column pass of that
// limb -- the switched polynomial never goes to HBM in coefficient form.  last: [items][N]; out: [items][nl-1][N]
// column-passed; grid (column tile, target limb, item).  The row pass + (c - tmp) * q_last^-1 tail follow in k_ntt_row_r.
template <int LOG_H, int AR>
__global__ __launch_bounds__(NTT_THREADS) void k_switch_col(const u64 *last, u64 *out, NttTables T, uint32_t n_targets,
                                                            u64 q_last, unsigned long long target_mask) {
    using TL = ColTile<LOG_H>;
    constexpr int H = TL::H, S = TL::S;
    __shared__ u64 lds[TL::WORDS + ColTwB<LOG_H>::WORDS];
    const uint32_t n = 1u << T.log_n, r2 = 1u << T.log_r2;
    const uint32_t sl = nth_set_bit(target_mask, blockIdx.y), item = blockIdx.z;  // remaining Q limb: slot == limb id
    const LimbConst lc = T.limb[sl];
    if ((lc.fp != 0) != (AR == AR_FP)) return;  // never: the host selects the targets of this instance's class
    const int c = threadIdx.x % S, j = threadIdx.x / S;
    const u64 *src = last + (size_t)item * n + blockIdx.x * S + c;
    u64 *dst = out + ((size_t)item * n_targets + sl) * n + blockIdx.x * S + c;
    const u64 half = q_last >> 1, ql_mod = reduce_word(q_last, lc);
    u64 x[H];
#pragma unroll
    for (int k = 0; k < H; ++k) {
        const u64 v = ld_pass(src + (size_t)(j + H * k) * r2);
        const u64 a = reduce_word(v, lc);
        const u64 r = v > half ? sub_mod(a, ql_mod, lc.q) : a;
        x[k] = (AR == AR_FP) ? dbits(u52_to_double(r)) : r;  // canonical: inside the first round's range for both classes
    }
    col_forward_finish<LOG_H, AR>(x, lds, T.tw + (size_t)sl * n, T.tw_sh + (size_t)sl * n, lc, j, c, dst, r2);
}

// ApproxModDown tail folded into the copy-out of the forward row pass:
//   out = (ctilde_Q - conv) * P^-1  (+ c0 on component 0)
struct TailArgs {
    const u64 *til;      // [polys][ext][N]  key-switch accumulators over Q_l P (component-major per ciphertext)
    const u64 *add;      // ciphertexts [..][2][nl][N]: c0 is added on even polys; may be null
    const u64 *pinv, *pinv_sh;  // [nl]
    size_t add_stride;   // words between ciphertexts in `add`
    uint32_t ext;        // nl + K
    uint32_t enabled;
    uint32_t accumulate; // out += result (running aggregate over clients) instead of out = result
};

// Row pass over rows of R2 = H*H contiguous words: one workgroup = S consecutive rows (S*R2 contiguous
// words).  The side that needs per-thread contiguous runs goes through LDS with coalesced 16-B accesses.
template <int LOG_H, bool INV, int AR>
__global__ __launch_bounds__(NTT_THREADS) void k_ntt_row_r(NttIo io, NttTables T, TailArgs tail) {
    using TL = RowTile<LOG_H>;
    using TA = RowTwA<LOG_H>;
    constexpr int H = TL::H, S = TL::S, R = TL::R, PAIRS = S * R / 2 / NTT_THREADS;
    __shared__ u64 lds[TL::WORDS + 2 * TA::WORDS];
    u64 *twa = lds + TL::WORDS, *twa_sh = twa + TA::WORDS;
    // 1-D grid over (limb slot, row tile, polynomial).  All polynomials of one (slot, tile) read the same
    // 2*S*R-word twiddle tile: they are made consecutive inside ONE XCD's queue (blocks b, b+8, ... share an
    // XCD under round-robin dispatch) so the tile is fetched over the fabric once and then hits in that L2.
    const uint32_t n = 1u << T.log_n, r1 = 1u << T.log_r1;
    const uint32_t tiles = r1 / S, groups = tiles * io.nsel, n_polys = gridDim.x / groups;
    uint32_t grp, poly;
    group_member(blockIdx.x, groups, n_polys, T.cu_affine, grp, poly);
    const uint32_t sl = nth_set_bit(io.slot_mask, grp / tiles);
    if (ntt_slot_skipped(io, poly, io.vslot0 + sl)) return;  // block-uniform
    const uint32_t id = limb_id_of(io.vslot0 + sl, io.nl, T.L);
    const LimbConst lc = T.limb[id];
    if ((lc.fp != 0) != (AR == AR_FP)) return;  // block-uniform
    const uint32_t row0 = (grp % tiles) * S;
    const int g = threadIdx.x / H, j = threadIdx.x % H;
    const u64 *src = io.in + ntt_in_offset(io, poly) + (size_t)(io.in_slot0 + sl) * n + (size_t)row0 * R;
    u64 *dst = io.out + (size_t)poly * io.out_stride + (size_t)(io.out_slot0 + sl) * n + (size_t)row0 * R;
    const u64 *tw = (INV ? T.itw : T.tw) + (size_t)id * n;
    const u64 *tw_sh = (INV ? T.itw_sh : T.tw_sh) + (size_t)id * n;
    const u64 *twb = (INV ? T.itwb : T.twb) + (size_t)id * 2 * n;
    u64 x[H], w[H - 1], wp[H - 1];
    if (!INV) {
#pragma unroll
        for (int k = 0; k < H; ++k) x[k] = ld_pass(src + (size_t)g * R + j + H * k);
        stage_twiddles_wave<LOG_H>(twa, twa_sh, tw, tw_sh, r1 + row0);
        wave_lds_sync();
        TA::fetch(twa, twa_sh, g, w, wp);
        radix_forward_any<LOG_H, AR>(x, w, wp, lc);
        u64 w2[H - 1], wp2[H - 1];  // round-B twiddles: requested before the exchange, used after it
        load_rowb_twiddles<LOG_H>(twb, row0 + g, j, w2, wp2);
#pragma unroll
        for (int k = 0; k < H; ++k) lds[TL::at(g, j + H * k)] = x[k];
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < H; ++k) x[k] = lds[TL::at(g, H * j + k)];
        radix_forward_any<LOG_H, AR>(x, w2, wp2, lc);
#pragma unroll
        for (int k = 0; k < H; ++k)  // canonical u64, own words only
            lds[TL::at(g, H * j + k)] = (AR == AR_FP) ? fp_to_canonical(bitsd(x[k]), lc.qd, lc.qinv) : canon8(x[k], lc.q, lc.q2);
        wave_lds_sync();
        if (!tail.enabled) {
            for (int i = 0; i < PAIRS; ++i) {
                const int e = wave_pair<LOG_H>(i);
                const int gg = (2 * e) / R, xx = (2 * e) % R;
                ulong2 v;
                v.x = lds[TL::at(gg, xx)];
                v.y = lds[TL::at(gg, xx + 1)];
                st_pass2(reinterpret_cast<ulong2 *>(dst) + e, v);
            }
        } else {
            const u64 pi = tail.pinv[sl], pi_sh = tail.pinv_sh[sl];
            const u64 *tq = tail.til + ((size_t)poly * tail.ext + sl) * n + (size_t)row0 * R;
            const u64 *c0 = (tail.add && (poly & 1) == 0)
                                ? tail.add + (size_t)(poly >> 1) * tail.add_stride + (size_t)sl * n + (size_t)row0 * R
                                : nullptr;
            for (int i = 0; i < PAIRS; ++i) {
                const int e = wave_pair<LOG_H>(i);
                const int gg = (2 * e) / R, xx = (2 * e) % R;
                const ulong2 t = ld_pass2(reinterpret_cast<const ulong2 *>(tq) + e);
                ulong2 v;
                v.x = shoup_mul(sub_mod(t.x, lds[TL::at(gg, xx)], lc.q), pi, pi_sh, lc.q);
                v.y = shoup_mul(sub_mod(t.y, lds[TL::at(gg, xx + 1)], lc.q), pi, pi_sh, lc.q);
                if (c0) {
                    const ulong2 a = reinterpret_cast<const ulong2 *>(c0)[e];
                    v.x = add_mod(v.x, a.x, lc.q);
                    v.y = add_mod(v.y, a.y, lc.q);
                }
                if (tail.accumulate) {
                    const ulong2 a = reinterpret_cast<const ulong2 *>(dst)[e];
                    v.x = add_mod(v.x, a.x, lc.q);
                    v.y = add_mod(v.y, a.y, lc.q);
                }
                st_pass2(reinterpret_cast<ulong2 *>(dst) + e, v);
            }
        }
    } else {
        load_rowb_twiddles<LOG_H>(twb, row0 + g, j, w, wp);  // first: in flight while the tile is staged
        for (int i = 0; i < PAIRS; ++i) {
                const int e = wave_pair<LOG_H>(i);
            const int gg = (2 * e) / R, xx = (2 * e) % R;
            const ulong2 v = ld_pass2(reinterpret_cast<const ulong2 *>(src) + e);
            lds[TL::at(gg, xx)] = v.x;
            lds[TL::at(gg, xx + 1)] = v.y;
        }
        stage_twiddles_wave<LOG_H>(twa, twa_sh, tw, tw_sh, r1 + row0);  // for the second (broadcast) round
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < H; ++k) x[k] = lds[TL::at(g, H * j + k)];
        if (AR == AR_FP) {  // canonical input -> doubles
#pragma unroll
            for (int k = 0; k < H; ++k) x[k] = dbits((double)x[k]);
        }
        radix_inverse_any<LOG_H, AR>(x, w, wp, lc);
#pragma unroll
        for (int k = 0; k < H; ++k) lds[TL::at(g, H * j + k)] = x[k];
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < H; ++k) x[k] = lds[TL::at(g, j + H * k)];
        TA::fetch(twa, twa_sh, g, w, wp);
        radix_inverse_any<LOG_H, AR>(x, w, wp, lc);
#pragma unroll
        for (int k = 0; k < H; ++k) st_pass(dst + (size_t)g * R + j + H * k, x[k]);  // lazy [0,2q) (doubles on an fp limb): the column pass scales
    }
}

// ModUp row pass + inner product with the eval key in ONE kernel, fp64 limbs only (EvalKeySwitchPrecomputeCore's
// SetFormat(EVALUATION) + EvalFastKeySwitchCoreExt, keyswitch-hybrid.cpp): for a (ciphertext, Q limb t, row tile) the
// workgroup finishes the forward transform of every converted digit d_j[t] and accumulates d_j[t] * b_j[t] and
// d_j[t] * a_j[t] in registers (exact doubles, fp_mulmod_any); the digit that owns t comes straight from c1.  The
// transformed digits never go to HBM: per ciphertext and limb this saves (beta-1) limb writes + beta limb reads.
struct InnerArgs {
    const u64 *dig;      // [item][nparts][ext][N] column-passed converted limbs (doubles on fp limbs)
    const u64 *c1;       // component 1 of the input ciphertexts, items c1_stride words apart: [nl][N] canonical
    const u64 *evk;      // [nparts][2][D][N]
    u64 *til;            // [item][2][ext][N]
    size_t c1_stride;
    uint32_t nl, ext, D, alpha, items;
    unsigned long long slot_mask;  // fp-class Q limbs
    uint32_t nsel;
    // n-client flow: item = client * ipc + index; c1 of an item at c1 + client * c1_gstride + index * c1_stride, the
    // client's key at evk + client * evk_cstride (ipc >= items: one client)
    uint32_t ipc = 0xFFFFFFFFu;
    size_t c1_gstride = 0, evk_cstride = 0;
    // til with only this launch's slots: [item][2][nsel][N], slot index = rank in slot_mask
    uint32_t til_compact = 0;
};
MK_D const u64 *inner_c1(const InnerArgs &a, uint32_t item) {
    return a.c1 + (size_t)(item / a.ipc) * a.c1_gstride + (size_t)(item % a.ipc) * a.c1_stride;
}
MK_D const u64 *inner_evk(const InnerArgs &a, uint32_t item) { return a.evk + (size_t)(item / a.ipc) * a.evk_cstride; }
template <int LOG_H, int NPARTS, int WAVES>
__global__ __launch_bounds__(NTT_THREADS, WAVES) void k_row_inner_fp(InnerArgs a, NttTables T) {
    using TL = RowTile<LOG_H>;
    using TA = RowTwA<LOG_H>;
    constexpr int H = TL::H, S = TL::S, R = TL::R, PAIRS = S * R / 2 / NTT_THREADS;
    __shared__ u64 lds[TL::WORDS + 2 * TA::WORDS];
    u64 *twa = lds + TL::WORDS, *twa_sh = twa + TA::WORDS;
    const uint32_t n = 1u << T.log_n, r1 = 1u << T.log_r1;
    const uint32_t tiles = r1 / S, groups = tiles * a.nsel;
    uint32_t grp, item;
    group_member(blockIdx.x, groups, a.items, T.cu_affine, grp, item);
    const uint32_t sl = nth_set_bit(a.slot_mask, grp / tiles);  // Q limb: slot == limb id
    const LimbConst lc = T.limb[sl];
    const int own = (int)(sl / a.alpha);
    const uint32_t row0 = (grp % tiles) * S;
    const int g = threadIdx.x / H, j = threadIdx.x % H;
    const u64 *tw = T.tw + (size_t)sl * n, *tw_sh = T.tw_sh + (size_t)sl * n;
    const u64 *twb = T.twb + (size_t)sl * 2 * n;
    const size_t tile_off = (size_t)row0 * R;
    const double q = lc.qd, qinv = lc.qinv;
    stage_twiddles_wave<LOG_H>(twa, twa_sh, tw, tw_sh, r1 + row0);  // round-A twiddles are the same for every digit
    int jn = own == 0 ? 1 : 0;  // first converted digit
    const u64 *dig0 = a.dig + ((size_t)item * NPARTS * a.ext + sl) * n + tile_off + (size_t)g * R + j;
    u64 x[H];
    if (jn < NPARTS) {
        const u64 *src = dig0 + (size_t)jn * a.ext * n;
#pragma unroll
        for (int k = 0; k < H; ++k) x[k] = ld_stream(src + H * k);
    }
    double2 acc0[PAIRS], acc1[PAIRS];
    {   // the digit that owns this limb: c1 itself, already in EVALUATION format
        const u64 *y0 = a.c1 + (size_t)item * a.c1_stride + (size_t)sl * n + tile_off;
        const u64 *e0 = a.evk + (((size_t)own * 2 + 0) * a.D + sl) * n + tile_off;
        const u64 *e1 = a.evk + (((size_t)own * 2 + 1) * a.D + sl) * n + tile_off;
#pragma unroll
        for (int i = 0; i < PAIRS; ++i) {
            const int e = wave_pair<LOG_H>(i);
            const ulong2 yy = ld_stream2(reinterpret_cast<const ulong2 *>(y0) + e);
            const ulong2 b = reinterpret_cast<const ulong2 *>(e0)[e];
            const ulong2 c = reinterpret_cast<const ulong2 *>(e1)[e];
            const double yx = u52_to_double(yy.x), yz = u52_to_double(yy.y);
            acc0[i].x = fp_mulmod_any(yx, u52_to_double(b.x), q, qinv);
            acc0[i].y = fp_mulmod_any(yz, u52_to_double(b.y), q, qinv);
            acc1[i].x = fp_mulmod_any(yx, u52_to_double(c.x), q, qinv);
            acc1[i].y = fp_mulmod_any(yz, u52_to_double(c.y), q, qinv);
        }
    }
#pragma unroll 1
    for (int dj = jn; dj < NPARTS; dj = jn) {
        jn = dj + 1 == own ? dj + 2 : dj + 1;  // next converted digit
        {
            u64 w[H - 1], wp[H - 1];
            wave_lds_sync();  // twiddles staged (first digit) / previous digit's products finished reading LDS
            TA::fetch(twa, twa_sh, g, w, wp);
            radix_forward_fp<LOG_H>(x, w, wp, q, qinv);
        }
        u64 w2[H - 1], wp2[H - 1];  // round-B twiddles: requested before the exchange, used after it
        load_rowb_twiddles<LOG_H>(twb, row0 + g, j, w2, wp2);
#pragma unroll
        for (int k = 0; k < H; ++k) lds[TL::at(g, j + H * k)] = x[k];
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < H; ++k) x[k] = lds[TL::at(g, H * j + k)];
        radix_forward_fp<LOG_H>(x, w2, wp2, q, qinv);
#pragma unroll
        for (int k = 0; k < H; ++k) lds[TL::at(g, H * j + k)] = dbits(fp_reduce(bitsd(x[k]), q, qinv));  // |y| <= 0.51 q
        if (jn < NPARTS) {  // next digit's inputs are requested while this digit's products stream
            const u64 *src = dig0 + (size_t)jn * a.ext * n;
#pragma unroll
            for (int k = 0; k < H; ++k) x[k] = ld_stream(src + H * k);
        }
        wave_lds_sync();
        const u64 *e0 = a.evk + (((size_t)dj * 2 + 0) * a.D + sl) * n + tile_off;
        const u64 *e1 = a.evk + (((size_t)dj * 2 + 1) * a.D + sl) * n + tile_off;
#pragma unroll
        for (int i = 0; i < PAIRS; ++i) {
            const int e = wave_pair<LOG_H>(i);
            const int gg = (2 * e) / R, xx = (2 * e) % R;
            const ulong2 b = reinterpret_cast<const ulong2 *>(e0)[e];
            const ulong2 c = reinterpret_cast<const ulong2 *>(e1)[e];
            const double yx = bitsd(lds[TL::at(gg, xx)]), yz = bitsd(lds[TL::at(gg, xx + 1)]);
            acc0[i].x += fp_mulmod_any(yx, u52_to_double(b.x), q, qinv);
            acc0[i].y += fp_mulmod_any(yz, u52_to_double(b.y), q, qinv);
            acc1[i].x += fp_mulmod_any(yx, u52_to_double(c.x), q, qinv);
            acc1[i].y += fp_mulmod_any(yz, u52_to_double(c.y), q, qinv);
        }
    }
    // |acc| <= (0.97 + 0.75 (NPARTS-1)) q < 2^53: exact integers
    u64 *t0 = a.til + (((size_t)item * 2 + 0) * a.ext + sl) * n + tile_off;
    u64 *t1 = a.til + (((size_t)item * 2 + 1) * a.ext + sl) * n + tile_off;
#pragma unroll
    for (int i = 0; i < PAIRS; ++i) {
        const int e = wave_pair<LOG_H>(i);
        ulong2 r0, r1v;
        r0.x = fp_to_canonical(acc0[i].x, q, qinv);
        r0.y = fp_to_canonical(acc0[i].y, q, qinv);
        r1v.x = fp_to_canonical(acc1[i].x, q, qinv);
        r1v.y = fp_to_canonical(acc1[i].y, q, qinv);
        st_stream2(reinterpret_cast<ulong2 *>(t0) + e, r0);
        st_stream2(reinterpret_cast<ulong2 *>(t1) + e, r1v);
    }
}

// ---- 512-point rows (N = 2^17 = 256 x 512): three rounds of radix 8 ---------------------------------------
// Position x = 64a + 8b + c of a row; a thread is (p, r) with p, r < 8 and holds 8 words per round:
//   round A: (a,b,c) = (k,p,r)  stages 0-2  base_eff = base
//   round B: (a,b,c) = (p,k,r)  stages 3-5  base_eff = 8 base + p
//   round C: (a,b,c) = (p,r,k)  stages 6-8  base_eff = 64 base + 8p + r       (forward; the inverse runs C, B, A)
// A workgroup owns 4 consecutive rows (2048 contiguous words); two LDS exchanges; layout x + x/8 per row (stride
// 576) is conflict-free for all three access patterns.  Round A and B twiddles are shared by 64 resp. 8 threads and
// go through LDS (contiguous runs over the tile), round C twiddles are per thread.
// Generic three-round geometry: R = 8 * 8 * C words per row with C = 2^LOGC in {4, 8} (256- and 512-point rows),
// position x = 8C a + C b + c, TPR = R/8 threads per row (8 words each), 256/TPR rows per workgroup.  The last round
// is one radix-8 (C = 8) or two radix-4 groups (C = 4) on the thread's 8 CONTIGUOUS words 8u..8u+7.
template <int LOGC>
struct RowT {
    static constexpr int H = 8, C = 1 << LOGC, R = 64 * C, TPR = R / 8, ROWS = NTT_THREADS / TPR;
    static constexpr int RS = R + R / 8, WORDS = ROWS * RS;
    static constexpr int TWA = ROWS * 7, TWB = ROWS * 8 * 7;  // per table
    static MK_D int at(int g, int x) { return g * RS + x + (x >> 3); }
};
using Row3 = RowT<3>;
template <bool INV, int AR>
__global__ __launch_bounds__(NTT_THREADS) void k_ntt_row3(NttIo io, NttTables T, TailArgs tail) {
    using TL = Row3;
    constexpr int H = 8, LOG_H = 3, R = TL::R, S = TL::ROWS;
    __shared__ u64 lds[TL::WORDS + 2 * (TL::TWA + TL::TWB)];
    u64 *twa = lds + TL::WORDS, *twa_sh = twa + TL::TWA, *twb = twa_sh + TL::TWA, *twb_sh = twb + TL::TWB;
    const uint32_t n = 1u << T.log_n, r1 = 1u << T.log_r1;
    const uint32_t tiles = r1 / S, groups = tiles * io.nsel, n_polys = gridDim.x / groups;
    uint32_t grp, poly;
    group_member(blockIdx.x, groups, n_polys, T.cu_affine, grp, poly);
    const uint32_t sl = nth_set_bit(io.slot_mask, grp / tiles);
    if (ntt_slot_skipped(io, poly, io.vslot0 + sl)) return;
    const uint32_t id = limb_id_of(io.vslot0 + sl, io.nl, T.L);
    const LimbConst lc = T.limb[id];
    if ((lc.fp != 0) != (AR == AR_FP)) return;
    const uint32_t row0 = (grp % tiles) * S;
    const int g = threadIdx.x / 64, t = threadIdx.x % 64, p = t / 8, r = t % 8;
    const u64 *src = io.in + (size_t)poly * io.in_stride + (size_t)(io.in_slot0 + sl) * n + (size_t)row0 * R;
    u64 *dst = io.out + (size_t)poly * io.out_stride + (size_t)(io.out_slot0 + sl) * n + (size_t)row0 * R;
    const u64 *tw = (INV ? T.itw : T.tw) + (size_t)id * n;
    const u64 *tw_sh = (INV ? T.itw_sh : T.tw_sh) + (size_t)id * n;
    const uint32_t base = r1 + row0 + g;
    // stage the shared twiddles: stage-s entries of rounds A / B are contiguous runs over the 4 rows
    for (int e = threadIdx.x; e < TL::TWA; e += NTT_THREADS) {
        const int s = 31 - __clz(e / S + 1), off = e - S * ((1 << s) - 1);
        const uint32_t idx = ((r1 + row0) << s) + (uint32_t)off;
        twa[e] = tw[idx];
        twa_sh[e] = tw_sh[idx];
    }
    for (int e = threadIdx.x; e < TL::TWB; e += NTT_THREADS) {
        const int s = 31 - __clz(e / (S * 8) + 1), off = e - S * 8 * ((1 << s) - 1);
        const uint32_t idx = (((r1 + row0) * 8) << s) + (uint32_t)off;
        twb[e] = tw[idx];
        twb_sh[e] = tw_sh[idx];
    }
    u64 x[H], w[H - 1], wp[H - 1];
    auto fetch_a = [&]() {
#pragma unroll
        for (int s = 0; s < LOG_H; ++s)
#pragma unroll
            for (int gg = 0; gg < (1 << s); ++gg) {
                const int e = S * ((1 << s) - 1) + (g << s) + gg;
                w[(1 << s) - 1 + gg] = twa[e];
                wp[(1 << s) - 1 + gg] = twa_sh[e];
            }
    };
    auto fetch_b = [&]() {
#pragma unroll
        for (int s = 0; s < LOG_H; ++s)
#pragma unroll
            for (int gg = 0; gg < (1 << s); ++gg) {
                const int e = S * 8 * ((1 << s) - 1) + ((g * 8 + p) << s) + gg;
                w[(1 << s) - 1 + gg] = twb[e];
                wp[(1 << s) - 1 + gg] = twb_sh[e];
            }
    };
    if (!INV) {
#pragma unroll
        for (int k = 0; k < H; ++k) x[k] = src[(size_t)g * R + t + 64 * k];  // doubles from the column pass on an fp limb
        __syncthreads();
        fetch_a();
        radix_forward_any<LOG_H, AR>(x, w, wp, lc);
#pragma unroll
        for (int k = 0; k < H; ++k) lds[TL::at(g, t + 64 * k)] = x[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < H; ++k) x[k] = lds[TL::at(g, 64 * p + 8 * k + r)];
        fetch_b();
        radix_forward_any<LOG_H, AR>(x, w, wp, lc);
#pragma unroll
        for (int k = 0; k < H; ++k) lds[TL::at(g, 64 * p + 8 * k + r)] = x[k];  // own words
        __syncthreads();
#pragma unroll
        for (int k = 0; k < H; ++k) x[k] = lds[TL::at(g, 64 * p + 8 * r + k)];
        load_round_twiddles<LOG_H>(tw, tw_sh, base * 64 + 8 * p + r, w, wp);
        radix_forward_any<LOG_H, AR>(x, w, wp, lc);
#pragma unroll
        for (int k = 0; k < H; ++k)
            lds[TL::at(g, 64 * p + 8 * r + k)] = (AR == AR_FP) ? fp_to_canonical(bitsd(x[k]), lc.qd, lc.qinv) : canon8(x[k], lc.q, lc.q2);
        __syncthreads();
        if (!tail.enabled) {
            for (int e = threadIdx.x; e < S * R / 2; e += NTT_THREADS) {
                const int gg = (2 * e) / R, xx = (2 * e) % R;
                ulong2 v;
                v.x = lds[TL::at(gg, xx)];
                v.y = lds[TL::at(gg, xx + 1)];
                reinterpret_cast<ulong2 *>(dst)[e] = v;
            }
        } else {  // ApproxModDown tail in the copy-out (same as k_ntt_row_r)
            const u64 pi = tail.pinv[sl], pi_sh = tail.pinv_sh[sl];
            const u64 *tq = tail.til + ((size_t)poly * tail.ext + sl) * n + (size_t)row0 * R;
            const u64 *c0 = (tail.add && (poly & 1) == 0)
                                ? tail.add + (size_t)(poly >> 1) * tail.add_stride + (size_t)sl * n + (size_t)row0 * R
                                : nullptr;
            for (int e = threadIdx.x; e < S * R / 2; e += NTT_THREADS) {
                const int gg = (2 * e) / R, xx = (2 * e) % R;
                const ulong2 tt = reinterpret_cast<const ulong2 *>(tq)[e];
                ulong2 v;
                v.x = shoup_mul(sub_mod(tt.x, lds[TL::at(gg, xx)], lc.q), pi, pi_sh, lc.q);
                v.y = shoup_mul(sub_mod(tt.y, lds[TL::at(gg, xx + 1)], lc.q), pi, pi_sh, lc.q);
                if (c0) {
                    const ulong2 z = reinterpret_cast<const ulong2 *>(c0)[e];
                    v.x = add_mod(v.x, z.x, lc.q);
                    v.y = add_mod(v.y, z.y, lc.q);
                }
                if (tail.accumulate) {
                    const ulong2 z = reinterpret_cast<const ulong2 *>(dst)[e];
                    v.x = add_mod(v.x, z.x, lc.q);
                    v.y = add_mod(v.y, z.y, lc.q);
                }
                reinterpret_cast<ulong2 *>(dst)[e] = v;
            }
        }
    } else {
        for (int e = threadIdx.x; e < S * R / 2; e += NTT_THREADS) {
            const int gg = (2 * e) / R, xx = (2 * e) % R;
            const ulong2 v = reinterpret_cast<const ulong2 *>(src)[e];
            lds[TL::at(gg, xx)] = v.x;
            lds[TL::at(gg, xx + 1)] = v.y;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < H; ++k) x[k] = lds[TL::at(g, 64 * p + 8 * r + k)];
        if (AR == AR_FP) {
#pragma unroll
            for (int k = 0; k < H; ++k) x[k] = dbits((double)x[k]);
        }
        load_round_twiddles<LOG_H>(tw, tw_sh, base * 64 + 8 * p + r, w, wp);
        radix_inverse_any<LOG_H, AR>(x, w, wp, lc);
#pragma unroll
        for (int k = 0; k < H; ++k) lds[TL::at(g, 64 * p + 8 * r + k)] = x[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < H; ++k) x[k] = lds[TL::at(g, 64 * p + 8 * k + r)];
        fetch_b();
        radix_inverse_any<LOG_H, AR>(x, w, wp, lc);
#pragma unroll
        for (int k = 0; k < H; ++k) lds[TL::at(g, 64 * p + 8 * k + r)] = x[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < H; ++k) x[k] = lds[TL::at(g, t + 64 * k)];
        fetch_a();
        radix_inverse_any<LOG_H, AR>(x, w, wp, lc);
#pragma unroll
        for (int k = 0; k < H; ++k) dst[(size_t)g * R + t + 64 * k] = x[k];  // lazy [0,2q) / doubles: the column pass scales
    }
}

// ---- fused kernels on the three-round row geometry (512-point rows of N = 2^17, 256-point rows as 8 x 8 x 4) ----------
// A wavefront owns ONE row (64 threads x 8 words) in every phase, so after the cooperative twiddle staging all LDS
// hand-offs are wave-level.  8 words per thread keep the accumulators small (16-32 registers): these kernels run at
// high occupancy.
struct Row3Ctx {
    u64 *lds, *twa, *twa_sh, *twb, *twb_sh;
    int g, t;  // row of the tile, thread inside the row
    // round-C twiddles parked in LDS by kernels that transform many polynomials per workgroup and are short of registers
    // (row3_park_c_twiddles): pair i of thread tid at twc[i * NTT_THREADS + tid]
    ulong2 *twc = nullptr;
};
// park / fetch this thread's round-C twiddles (wave-private LDS words: a thread only reads what it wrote)
template <int LOGC>
MK_D void row3_park_c_twiddles(const Row3Ctx &c, const u64 (&wc)[7], const u64 (&wpc)[7]) {
#pragma unroll
    for (int i = 0; i < (LOGC == 3 ? 7 : 6); ++i) c.twc[i * NTT_THREADS + threadIdx.x] = ulong2{wc[i], wpc[i]};
}
template <int LOGC>
MK_D void row3_fetch_c_twiddles(const Row3Ctx &c, u64 (&wc)[7], u64 (&wpc)[7]) {
#pragma unroll
    for (int i = 0; i < (LOGC == 3 ? 7 : 6); ++i) {
        const ulong2 t = c.twc[i * NTT_THREADS + threadIdx.x];
        wc[i] = t.x;
        wpc[i] = t.y;
    }
    if (LOGC != 3) wc[6] = wpc[6] = 0;
}
// stage the round-A / round-B twiddles of the tile's 4 rows (cooperative; the caller synchronises the workgroup)
template <int LOGC>
MK_D void row3_stage_twiddles(const Row3Ctx &c, const u64 *tw, const u64 *tw_sh, uint32_t base0) {
    constexpr int S = RowT<LOGC>::ROWS;
    for (int e = threadIdx.x; e < RowT<LOGC>::TWA; e += NTT_THREADS) {
        const int s = 31 - __clz(e / S + 1), off = e - S * ((1 << s) - 1);
        const uint32_t idx = (base0 << s) + (uint32_t)off;
        c.twa[e] = tw[idx];
        c.twa_sh[e] = tw_sh[idx];
    }
    for (int e = threadIdx.x; e < RowT<LOGC>::TWB; e += NTT_THREADS) {
        const int s = 31 - __clz(e / (S * 8) + 1), off = e - S * 8 * ((1 << s) - 1);
        const uint32_t idx = ((base0 * 8) << s) + (uint32_t)off;
        c.twb[e] = tw[idx];
        c.twb_sh[e] = tw_sh[idx];
    }
}
// round-C twiddles of a thread (the same for every polynomial of the limb): one radix-8 set, or two radix-4 sets
template <int LOGC>
MK_D void row3_load_c_twiddles(const u64 *tw, const u64 *tw_sh, uint32_t base, int t, u64 (&wc)[7], u64 (&wpc)[7]) {
    if (LOGC == 3) {
        load_round_twiddles<3>(tw, tw_sh, base * 64 + (uint32_t)t, wc, wpc);
    } else {
        u64 a[3], ap[3], b[3], bp[3];
        load_round_twiddles<2>(tw, tw_sh, base * 64 + 2u * (uint32_t)t, a, ap);
        load_round_twiddles<2>(tw, tw_sh, base * 64 + 2u * (uint32_t)t + 1u, b, bp);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            wc[i] = a[i];
            wpc[i] = ap[i];
            wc[3 + i] = b[i];
            wpc[3 + i] = bp[i];
        }
        wc[6] = wpc[6] = 0;
    }
}
// forward transform of this thread's row: x[k] = word t + TPR k on entry; on exit x[k] = word 8 t + k in the lazy
// range of the arithmetic
template <int AR, int LOGC, bool C_PARKED = false>
MK_D void row3_forward(u64 (&x)[8], const Row3Ctx &c, const u64 (&wc_in)[7], const u64 (&wpc_in)[7], const LimbConst &lc) {
    using TL = RowT<LOGC>;
    constexpr int S = TL::ROWS, TPR = TL::TPR, C = TL::C;
    const int a = c.t / C, cc = c.t % C;
    u64 w[7], wp[7];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int gg = 0; gg < (1 << s); ++gg) {
            const int e = S * ((1 << s) - 1) + (c.g << s) + gg;
            w[(1 << s) - 1 + gg] = c.twa[e];
            wp[(1 << s) - 1 + gg] = c.twa_sh[e];
        }
    radix_forward_any<3, AR>(x, w, wp, lc);
#pragma unroll
    for (int k = 0; k < 8; ++k) c.lds[TL::at(c.g, c.t + TPR * k)] = x[k];
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = c.lds[TL::at(c.g, 8 * C * a + C * k + cc)];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int gg = 0; gg < (1 << s); ++gg) {
            const int e = S * 8 * ((1 << s) - 1) + ((c.g * 8 + a) << s) + gg;
            w[(1 << s) - 1 + gg] = c.twb[e];
            wp[(1 << s) - 1 + gg] = c.twb_sh[e];
        }
    radix_forward_any<3, AR>(x, w, wp, lc);
#pragma unroll
    for (int k = 0; k < 8; ++k) c.lds[TL::at(c.g, 8 * C * a + C * k + cc)] = x[k];  // own words
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = c.lds[TL::at(c.g, 8 * c.t + k)];
    u64 wc[7], wpc[7];
    if (C_PARKED) {
        row3_fetch_c_twiddles<LOGC>(c, wc, wpc);
    } else {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            wc[i] = wc_in[i];
            wpc[i] = wpc_in[i];
        }
    }
    if (LOGC == 3) {
        radix_forward_any<3, AR>(x, wc, wpc, lc);
    } else {  // two radix-4 groups: words 8t..8t+3 and 8t+4..8t+7
        u64 y0[4] = {x[0], x[1], x[2], x[3]}, y1[4] = {x[4], x[5], x[6], x[7]};
        const u64 w0[3] = {wc[0], wc[1], wc[2]}, wp0[3] = {wpc[0], wpc[1], wpc[2]};
        const u64 w1[3] = {wc[3], wc[4], wc[5]}, wp1[3] = {wpc[3], wpc[4], wpc[5]};
        radix_forward_any<2, AR>(y0, w0, wp0, lc);
        radix_forward_any<2, AR>(y1, w1, wp1, lc);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            x[k] = y0[k];
            x[4 + k] = y1[k];
        }
    }
}
// i-th 16-byte access of this lane inside its wave's row (pair index into the 4-row tile)
template <int LOGC>
MK_D int row3_pair(int g, int t, int i) { return g * (RowT<LOGC>::R / 2) + t + RowT<LOGC>::TPR * i; }

// ModUp row pass + eval-key inner product on the fp64 Q limbs (see k_row_inner_fp) on 512-point rows
template <int NPARTS, int LOGC>
__global__ __launch_bounds__(NTT_THREADS, 3) void k_row3_inner_fp(InnerArgs a, NttTables T) {
    using TL = RowT<LOGC>;
    constexpr int R = TL::R, S = TL::ROWS, TPR = TL::TPR, PAIRS = 4;
    __shared__ u64 lds[TL::WORDS + 2 * (TL::TWA + TL::TWB)];
    Row3Ctx c;
    c.lds = lds;
    c.twa = lds + TL::WORDS;
    c.twa_sh = c.twa + TL::TWA;
    c.twb = c.twa_sh + TL::TWA;
    c.twb_sh = c.twb + TL::TWB;
    const uint32_t n = 1u << T.log_n, r1 = 1u << T.log_r1;
    const uint32_t tiles = r1 / S, groups = tiles * a.nsel;
    uint32_t grp, item;
    group_member(blockIdx.x, groups, a.items, T.cu_affine, grp, item);
    const uint32_t sl = nth_set_bit(a.slot_mask, grp / tiles);
    const LimbConst lc = T.limb[sl];
    const int own = (int)(sl / a.alpha);
    const uint32_t row0 = (grp % tiles) * S;
    c.g = threadIdx.x / TPR;
    c.t = threadIdx.x % TPR;
    const u64 *tw = T.tw + (size_t)sl * n, *tw_sh = T.tw_sh + (size_t)sl * n;
    row3_stage_twiddles<LOGC>(c, tw, tw_sh, r1 + row0);
    u64 wc[7], wpc[7];
    row3_load_c_twiddles<LOGC>(tw, tw_sh, r1 + row0 + c.g, c.t, wc, wpc);
    const size_t tile_off = (size_t)row0 * R;
    const double q = lc.qd, qinv = lc.qinv;
    int jn = own == 0 ? 1 : 0;
    const u64 *dig0 = a.dig + ((size_t)item * NPARTS * a.ext + sl) * n + tile_off + (size_t)c.g * R + c.t;
    u64 x[8];
    if (jn < NPARTS) {
        const u64 *src = dig0 + (size_t)jn * a.ext * n;
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = ld_stream(src + TPR * k);
    }
    double2 acc0[PAIRS], acc1[PAIRS];
    {
        const u64 *y0 = a.c1 + (size_t)item * a.c1_stride + (size_t)sl * n + tile_off;
        const u64 *e0 = a.evk + (((size_t)own * 2 + 0) * a.D + sl) * n + tile_off;
        const u64 *e1 = a.evk + (((size_t)own * 2 + 1) * a.D + sl) * n + tile_off;
#pragma unroll
        for (int i = 0; i < PAIRS; ++i) {
            const int e = row3_pair<LOGC>(c.g, c.t, i);
            const ulong2 yy = ld_stream2(reinterpret_cast<const ulong2 *>(y0) + e);
            const ulong2 b = reinterpret_cast<const ulong2 *>(e0)[e];
            const ulong2 cc = reinterpret_cast<const ulong2 *>(e1)[e];
            const double yx = u52_to_double(yy.x), yz = u52_to_double(yy.y);
            acc0[i].x = fp_mulmod_any(yx, u52_to_double(b.x), q, qinv);
            acc0[i].y = fp_mulmod_any(yz, u52_to_double(b.y), q, qinv);
            acc1[i].x = fp_mulmod_any(yx, u52_to_double(cc.x), q, qinv);
            acc1[i].y = fp_mulmod_any(yz, u52_to_double(cc.y), q, qinv);
        }
    }
    __syncthreads();  // twiddles staged
#pragma unroll 1
    for (int dj = jn; dj < NPARTS; dj = jn) {
        jn = dj + 1 == own ? dj + 2 : dj + 1;
        wave_lds_sync();  // previous digit's products finished reading this wave's row
        row3_forward<AR_FP, LOGC>(x, c, wc, wpc, lc);
#pragma unroll
        for (int k = 0; k < 8; ++k) lds[TL::at(c.g, 8 * c.t + k)] = dbits(fp_reduce(bitsd(x[k]), q, qinv));
        if (jn < NPARTS) {
            const u64 *src = dig0 + (size_t)jn * a.ext * n;
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = ld_stream(src + TPR * k);
        }
        wave_lds_sync();
        const u64 *e0 = a.evk + (((size_t)dj * 2 + 0) * a.D + sl) * n + tile_off;
        const u64 *e1 = a.evk + (((size_t)dj * 2 + 1) * a.D + sl) * n + tile_off;
#pragma unroll
        for (int i = 0; i < PAIRS; ++i) {
            const int e = row3_pair<LOGC>(c.g, c.t, i);
            const int xx = (2 * e) % R;
            const ulong2 b = reinterpret_cast<const ulong2 *>(e0)[e];
            const ulong2 cc = reinterpret_cast<const ulong2 *>(e1)[e];
            const double yx = bitsd(lds[TL::at(c.g, xx)]), yz = bitsd(lds[TL::at(c.g, xx + 1)]);
            acc0[i].x += fp_mulmod_any(yx, u52_to_double(b.x), q, qinv);
            acc0[i].y += fp_mulmod_any(yz, u52_to_double(b.y), q, qinv);
            acc1[i].x += fp_mulmod_any(yx, u52_to_double(cc.x), q, qinv);
            acc1[i].y += fp_mulmod_any(yz, u52_to_double(cc.y), q, qinv);
        }
    }
    u64 *t0 = a.til + (((size_t)item * 2 + 0) * a.ext + sl) * n + tile_off;
    u64 *t1 = a.til + (((size_t)item * 2 + 1) * a.ext + sl) * n + tile_off;
#pragma unroll
    for (int i = 0; i < PAIRS; ++i) {
        const int e = row3_pair<LOGC>(c.g, c.t, i);
        ulong2 r0, r1v;
        r0.x = fp_to_canonical(acc0[i].x, q, qinv);
        r0.y = fp_to_canonical(acc0[i].y, q, qinv);
        r1v.x = fp_to_canonical(acc1[i].x, q, qinv);
        r1v.y = fp_to_canonical(acc1[i].y, q, qinv);
        st_stream2(reinterpret_cast<ulong2 *>(t0) + e, r0);
        st_stream2(reinterpret_cast<ulong2 *>(t1) + e, r1v);
    }
}

// Integer-class Q limb (q_0) of the merged n-client flow: forward row pass of the ModDown conversion SUMMED over the
// clients (k_icol_sum + k_conv_col_psum: one conversion and one transform per group instead of one per client), then
//   out = ( sum_c ctilde_c - conv_sum ) * P^-1 + sum_c c0_c   (component 0; without the c0 term on component 1)
// with the clients' key-switch accumulators ctilde_c read from the compact til of k_row3_inner_int.  Ring arithmetic mod
// q throughout, so the residues are those of the per-client chain.  Three-round geometry: 256- and 512-point rows.
struct TailOnceArgs {
    const u64 *conv;   // [poly][nl][N] column-passed summed conversions (lazy < 8q)
    const u64 *til;    // [client][poly][nsel][N] canonical accumulators of this launch's slots
    const u64 *cts;    // client c, ciphertext i at cts + c * ct_cstride + i * ct_stride: [2][nl][N]
    u64 *out;          // [poly][nl][N]
    const u64 *pinv, *pinv_sh;
    size_t til_cstride, ct_cstride, ct_stride;
    uint32_t n_clients, nl, n_polys;
    unsigned long long slot_mask;
    uint32_t nsel;
    uint32_t init_from_out;  // continue a running sum held in `out` (client groups)
};
template <int LOGC, int AR>
__global__ __launch_bounds__(NTT_THREADS, 3) void k_row3_tail_once(TailOnceArgs a, NttTables T) {
    using TL = RowT<LOGC>;
    constexpr int R = TL::R, S = TL::ROWS, TPR = TL::TPR, PAIRS = 4;
    __shared__ u64 lds[TL::WORDS + 2 * (TL::TWA + TL::TWB)];
    Row3Ctx c;
    c.lds = lds;
    c.twa = lds + TL::WORDS;
    c.twa_sh = c.twa + TL::TWA;
    c.twb = c.twa_sh + TL::TWA;
    c.twb_sh = c.twb + TL::TWB;
    const uint32_t n = 1u << T.log_n, r1 = 1u << T.log_r1;
    const uint32_t tiles = r1 / S, groups = tiles * a.nsel;
    uint32_t grp, poly;
    group_member(blockIdx.x, groups, a.n_polys, T.cu_affine, grp, poly);
    const uint32_t rank = grp / tiles, sl = nth_set_bit(a.slot_mask, rank);  // Q limb: slot == limb id
    const LimbConst lc = T.limb[sl];
    const uint32_t row0 = (grp % tiles) * S;
    c.g = threadIdx.x / TPR;
    c.t = threadIdx.x % TPR;
    const u64 *tw = T.tw + (size_t)sl * n, *tw_sh = T.tw_sh + (size_t)sl * n;
    row3_stage_twiddles<LOGC>(c, tw, tw_sh, r1 + row0);
    u64 wc[7], wpc[7];
    row3_load_c_twiddles<LOGC>(tw, tw_sh, r1 + row0 + c.g, c.t, wc, wpc);
    const size_t tile_off = (size_t)row0 * R;
    u64 x[8];
    {
        const u64 *src = a.conv + ((size_t)poly * a.nl + sl) * n + tile_off + (size_t)c.g * R + c.t;
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = ld_stream(src + TPR * k);
    }
    __syncthreads();  // twiddles staged
    row3_forward<AR, LOGC>(x, c, wc, wpc, lc);
#pragma unroll
    for (int k = 0; k < 8; ++k) lds[TL::at(c.g, 8 * c.t + k)] = canon8(x[k], lc.q, lc.q2);
    wave_lds_sync();
    const u64 pi = a.pinv[sl], pi_sh = a.pinv_sh[sl];
    const bool with_c0 = (poly & 1) == 0;
    u64 *dst = a.out + ((size_t)poly * a.nl + sl) * n + tile_off;
#pragma unroll
    for (int i = 0; i < PAIRS; ++i) {
        const int e = row3_pair<LOGC>(c.g, c.t, i);
        const int xx = (2 * e) % R;
        // sums of canonical residues (< 2^60): reduced every 8 terms, so they never pass 9 * 2^60 < 2^64
        u64 tx = 0, ty = 0, zx = 0, zy = 0;
        for (uint32_t cl = 0; cl < a.n_clients; ++cl) {
            const ulong2 tt = ld_stream2(reinterpret_cast<const ulong2 *>(
                                             a.til + (size_t)cl * a.til_cstride + ((size_t)poly * a.nsel + rank) * n + tile_off) + e);
            tx += tt.x;
            ty += tt.y;
            if (with_c0) {
                const ulong2 zz = ld_stream2(reinterpret_cast<const ulong2 *>(
                                                 a.cts + (size_t)cl * a.ct_cstride + (size_t)(poly >> 1) * a.ct_stride +
                                                 (size_t)sl * n + tile_off) + e);
                zx += zz.x;
                zy += zz.y;
            }
            if ((cl & 7) == 7) {
                tx = reduce_word(tx, lc);
                ty = reduce_word(ty, lc);
                zx = reduce_word(zx, lc);
                zy = reduce_word(zy, lc);
            }
        }
        tx = reduce_word(tx, lc);
        ty = reduce_word(ty, lc);
        ulong2 v;
        v.x = shoup_mul(sub_mod(tx, lds[TL::at(c.g, xx)], lc.q), pi, pi_sh, lc.q);
        v.y = shoup_mul(sub_mod(ty, lds[TL::at(c.g, xx + 1)], lc.q), pi, pi_sh, lc.q);
        if (with_c0) {
            v.x = add_mod(v.x, reduce_word(zx, lc), lc.q);
            v.y = add_mod(v.y, reduce_word(zy, lc), lc.q);
        }
        if (a.init_from_out) {
            const ulong2 o = reinterpret_cast<const ulong2 *>(dst)[e];
            v.x = add_mod(v.x, o.x, lc.q);
            v.y = add_mod(v.y, o.y, lc.q);
        }
        reinterpret_cast<ulong2 *>(dst)[e] = v;
    }
}

// inverse ROW pass of this thread's row (integer limbs): x[k] = word 8 t + k on entry (values in [0,2q)), on exit
// x[k] = word t + TPR k in [0,2q) -- what k_ntt_row_r<INV> hands to the inverse column pass.  c.twa / c.twb hold the
// INVERSE tables' round-A / round-B twiddles, wc / wpc the thread's inverse round-C twiddles.
template <int LOGC, int AR>
MK_D void row3_inverse_int(u64 (&x)[8], const Row3Ctx &c, const u64 (&wc)[7], const u64 (&wpc)[7], const LimbConst &lc) {
    using TL = RowT<LOGC>;
    constexpr int S = TL::ROWS, TPR = TL::TPR, C = TL::C;
    const int a = c.t / C, cc = c.t % C;
    if (LOGC == 3) {
        radix_inverse_any<3, AR>(x, wc, wpc, lc);
    } else {
        u64 y0[4] = {x[0], x[1], x[2], x[3]}, y1[4] = {x[4], x[5], x[6], x[7]};
        const u64 w0[3] = {wc[0], wc[1], wc[2]}, wp0[3] = {wpc[0], wpc[1], wpc[2]};
        const u64 w1[3] = {wc[3], wc[4], wc[5]}, wp1[3] = {wpc[3], wpc[4], wpc[5]};
        radix_inverse_any<2, AR>(y0, w0, wp0, lc);
        radix_inverse_any<2, AR>(y1, w1, wp1, lc);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            x[k] = y0[k];
            x[4 + k] = y1[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) c.lds[TL::at(c.g, 8 * c.t + k)] = x[k];
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = c.lds[TL::at(c.g, 8 * C * a + C * k + cc)];
    u64 w[7], wp[7];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int gg = 0; gg < (1 << s); ++gg) {
            const int e = S * 8 * ((1 << s) - 1) + ((c.g * 8 + a) << s) + gg;
            w[(1 << s) - 1 + gg] = c.twb[e];
            wp[(1 << s) - 1 + gg] = c.twb_sh[e];
        }
    radix_inverse_any<3, AR>(x, w, wp, lc);
#pragma unroll
    for (int k = 0; k < 8; ++k) c.lds[TL::at(c.g, 8 * C * a + C * k + cc)] = x[k];
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = c.lds[TL::at(c.g, c.t + TPR * k)];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int gg = 0; gg < (1 << s); ++gg) {
            const int e = S * ((1 << s) - 1) + (c.g << s) + gg;
            w[(1 << s) - 1 + gg] = c.twa[e];
            wp[(1 << s) - 1 + gg] = c.twa_sh[e];
        }
    radix_inverse_any<3, AR>(x, w, wp, lc);
}

// The same fusion for the INTEGER limbs (q0 and the P limbs): forward row pass of every converted digit + eval-key
// inner product with 128-bit accumulators (one Barrett at the end), three-round geometry (8 words per thread keep the
// 16 accumulator words + data + twiddles within 3 waves per SIMD).  a.slot_mask selects SLOTS (q0 = slot 0, P limbs =
// slots nl..ext-1); P limbs have no owning digit.
#ifndef MK_INVP_WAVES
#define MK_INVP_WAVES 2  // 168 VGPRs would spill 19 registers in the P instance; 2 waves measured +0.7 %
#endif
#ifndef MK_INNERQ_WAVES
#define MK_INNERQ_WAVES 2  // the q_0 instance too: no scratch (9 registers at 3 waves), 4 096 workgroups = 8 rounds of 512 slots, +0.36 %
#endif
template <int NPARTS, int LOGC, bool INVP, int AR>
__global__ __launch_bounds__(NTT_THREADS, INVP ? MK_INVP_WAVES : MK_INNERQ_WAVES) void k_row3_inner_int(InnerArgs a, NttTables T, uint32_t L, u64 *pc,
                                                                   uint32_t K) {
    using TL = RowT<LOGC>;
    constexpr int R = TL::R, S = TL::ROWS, TPR = TL::TPR, PAIRS = 4;
    __shared__ u64 lds[TL::WORDS + 2 * (TL::TWA + TL::TWB)];
    Row3Ctx c;
    c.lds = lds;
    c.twa = lds + TL::WORDS;
    c.twa_sh = c.twa + TL::TWA;
    c.twb = c.twa_sh + TL::TWA;
    c.twb_sh = c.twb + TL::TWB;
    const uint32_t n = 1u << T.log_n, r1 = 1u << T.log_r1;
    const uint32_t tiles = r1 / S, groups = tiles * a.nsel;
    uint32_t grp, item;
    group_member(blockIdx.x, groups, a.items, T.cu_affine, grp, item);
    const uint32_t sl = nth_set_bit(a.slot_mask, grp / tiles);
    const uint32_t id = limb_id_of(sl, a.nl, L);
    const LimbConst lc = T.limb[id];
    const int own = sl < a.nl ? (int)(sl / a.alpha) : -1;
    const uint32_t row0 = (grp % tiles) * S;
    c.g = threadIdx.x / TPR;
    c.t = threadIdx.x % TPR;
    const u64 *tw = T.tw + (size_t)id * n, *tw_sh = T.tw_sh + (size_t)id * n;
    row3_stage_twiddles<LOGC>(c, tw, tw_sh, r1 + row0);
    u64 wc[7], wpc[7];
    row3_load_c_twiddles<LOGC>(tw, tw_sh, r1 + row0 + c.g, c.t, wc, wpc);
    const size_t tile_off = (size_t)row0 * R;
    const u64 *evk = inner_evk(a, item);
    int jn = own == 0 ? 1 : 0;
    const u64 *dig0 = a.dig + ((size_t)item * NPARTS * a.ext + sl) * n + tile_off + (size_t)c.g * R + c.t;
    u64 x[8];
    if (jn < NPARTS) {
        const u64 *src = dig0 + (size_t)jn * a.ext * n;
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = ld_stream(src + TPR * k);
    }
    u64 h0[2 * PAIRS], l0[2 * PAIRS], h1[2 * PAIRS], l1[2 * PAIRS];
#pragma unroll
    for (int i = 0; i < 2 * PAIRS; ++i) h0[i] = l0[i] = h1[i] = l1[i] = 0;
    if (own >= 0) {  // the digit that owns this limb: c1 itself
        const u64 *y0 = inner_c1(a, item) + (size_t)sl * n + tile_off;
        const u64 *e0 = evk + (((size_t)own * 2 + 0) * a.D + id) * n + tile_off;
        const u64 *e1 = evk + (((size_t)own * 2 + 1) * a.D + id) * n + tile_off;
#pragma unroll
        for (int i = 0; i < PAIRS; ++i) {
            const int e = row3_pair<LOGC>(c.g, c.t, i);
            const ulong2 yy = ld_stream2(reinterpret_cast<const ulong2 *>(y0) + e);
            const ulong2 b = reinterpret_cast<const ulong2 *>(e0)[e];
            const ulong2 cc = reinterpret_cast<const ulong2 *>(e1)[e];
            mac128(h0[2 * i], l0[2 * i], yy.x, b.x);
            mac128(h0[2 * i + 1], l0[2 * i + 1], yy.y, b.y);
            mac128(h1[2 * i], l1[2 * i], yy.x, cc.x);
            mac128(h1[2 * i + 1], l1[2 * i + 1], yy.y, cc.y);
        }
    }
    __syncthreads();  // twiddles staged
#pragma unroll 1
    for (int dj = jn; dj < NPARTS; dj = jn) {
        jn = dj + 1 == own ? dj + 2 : dj + 1;
        wave_lds_sync();
        row3_forward<AR, LOGC>(x, c, wc, wpc, lc);
#pragma unroll
        // AR_PM: the lazy words (< 7.001U) go into the products as they are -- pm_reduce128 takes 6 x 2^63 x q
        for (int k = 0; k < 8; ++k) lds[TL::at(c.g, 8 * c.t + k)] = AR == AR_PM ? x[k] : canon8(x[k], lc.q, lc.q2);
        if (jn < NPARTS) {
            const u64 *src = dig0 + (size_t)jn * a.ext * n;
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = ld_stream(src + TPR * k);
        }
        wave_lds_sync();
        const u64 *e0 = evk + (((size_t)dj * 2 + 0) * a.D + id) * n + tile_off;
        const u64 *e1 = evk + (((size_t)dj * 2 + 1) * a.D + id) * n + tile_off;
        ulong2 eb[PAIRS], ec[PAIRS];
        // the digit's eval-key tiles in one burst (see k_qsum3_fp; +0.2 % on the step) in the 2-wave instances, which have the
        // registers; a 3-wave instance would park 14 more in scratch
        constexpr bool BURST = INVP ? MK_INVP_WAVES == 2 : MK_INNERQ_WAVES == 2;
        if (BURST) {
#pragma unroll
            for (int i = 0; i < PAIRS; ++i) {
                const int e = row3_pair<LOGC>(c.g, c.t, i);
                eb[i] = reinterpret_cast<const ulong2 *>(e0)[e];
                ec[i] = reinterpret_cast<const ulong2 *>(e1)[e];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < PAIRS; ++i) {
            const int e = row3_pair<LOGC>(c.g, c.t, i);
            const int xx = (2 * e) % R;
            const ulong2 b = BURST ? eb[i] : reinterpret_cast<const ulong2 *>(e0)[e];
            const ulong2 cc = BURST ? ec[i] : reinterpret_cast<const ulong2 *>(e1)[e];
            const u64 yx = lds[TL::at(c.g, xx)], yz = lds[TL::at(c.g, xx + 1)];
            mac128(h0[2 * i], l0[2 * i], yx, b.x);
            mac128(h0[2 * i + 1], l0[2 * i + 1], yz, b.y);
            mac128(h1[2 * i], l1[2 * i], yx, cc.x);
            mac128(h1[2 * i + 1], l1[2 * i + 1], yz, cc.y);
        }
    }
    // P limbs with `pc` given: the results go straight through the INVERSE row pass (first pass of ApproxModDown's
    // SetFormat(COEFFICIENT)) and land in pc[2 item + comp][K][N]; the accumulators over P never reach HBM
    constexpr bool inv = INVP;  // this instance is launched over P slots only, with pc given
    const u64 *itw = T.itw + (size_t)id * n, *itw_sh = T.itw_sh + (size_t)id * n;
    if (inv) {
        __syncthreads();  // every wave is done with the forward round-A/B twiddles
        row3_stage_twiddles<LOGC>(c, itw, itw_sh, r1 + row0);
        __syncthreads();
    }
#pragma unroll 1
    for (int comp = 0; comp < 2; ++comp) {
        ulong2 res[PAIRS];
#pragma unroll
        for (int i = 0; i < PAIRS; ++i) {
            const u64 hx = comp ? h1[2 * i] : h0[2 * i], lx = comp ? l1[2 * i] : l0[2 * i];
            const u64 hy = comp ? h1[2 * i + 1] : h0[2 * i + 1], ly = comp ? l1[2 * i + 1] : l0[2 * i + 1];
            if (AR == AR_PM) {
                const PmK P = pm_consts(lc);
                res[i].x = pm_reduce128(hx, lx, P, lc.q);
                res[i].y = pm_reduce128(hy, ly, P, lc.q);
            } else {
                res[i].x = NPARTS <= 4 ? reduce_sum4(hx, lx, lc) : reduce_wide(hx, lx, lc);
                res[i].y = NPARTS <= 4 ? reduce_sum4(hy, ly, lc) : reduce_wide(hy, ly, lc);
            }
        }
        if (!inv) {
            u64 *td = a.til + (a.til_compact ? ((size_t)item * 2 + comp) * a.nsel + grp / tiles
                                             : ((size_t)item * 2 + comp) * a.ext + sl) * n + tile_off;
#pragma unroll
            for (int i = 0; i < PAIRS; ++i) st_stream2(reinterpret_cast<ulong2 *>(td) + row3_pair<LOGC>(c.g, c.t, i), res[i]);
        } else {
            wave_lds_sync();  // the previous use of this wave's row (last digit's products / component 0) is over
#pragma unroll
            for (int i = 0; i < PAIRS; ++i) {
                const int xx = (2 * row3_pair<LOGC>(c.g, c.t, i)) % R;
                lds[TL::at(c.g, xx)] = res[i].x;
                lds[TL::at(c.g, xx + 1)] = res[i].y;
            }
            wave_lds_sync();
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = lds[TL::at(c.g, 8 * c.t + k)];
            u64 iwc[7], iwpc[7];  // (re)loaded per component: keeps them out of the accumulators' live range
            row3_load_c_twiddles<LOGC>(itw, itw_sh, r1 + row0 + c.g, c.t, iwc, iwpc);
            row3_inverse_int<LOGC, AR>(x, c, iwc, iwpc, lc);
            u64 *pd = pc + (((size_t)item * 2 + comp) * K + (sl - a.nl)) * n + tile_off + (size_t)c.g * R + c.t;
#pragma unroll
            for (int k = 0; k < 8; ++k) st_pass(pd + TPR * k, x[k]);  // lazy [0,2q): the inverse column pass scales
        }
    }
}

}  // namespace mk
