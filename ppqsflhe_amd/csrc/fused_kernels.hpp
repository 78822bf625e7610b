// fused_kernels.hpp -- round-2 fusions of the hybrid key switch (gfx950).
//
//   k_icol_conv_col : last pass of the INVERSE transform of a digit's source limbs (column pass, scaling by
//                     N^-1 * [(S/s_i)^-1]_{s_i}) + ApproxSwitchCRTBasis + FORWARD column pass of every target limb, in
//                     one workgroup per (polynomial, digit, column tile).  The inverse column pass and the forward
//                     column pass work on the SAME tile (R1 rows x S columns) with the SAME thread -> element map
//                     (thread (j, c) ends the inverse pass holding rows j + H k of column c and starts the forward pass
//                     from exactly those rows), so the scaled sources stay in registers: the coefficient-format
//                     polynomial never goes to HBM (EvalKeySwitchPrecomputeCore's SetFormat(COEFFICIENT) +
//                     ApproxSwitchCRTBasis + SetFormat(EVALUATION), and the same triple inside ApproxModDown;
//                     [upstream] keyswitch-hybrid.cpp, dcrtpoly-impl.h).
//   k_qsum_fp       : for an fp64-class Q limb t and a row tile, over ALL clients: forward row pass of every converted
//                     digit + eval-key inner product, forward row pass of ModDown's converted limbs, the ModDown tail
//                     and the coefficient-wise sum over clients -- the key-switch accumulators over Q never reach HBM
//                     (EvalFastKeySwitchCoreExt + ApproxModDown's tail + EvalAdd chain,
//                     changeCipherDomain.cpp:74 x n_clients, aggregateEncryptedWeights.cpp:82).
//
// Bit-exactness: ApproxSwitchCRTBasis needs the CANONICAL residue [x * (S/s_i)^-1]_{s_i} in [0, s_i) of every source
// (another representative shifts the result by a multiple of S); everything downstream is ring arithmetic mod the
// target, where only the final canonical representative is stored.
#pragma once
#include "ntt_radix.hpp"

namespace mk {

// scheduling fence every MK_CONV_SB elements of a conversion: without it the scheduler interleaves all 16 elements of a
// thread and their temporaries push the kernel far beyond the register file
#ifndef MK_CONV_SB
#define MK_CONV_SB 2
#endif

#ifndef MK_BF_SB
#define MK_BF_SB 0  // fences inside the radix stages: measured to raise the spill count, off
#endif

struct FusedIo {
    const u64 *in;   // [items][in_slots][N]: sources after the inverse ROW pass (lazy [0,2q) / doubles on fp limbs)
    u64 *out;        // [items](digit)[out_slots][N]: column-passed targets (lazy u64 / doubles on fp limbs)
    size_t in_stride, out_stride;  // words between items
    size_t out_part_stride;        // words between the digits of one item in `out` (0 for ModDown)
    uint32_t items;
    uint32_t part0, nparts;        // digits cvs[part0 .. part0 + nparts) are handled by this launch (same fan-in)
    const u64 *scale, *scale_sh;   // per limb id: N^-1 * hatinv as (u64, Shoup) or (double, double / q)
};

// canonical integer below 2^52 held in a double -> its 30-bit halves (what split30 gives for the u64)
MK_D void split30_d(u64 dbl_bits, uint32_t &lo, uint32_t &hi) {
    const u64 b = dbits(bitsd(dbl_bits) + 4503599627370496.0) & 0xFFFFFFFFFFFFFull;  // mantissa of 2^52 + v = v
    lo = (uint32_t)b & 0x3FFFFFFFu;
    hi = (uint32_t)(b >> 30);
}

// ---- column rounds with the twiddles of BOTH rounds staged in LDS ------------------------------------------------
// A column kernel's round-A twiddles are the same for every thread (table entries 1 .. H-1) and its round-B twiddles
// depend on j only (entries ((H + j) << s) + g): 16 H - 1 pairs (w, companion) per transform.  Loaded per thread they
// are 2 (H - 1) global loads and 4 (H - 1) live registers; staged once per workgroup they are one 16-byte LDS write per
// thread, and the rounds read each pair (a broadcast within a wave) right where it is used.
template <int LOG_H>
struct ColTw {
    static constexpr int H = 1 << LOG_H, PAIRS = (H + 1) * (H - 1);  // j = 0..H-1: round B of thread row j; j = H: round A
    static MK_D void stage(ulong2 *twl, const u64 *tw, const u64 *tw_sh) {
        for (int e = threadIdx.x; e < PAIRS; e += NTT_THREADS) {
            const int jj = e / (H - 1), idx = e % (H - 1);
            const int s = 31 - __clz(idx + 1), g = idx - ((1 << s) - 1);
            const uint32_t base = jj < H ? (uint32_t)(H + jj) : 1u;
            const uint32_t ti = (base << s) + (uint32_t)g;
            twl[e] = ulong2{tw[ti], tw_sh[ti]};
        }
    }
};
template <int LOG_H, bool FP>
MK_D void radix_forward_l(u64 (&x)[1 << LOG_H], const ulong2 *twl, const LimbConst &lc) {
    constexpr int H = 1 << LOG_H;
    const u64 q4 = lc.q2 + lc.q2;
#pragma unroll
    for (int s = 0; s < LOG_H; ++s) {
        const int dist = H >> (s + 1);
#pragma unroll
        for (int g = 0; g < (1 << s); ++g) {
            const ulong2 t = twl[(1 << s) - 1 + g];
#pragma unroll
            for (int p = 0; p < dist; ++p) {
                const int k0 = g * 2 * dist + p;
                if (FP) {
                    if (s % 2 == 0) ct_butterfly_fp(x[k0], x[k0 + dist], t.x, t.y, lc.qd, lc.qinv);
                    else ct_butterfly_fp_nr(x[k0], x[k0 + dist], t.x, t.y, lc.qd);
                } else {
                    if (s % 2 == 0) ct_butterfly_c4(x[k0], x[k0 + dist], t.x, t.y, lc.q, lc.q2, q4);
                    else ct_butterfly_nc(x[k0], x[k0 + dist], t.x, t.y, lc.q, lc.q2);
                }
                if (MK_BF_SB && (g * dist + p + 1) % MK_BF_SB == 0) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}
template <int LOG_H, bool FP>
MK_D void radix_inverse_l(u64 (&x)[1 << LOG_H], const ulong2 *twl, const LimbConst &lc) {
    constexpr int H = 1 << LOG_H;
#pragma unroll
    for (int s = LOG_H - 1; s >= 0; --s) {
        const int dist = H >> (s + 1);
#pragma unroll
        for (int g = 0; g < (1 << s); ++g) {
            const ulong2 t = twl[(1 << s) - 1 + g];
#pragma unroll
            for (int p = 0; p < dist; ++p) {
                const int k0 = g * 2 * dist + p;
                if (FP) gs_butterfly_fp(x[k0], x[k0 + dist], t.x, t.y, lc.qd, lc.qinv);
                else gs_butterfly(x[k0], x[k0 + dist], t.x, t.y, lc.q, lc.q2);
                if (MK_BF_SB && (g * dist + p + 1) % MK_BF_SB == 0) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// arithmetic class of source i under SRCMODE: 0 = all integer, 1 = all fp64, 2 = source 0 integer, the rest fp64
template <int SRCMODE>
MK_D constexpr bool src_is_fp(int i) {
    return SRCMODE == 1 || (SRCMODE == 2 && i > 0);
}

// phase 1 of k_icol_conv_col for source I (and, recursively, the following ones): a compile-time recursion instead of an
// unrolled loop -- the body (two radix rounds on 16 words) is beyond the size up to which #pragma unroll is honoured,
// and a rolled loop would index sv[][] dynamically, i.e. put it into scratch memory
template <int LOG_H>
struct IcolCtx {
    u64 *lds2;
    ulong2 *twl2;
    const u64 *src;
    uint32_t n, r2;
    int j, c;
};
template <int LOG_H, int N_IN, int SRCMODE, int I, typename StageFn>
MK_D void icol_sources(u64 (&sv)[N_IN][1 << LOG_H], const IcolCtx<LOG_H> &cx, const FusedIo &io, const NttTables &T,
                       const DevConv &cv, StageFn &stage_unit) {
    if constexpr (I < N_IN) {
        using TL = ColTile<LOG_H>;
        using TW = ColTw<LOG_H>;
        constexpr int H = TL::H;
        constexpr bool FP = src_is_fp<SRCMODE>(I);
        const int j = cx.j, c = cx.c;
        const uint32_t id = cv.src_id[I];
        const LimbConst lc = T.limb[id];
        u64 *lds = cx.lds2 + (I & 1) * TL::WORDS;
        const ulong2 *twl = cx.twl2 + (I % 3) * TW::PAIRS;
        const u64 *p = cx.src + (size_t)cv.src_slot[I] * cx.n;
        u64 x[H];
#pragma unroll
        for (int k = 0; k < H; ++k) x[k] = ld_pass(p + (size_t)(H * j + k) * cx.r2);
        stage_unit((uint32_t)I + 1);
        radix_inverse_l<LOG_H, FP>(x, twl + j * (H - 1), lc);
#pragma unroll
        for (int k = 0; k < H; ++k) lds[TL::at(j, k, c)] = x[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < H; ++k) x[k] = lds[TL::at(k, j, c)];
        radix_inverse_l<LOG_H, FP>(x, twl + H * (H - 1), lc);
        const u64 sc = io.scale[id], sc_sh = io.scale_sh[id];
#pragma unroll
        for (int k = 0; k < H; ++k) {
            if (FP) {
                // |x| <= 1.33 q  ->  |s| <= 0.92 q: one conditional add gives the canonical residue
                double s = fp_mulmod(bitsd(x[k]), bitsd(sc), bitsd(sc_sh), lc.qd);
                s = s < 0.0 ? s + lc.qd : s;
                sv[I][k] = dbits(s);
            } else {
                sv[I][k] = pack30(shoup_mul(x[k], sc, sc_sh, lc.q));
            }
        }
        icol_sources<LOG_H, N_IN, SRCMODE, I + 1>(sv, cx, io, T, cv, stage_unit);
    }
}

// Work of a workgroup = N_IN + n_out "units" (one column transform of H*H points each).  Every unit has exactly one
// workgroup barrier, between its two rounds.  Exchange e uses tile e & 1: a thread writing tile e & 1 has passed barrier
// e - 1, so every thread has finished reading that tile in unit e - 2.  Unit u stages the twiddles of unit u + 1 ahead of
// its barrier (which publishes them) into buffer (u + 1) % 3: the last readers of that buffer were in unit u - 2, which
// every thread left before barrier u - 1 (two buffers would not do: unit u - 1's second round runs after barrier u - 1).
// Used for 64-point columns (H = 8); 256-point columns take k_icol3_conv_col (8 words per thread).
template <int LOG_H, int N_IN, int SRCMODE>
__global__ __launch_bounds__(NTT_THREADS, 2) void k_icol_conv_col(FusedIo io, NttTables T, const DevConv *cvs) {
    using TL = ColTile<LOG_H>;
    using TW = ColTw<LOG_H>;
    constexpr int H = TL::H, S = TL::S;
    __shared__ u64 lds2[2 * TL::WORDS];
    __shared__ ulong2 twl2[3 * TW::PAIRS];
    const uint32_t n = 1u << T.log_n, r2 = 1u << T.log_r2, tiles = r2 / S;
    uint32_t b = blockIdx.x;
    const uint32_t tile = b % tiles;
    b /= tiles;
    const uint32_t part = io.part0 + b % io.nparts;
    const uint32_t item = b / io.nparts;
    const DevConv &cv = cvs[part];
    const uint32_t n_out = cv.n_out;
    const int c = threadIdx.x % S, j = threadIdx.x / S;
    const u64 *src = io.in + (size_t)item * io.in_stride + tile * S + c;
    // twiddles of unit u into buffer u % 3 (sources use the inverse tables, targets the forward ones)
    auto stage_unit = [&](uint32_t u) {
        if (u < (uint32_t)N_IN) {
            const uint32_t id = cv.src_id[u];
            TW::stage(twl2 + (u % 3) * TW::PAIRS, T.itw + (size_t)id * n, T.itw_sh + (size_t)id * n);
        } else if (u < (uint32_t)N_IN + n_out) {
            const uint32_t id = cv.dst_id[u - N_IN];
            TW::stage(twl2 + (u % 3) * TW::PAIRS, T.tw + (size_t)id * n, T.tw_sh + (size_t)id * n);
        }
    };
    stage_unit(0);
    // ---- phase 1: inverse column pass of the N_IN sources; results stay in registers -------------------------
    // sv[i][k] = [x_i * N^-1 * (S/s_i)^-1]_{s_i} at row j + H k: canonical, as a double (fp64-class source) or as
    // packed 30-bit halves (integer-class source)
    u64 sv[N_IN][H];
    __syncthreads();  // unit 0's twiddles are published
    IcolCtx<LOG_H> cx{lds2, twl2, src, n, r2, j, c};
    icol_sources<LOG_H, N_IN, SRCMODE, 0>(sv, cx, io, T, cv, stage_unit);
    // ---- phase 2: every target limb: conversion + forward column pass ------------------------------------------
    u64 *dst0 = io.out + (size_t)item * io.out_stride + (size_t)part * io.out_part_stride + tile * S + c;
#pragma unroll 1
    for (uint32_t t = 0; t < n_out; ++t) {
        const uint32_t u = (uint32_t)N_IN + t;
        const uint32_t id = cv.dst_id[t];
        const LimbConst lc = T.limb[id];
        u64 *lds = lds2 + (u & 1) * TL::WORDS;
        const ulong2 *twl = twl2 + (u % 3) * TW::PAIRS;
        u64 *dst = dst0 + (size_t)cv.dst_slot[t] * n;
        u64 x[H];
        stage_unit(u + 1);
        if (lc.fp) {
            // fp64-class target: fp64-class sources enter as exact fp64 modular products, integer-class sources
            // through the 30-bit column accumulation (their sum is < 4q < 2^53, exact in a double)
            double hd[N_IN], hq[N_IN];
            uint32_t h0[N_IN], h1[N_IN];
#pragma unroll
            for (int i = 0; i < N_IN; ++i) {
                if (src_is_fp<SRCMODE>(i)) {
                    hd[i] = cv.hat_d[i * n_out + t];
                    hq[i] = cv.hatq_d[i * n_out + t];
                } else {
                    split30(cv.hat[i * n_out + t], h0[i], h1[i]);
                }
            }
#pragma unroll
            for (int k = 0; k < H; ++k) {
                double acc = 0.0;
                if (SRCMODE != 1) {  // integer-class sources
                    Cols ia{0, 0, 0};
#pragma unroll
                    for (int i = 0; i < N_IN; ++i)
                        if (!src_is_fp<SRCMODE>(i))
                            mac_cols(ia, (uint32_t)sv[i][k], (uint32_t)(sv[i][k] >> 32), h0[i], h1[i]);
                    acc = (double)reduce_cols_lazy(ia, lc);
                }
#pragma unroll
                for (int i = 0; i < N_IN; ++i)
                    if (src_is_fp<SRCMODE>(i)) acc += fp_mulmod(bitsd(sv[i][k]), hd[i], hq[i], lc.qd);
                // |acc| <= 4q + N_IN * 0.8q < 2^53 (N_IN <= 4, q < 1.25 * 2^50): exact; bring into the rounds' range
                x[k] = dbits(fp_reduce(acc, lc.qd, lc.qinv));
                if ((k + 1) % MK_CONV_SB == 0) __builtin_amdgcn_sched_barrier(0);
            }
            radix_forward_l<LOG_H, true>(x, twl + H * (H - 1), lc);
        } else {
            uint32_t h0[N_IN], h1[N_IN];
#pragma unroll
            for (int i = 0; i < N_IN; ++i) split30(cv.hat[i * n_out + t], h0[i], h1[i]);
#pragma unroll
            for (int k = 0; k < H; ++k) {
                Cols ia{0, 0, 0};
#pragma unroll
                for (int i = 0; i < N_IN; ++i) {
                    uint32_t a0, a1;
                    if (src_is_fp<SRCMODE>(i)) {
                        split30_d(sv[i][k], a0, a1);
                    } else {
                        a0 = (uint32_t)sv[i][k];
                        a1 = (uint32_t)(sv[i][k] >> 32);
                    }
                    mac_cols(ia, a0, a1, h0[i], h1[i]);
                }
                x[k] = reduce_cols_lazy(ia, lc);  // < 4q: the first butterfly stage accepts < 8q
                if ((k + 1) % MK_CONV_SB == 0) __builtin_amdgcn_sched_barrier(0);
            }
            radix_forward_l<LOG_H, false>(x, twl + H * (H - 1), lc);
        }
#pragma unroll
        for (int k = 0; k < H; ++k) lds[TL::at(k, j, c)] = x[k];  // row j + H k
        __syncthreads();
#pragma unroll
        for (int k = 0; k < H; ++k) x[k] = lds[TL::at(j, k, c)];  // row H j + k
        if (lc.fp) radix_forward_l<LOG_H, true>(x, twl + j * (H - 1), lc);
        else radix_forward_l<LOG_H, false>(x, twl + j * (H - 1), lc);
#pragma unroll
        for (int k = 0; k < H; ++k) st_pass(dst + (size_t)(H * j + k) * r2, x[k]);  // lazy u64, or doubles on an fp limb
    }
}

// =====================================================================================================================
// 256-point columns with 8 words per thread: three rounds of radix 8, 8, 4 (k_icol3_conv_col)
// =====================================================================================================================
// Row r = 32 a + 4 b + c4 of the column (a, b < 8, c4 < 4); 32 threads per column, thread t:
//   round A (stages 0-2): rows t + 32 k            (a = k)                 twiddles (1 << s) + g: uniform
//   round B (stages 3-5): rows 32 (t/4) + 4 k + t%4 (b = k)                twiddles ((8 + t/4) << s) + g
//   round C (stages 6-7): rows 8 t + k: two radix-4 groups k < 4, k >= 4   twiddles ((64 + 2 t + k/4) << s) + g
// (the forward transform runs A, B, C; the inverse C, B, A with the inverse table).  A tile is 256 rows x 16 columns,
// 512 threads: the N_IN x 8 source words of a thread are 64 registers, the kernel fits 128 registers and runs 4 waves
// per SIMD (two workgroups per CU).  All 255 column twiddles of a limb (table entries 1..255) are staged in LDS.
template <int S_>
struct Col3T {
    static constexpr int S = S_, THREADS = 32 * S, TILE_Q = 256 * S, TILE_P = (256 + 32) * S, TW = 256;
    // tile Q (even exchanges): plain rows; tile P (odd exchanges): one pad row per 8 rows, so that the rows 8 t + k of
    // two neighbouring threads (9 rows apart) fall into different halves of the 64 banks
    static MK_D int q_at(int r, int c) { return r * S + c; }
    static MK_D int p_at(int r, int c) { return (r + (r >> 3)) * S + c; }
};
using Col3 = Col3T<16>;
template <int LOG_R, bool FP>
MK_D void radix_fwd_tab(u64 (&x)[1 << LOG_R], const ulong2 *twl, uint32_t base, const LimbConst &lc) {
    constexpr int R = 1 << LOG_R;
    const u64 q4 = lc.q2 + lc.q2;
#pragma unroll
    for (int s = 0; s < LOG_R; ++s) {
        const int dist = R >> (s + 1);
#pragma unroll
        for (int g = 0; g < (1 << s); ++g) {
            const ulong2 t = twl[(base << s) + g];
#pragma unroll
            for (int p = 0; p < dist; ++p) {
                const int k0 = g * 2 * dist + p;
                if (FP) {
                    if (s % 2 == 0) ct_butterfly_fp(x[k0], x[k0 + dist], t.x, t.y, lc.qd, lc.qinv);
                    else ct_butterfly_fp_nr(x[k0], x[k0 + dist], t.x, t.y, lc.qd);
                } else {
                    if (s % 2 == 0) ct_butterfly_c4(x[k0], x[k0 + dist], t.x, t.y, lc.q, lc.q2, q4);
                    else ct_butterfly_nc(x[k0], x[k0 + dist], t.x, t.y, lc.q, lc.q2);
                }
            }
        }
    }
}
template <int LOG_R, bool FP>
MK_D void radix_inv_tab(u64 (&x)[1 << LOG_R], const ulong2 *twl, uint32_t base, const LimbConst &lc) {
    constexpr int R = 1 << LOG_R;
#pragma unroll
    for (int s = LOG_R - 1; s >= 0; --s) {
        const int dist = R >> (s + 1);
#pragma unroll
        for (int g = 0; g < (1 << s); ++g) {
            const ulong2 t = twl[(base << s) + g];
#pragma unroll
            for (int p = 0; p < dist; ++p) {
                const int k0 = g * 2 * dist + p;
                if (FP) gs_butterfly_fp(x[k0], x[k0 + dist], t.x, t.y, lc.qd, lc.qinv);
                else gs_butterfly(x[k0], x[k0 + dist], t.x, t.y, lc.q, lc.q2);
            }
        }
    }
}
// the two radix-4 groups of round C on x[0..3] and x[4..7]
template <bool FP, bool INV>
MK_D void col3_round_c(u64 (&x)[8], const ulong2 *twl, int t, const LimbConst &lc) {
    u64 y0[4] = {x[0], x[1], x[2], x[3]}, y1[4] = {x[4], x[5], x[6], x[7]};
    if (INV) {
        radix_inv_tab<2, FP>(y0, twl, 64u + 2u * (uint32_t)t, lc);
        radix_inv_tab<2, FP>(y1, twl, 65u + 2u * (uint32_t)t, lc);
    } else {
        radix_fwd_tab<2, FP>(y0, twl, 64u + 2u * (uint32_t)t, lc);
        radix_fwd_tab<2, FP>(y1, twl, 65u + 2u * (uint32_t)t, lc);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        x[k] = y0[k];
        x[4 + k] = y1[k];
    }
}
struct Col3Ctx {
    u64 *tq, *tp;     // exchange tiles (even / odd exchanges)
    ulong2 *twl2;     // two twiddle buffers of Col3::TW pairs
    int t, c;         // thread inside the column, column inside the tile
};
template <typename G>
MK_D void col3_stage_twiddles(ulong2 *twl, const u64 *tw, const u64 *tw_sh) {
    for (int e = threadIdx.x; e < G::TW; e += G::THREADS)
        if (e >= 1) twl[e] = ulong2{tw[e], tw_sh[e]};
}
// inverse column pass of one source limb: x[k] = row 8 t + k on entry (output of the inverse row pass), row t + 32 k on
// exit (lazy [0,2q), or doubles with |x| <= 1.33 q).  `stage_next` runs between the two barriers.
template <typename G, bool FP, typename StageFn>
MK_D void col3_inverse(u64 (&x)[8], const Col3Ctx &cx, const ulong2 *twl, const LimbConst &lc, StageFn &&stage_next) {
    const int t = cx.t, c = cx.c, a = t >> 2, c4 = t & 3;
    col3_round_c<FP, true>(x, twl, t, lc);
#pragma unroll
    for (int k = 0; k < 8; ++k) cx.tq[G::q_at(8 * t + k, c)] = x[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = cx.tq[G::q_at(32 * a + 4 * k + c4, c)];
    stage_next();
    radix_inv_tab<3, FP>(x, twl, 8u + (uint32_t)a, lc);
#pragma unroll
    for (int k = 0; k < 8; ++k) cx.tp[G::p_at(32 * a + 4 * k + c4, c)] = x[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = cx.tp[G::p_at(t + 32 * k, c)];
    radix_inv_tab<3, FP>(x, twl, 1u, lc);
}
// forward column pass: x[k] = row t + 32 k on entry, row 8 t + k on exit (lazy < 8q / doubles: the row pass finishes)
template <typename G, bool FP, typename StageFn>
MK_D void col3_forward(u64 (&x)[8], const Col3Ctx &cx, const ulong2 *twl, const LimbConst &lc, StageFn &&stage_next) {
    const int t = cx.t, c = cx.c, a = t >> 2, c4 = t & 3;
    radix_fwd_tab<3, FP>(x, twl, 1u, lc);
#pragma unroll
    for (int k = 0; k < 8; ++k) cx.tq[G::q_at(t + 32 * k, c)] = x[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = cx.tq[G::q_at(32 * a + 4 * k + c4, c)];
    stage_next();
    radix_fwd_tab<3, FP>(x, twl, 8u + (uint32_t)a, lc);
#pragma unroll
    for (int k = 0; k < 8; ++k) cx.tp[G::p_at(32 * a + 4 * k + c4, c)] = x[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = cx.tp[G::p_at(8 * t + k, c)];
    col3_round_c<FP, false>(x, twl, t, lc);
}

// sources of k_icol3_conv_col, compile-time recursion over the source index (see icol_sources)
template <typename G, int N_IN, int SRCMODE, int I, typename StageFn>
MK_D void icol3_sources(u64 (&sv)[N_IN][8], const Col3Ctx &cx, const u64 *src, uint32_t n, uint32_t r2, const FusedIo &io,
                        const NttTables &T, const DevConv &cv, StageFn &stage_unit) {
    if constexpr (I < N_IN) {
        constexpr bool FP = src_is_fp<SRCMODE>(I);
        const uint32_t id = cv.src_id[I];
        const LimbConst lc = T.limb[id];
        const ulong2 *twl = cx.twl2 + (I & 1) * G::TW;
        const u64 *p = src + (size_t)cv.src_slot[I] * n;
        u64 x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = ld_pass(p + (size_t)(8 * cx.t + k) * r2);
        col3_inverse<G, FP>(x, cx, twl, lc, [&] { stage_unit((uint32_t)I + 1); });
        const u64 sc = io.scale[id], sc_sh = io.scale_sh[id];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (FP) {
                // |x| <= 1.33 q  ->  |s| <= 0.92 q: one conditional add gives the canonical residue
                double s = fp_mulmod(bitsd(x[k]), bitsd(sc), bitsd(sc_sh), lc.qd);
                s = s < 0.0 ? s + lc.qd : s;
                sv[I][k] = dbits(s);
            } else {
                sv[I][k] = pack30(shoup_mul(x[k], sc, sc_sh, lc.q));
            }
        }
        icol3_sources<G, N_IN, SRCMODE, I + 1>(sv, cx, src, n, r2, io, T, cv, stage_unit);
    }
}

// Synchronisation: a unit (one column transform) has two exchanges and two workgroup barriers; exchange e uses tile
// e & 1 (tq for the first, tp for the second exchange of every unit): a thread that writes a tile in exchange e has
// passed barrier e - 1, so every thread has finished its reads of exchange e - 2.  The twiddles of unit u + 1 are staged
// between the two barriers of unit u into buffer (u + 1) & 1 -- its last readers were in unit u - 1, which every thread
// left before the first barrier of unit u -- and are published by the second barrier of unit u.
template <int N_IN, int SRCMODE, int S_, int MINW>
__global__ __launch_bounds__(32 * S_, MINW) void k_icol3_conv_col(FusedIo io, NttTables T, const DevConv *cvs) {
    using G = Col3T<S_>;
    constexpr int S = G::S;
    __shared__ u64 tile_q[G::TILE_Q];
    __shared__ u64 tile_p[G::TILE_P];
    __shared__ ulong2 twl2[2 * G::TW];
    const uint32_t n = 1u << T.log_n, r2 = 1u << T.log_r2, tiles = r2 / S;
    uint32_t b = blockIdx.x;
    const uint32_t tile = b % tiles;
    b /= tiles;
    const uint32_t part = io.part0 + b % io.nparts;
    const uint32_t item = b / io.nparts;
    const DevConv &cv = cvs[part];
    const uint32_t n_out = cv.n_out;
    Col3Ctx cx{tile_q, tile_p, twl2, (int)(threadIdx.x / S), (int)(threadIdx.x % S)};
    const u64 *src = io.in + (size_t)item * io.in_stride + tile * S + cx.c;
    auto stage_unit = [&](uint32_t u) {
        if (u < (uint32_t)N_IN) {
            const uint32_t id = cv.src_id[u];
            col3_stage_twiddles<G>(twl2 + (u & 1) * G::TW, T.itw + (size_t)id * n, T.itw_sh + (size_t)id * n);
        } else if (u < (uint32_t)N_IN + n_out) {
            const uint32_t id = cv.dst_id[u - N_IN];
            col3_stage_twiddles<G>(twl2 + (u & 1) * G::TW, T.tw + (size_t)id * n, T.tw_sh + (size_t)id * n);
        }
    };
    stage_unit(0);
    __syncthreads();
    // ---- phase 1: sv[i][k] = [x_i * N^-1 * (S/s_i)^-1]_{s_i} at row t + 32 k, canonical (double / packed halves) ----
    u64 sv[N_IN][8];
    icol3_sources<G, N_IN, SRCMODE, 0>(sv, cx, src, n, r2, io, T, cv, stage_unit);
    // ---- phase 2: every target limb: conversion + forward column pass ------------------------------------------
    u64 *dst0 = io.out + (size_t)item * io.out_stride + (size_t)part * io.out_part_stride + tile * S + cx.c;
#pragma unroll 1
    for (uint32_t t = 0; t < n_out; ++t) {
        const uint32_t u = (uint32_t)N_IN + t;
        const uint32_t id = cv.dst_id[t];
        const LimbConst lc = T.limb[id];
        const ulong2 *twl = twl2 + (u & 1) * G::TW;
        u64 *dst = dst0 + (size_t)cv.dst_slot[t] * n;
        u64 x[8];
        if (lc.fp) {
            double hd[N_IN], hq[N_IN];
            uint32_t h0[N_IN], h1[N_IN];
#pragma unroll
            for (int i = 0; i < N_IN; ++i) {
                if (src_is_fp<SRCMODE>(i)) {
                    hd[i] = cv.hat_d[i * n_out + t];
                    hq[i] = cv.hatq_d[i * n_out + t];
                } else {
                    split30(cv.hat[i * n_out + t], h0[i], h1[i]);
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                double acc = 0.0;
                if (SRCMODE != 1) {  // integer-class sources: 30-bit column accumulation, < 4q < 2^53
                    Cols ia{0, 0, 0};
#pragma unroll
                    for (int i = 0; i < N_IN; ++i)
                        if (!src_is_fp<SRCMODE>(i))
                            mac_cols(ia, (uint32_t)sv[i][k], (uint32_t)(sv[i][k] >> 32), h0[i], h1[i]);
                    acc = (double)reduce_cols_lazy(ia, lc);
                }
#pragma unroll
                for (int i = 0; i < N_IN; ++i)
                    if (src_is_fp<SRCMODE>(i)) acc += fp_mulmod(bitsd(sv[i][k]), hd[i], hq[i], lc.qd);
                x[k] = dbits(fp_reduce(acc, lc.qd, lc.qinv));
            }
            col3_forward<G, true>(x, cx, twl, lc, [&] { stage_unit(u + 1); });
        } else {
            uint32_t h0[N_IN], h1[N_IN];
#pragma unroll
            for (int i = 0; i < N_IN; ++i) split30(cv.hat[i * n_out + t], h0[i], h1[i]);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                Cols ia{0, 0, 0};
#pragma unroll
                for (int i = 0; i < N_IN; ++i) {
                    uint32_t a0, a1;
                    if (src_is_fp<SRCMODE>(i)) {
                        split30_d(sv[i][k], a0, a1);
                    } else {
                        a0 = (uint32_t)sv[i][k];
                        a1 = (uint32_t)(sv[i][k] >> 32);
                    }
                    mac_cols(ia, a0, a1, h0[i], h1[i]);
                }
                x[k] = reduce_cols_lazy(ia, lc);  // < 4q: the first butterfly stage accepts < 8q
            }
            col3_forward<G, false>(x, cx, twl, lc, [&] { stage_unit(u + 1); });
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) st_pass(dst + (size_t)(8 * cx.t + k) * r2, x[k]);  // lazy u64 / doubles on an fp limb
    }
}

}  // namespace mk
