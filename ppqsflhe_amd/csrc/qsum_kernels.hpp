// qsum_kernels.hpp -- the Q-limb half of the n-client server step in ONE kernel (gfx950, fp64-class limbs).
//
// For a (ciphertext index b, Q limb t, row tile) the workgroup walks ALL clients c and, per client,
//   * finishes the forward transform (row pass) of every converted ModUp digit d_j[t] and accumulates the eval-key
//     inner products  d_j[t] * b_j[t],  d_j[t] * a_j[t]   (EvalFastKeySwitchCoreExt; the digit that owns t is c1 itself),
//   * adds c0 * P (component 0),
// then finishes the forward transform of ApproxModDown's converted limbs -- already summed over the clients in
// coefficient format (k_icol_sum + k_conv_col_psum: the conversion is linear in the INTEGER sum of the clients' canonical
// coefficients, the transform is linear) -- and subtracts them,
// keeping the two running sums over digits AND clients in registers as exact doubles; after the last client one
// multiplication by P^-1 gives  sum_c [ (ctilde_c - conv_c) * P^-1 (+ c0_c) ]  mod q_t, the coefficient-wise sum of the
// clients' re-encryptions (ReEncrypt x n at changeCipherDomain.cpp:74 + the EvalAdd chain of
// aggregateEncryptedWeights.cpp:82).  Everything between the products and the final canonical value is ring arithmetic
// mod q_t, so the stored residues are those of the reference's per-client chain, bit for bit.
//
// Against a per-client fused row pass + inner product followed by a tail + sum kernel (rounds 1-2) this removes the
// round trip of the key-switch accumulators over Q through HBM (2 L limb writes + 2 L limb reads per client ciphertext)
// and n - 1 of the n multiplications by P^-1.
#pragma once
#include "ntt_radix.hpp"

#ifndef MK_QSUM_PARK_C
#define MK_QSUM_PARK_C 1
#endif

namespace mk {

struct QSumArgs {
    const u64 *dig;    // [client][cnt][nparts][ext][N] column-passed converted digits (doubles on fp64-class limbs)
    const u64 *conv;   // [cnt][2][nl][N]  column-passed ModDown conversions SUMMED over the clients (k_conv_col_psum)
    const u64 *cts;    // input ciphertexts: client c, index i at cts + c * ct_cstride + i * ct_stride, [2][nl][N]
    const u64 *evk;    // client c at evk + c * evk_cstride: [nparts][2][D][N]
    u64 *out;          // index i at out + i * ct_stride_out: [2][nl][N]
    const u64 *pq;     // per Q limb t, 4 doubles: P mod q, (P mod q) / q, P^-1 mod q, (P^-1 mod q) / q
    size_t ct_cstride, ct_stride, evk_cstride, out_stride;
    uint32_t n_clients, cnt, nl, ext, D, alpha;
    unsigned long long slot_mask;  // fp64-class Q limbs
    uint32_t nsel;
    uint32_t init_from_out;  // continue a running sum held in `out` (client groups)
};

// Three-round row geometry (RowT<LOGC>: 8 words per thread, 256-point rows as 8 x 8 x 4, 512-point rows as 8 x 8 x 8): the
// accumulators are 32 registers, every twiddle of the limb's rows is staged once (rounds A, B in LDS, round C parked in
// LDS too: MK_QSUM_PARK_C) instead of being re-read from L2 per transform, and the kernel runs 3 waves per SIMD.  (A
// two-round form with 16 words per thread and 2 waves was 1.5 % slower and is gone.)
template <int NPARTS, int LOGC, int MINW>
__global__ __launch_bounds__(NTT_THREADS, MINW) void k_qsum3_fp(QSumArgs a, NttTables T) {
    using TL = RowT<LOGC>;
    constexpr int R = TL::R, S = TL::ROWS, TPR = TL::TPR, PAIRS = 4;
    constexpr int ND = NPARTS - 1;
    // round-C twiddles parked in LDS (MK_QSUM_PARK_C): 28 registers less across the client loop
    constexpr int NC = MK_QSUM_PARK_C ? (LOGC == 3 ? 7 : 6) : 0;
    __shared__ u64 lds[TL::WORDS + 2 * (TL::TWA + TL::TWB)];
    __shared__ ulong2 ldsc[NC ? NC * NTT_THREADS : 1];
    Row3Ctx c;
    c.lds = lds;
    c.twa = lds + TL::WORDS;
    c.twa_sh = c.twa + TL::TWA;
    c.twb = c.twa_sh + TL::TWA;
    c.twb_sh = c.twb + TL::TWB;
    c.twc = ldsc;
    const uint32_t n = 1u << T.log_n, r1 = 1u << T.log_r1;
    const uint32_t tiles = r1 / S, groups = tiles * a.nsel;
    uint32_t grp, b;
    group_member(blockIdx.x, groups, a.cnt, T.cu_affine, grp, b);
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
    if (MK_QSUM_PARK_C) {
        row3_park_c_twiddles<LOGC>(c, wc, wpc);
#pragma unroll
        for (int i = 0; i < 7; ++i) wc[i] = wpc[i] = 0;
    }
    const size_t tile_off = (size_t)row0 * R;
    const double q = lc.qd, qinv = lc.qinv;
    const double pm = bitsd(a.pq[4 * sl]), pmq = bitsd(a.pq[4 * sl + 1]);
    u64 *o0 = a.out + (size_t)b * a.out_stride + (size_t)sl * n + tile_off;
    u64 *o1 = o0 + (size_t)a.nl * n;
    double2 acc0[PAIRS], acc1[PAIRS];
#pragma unroll
    for (int i = 0; i < PAIRS; ++i) {
        if (a.init_from_out) {
            const int e = row3_pair<LOGC>(c.g, c.t, i);
            const ulong2 v0 = reinterpret_cast<const ulong2 *>(o0)[e], v1 = reinterpret_cast<const ulong2 *>(o1)[e];
            acc0[i].x = fp_mulmod(u52_to_double(v0.x), pm, pmq, q);
            acc0[i].y = fp_mulmod(u52_to_double(v0.y), pm, pmq, q);
            acc1[i].x = fp_mulmod(u52_to_double(v1.x), pm, pmq, q);
            acc1[i].y = fp_mulmod(u52_to_double(v1.y), pm, pmq, q);
        } else {
            acc0[i] = double2{0.0, 0.0};
            acc1[i] = double2{0.0, 0.0};
        }
    }
    const size_t th_off = tile_off + (size_t)c.g * R + c.t;
    auto src_of = [&](uint32_t cl, int u) -> const u64 * {
        if (cl < a.n_clients && ND > 0) {
            const size_t item = (size_t)cl * a.cnt + b;
            const int dj = u < own ? u : u + 1;
            return a.dig + ((item * NPARTS + dj) * a.ext + sl) * n + th_off;
        }
        return a.conv + (((size_t)b * 2 + u) * a.nl + sl) * n + th_off;
    };
    Stamper stm;  // diagnostic build: [0] own-digit products, [1] transforms, [2] digit products, [3] conversions + store, [4] total
    auto transform = [&](u64 (&x)[8], const u64 *next) {
        wave_lds_sync();  // previous transform's consumers finished reading this wave's rows
        row3_forward<AR_FP, LOGC, MK_QSUM_PARK_C != 0>(x, c, wc, wpc, lc);
#pragma unroll
        for (int k = 0; k < 8; ++k) lds[TL::at(c.g, 8 * c.t + k)] = dbits(fp_reduce(bitsd(x[k]), q, qinv));
        if (next) {
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = ld_stream(next + TPR * k);
        }
        wave_lds_sync();
    };
    u64 x[8];
    {
        const u64 *src = src_of(ND > 0 ? 0 : a.n_clients, 0);
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = ld_stream(src + TPR * k);
    }
    __syncthreads();  // twiddles staged
    const unsigned long long t_begin = stm.now();
    unsigned long long t_mark = t_begin;
#pragma unroll 1
    for (uint32_t cl = 0; cl < a.n_clients; ++cl) {
        const u64 *ct = a.cts + (size_t)cl * a.ct_cstride + (size_t)b * a.ct_stride + (size_t)sl * n + tile_off;
        const u64 *ek = a.evk + (size_t)cl * a.evk_cstride + (size_t)sl * n + tile_off;
        {
            const u64 *y1 = ct + (size_t)a.nl * n;
            const u64 *e0 = ek + ((size_t)own * 2 + 0) * a.D * n, *e1 = ek + ((size_t)own * 2 + 1) * a.D * n;
#pragma unroll
            for (int i = 0; i < PAIRS; ++i) {
                const int e = row3_pair<LOGC>(c.g, c.t, i);
                const ulong2 yy = ld_stream2(reinterpret_cast<const ulong2 *>(y1) + e);
                const ulong2 zz = ld_stream2(reinterpret_cast<const ulong2 *>(ct) + e);
                const ulong2 bb = reinterpret_cast<const ulong2 *>(e0)[e];
                const ulong2 aa = reinterpret_cast<const ulong2 *>(e1)[e];
                const double yx = u52_to_double(yy.x), yz = u52_to_double(yy.y);
                acc0[i].x += fp_mulmod_any(yx, u52_to_double(bb.x), q, qinv) + fp_mulmod(u52_to_double(zz.x), pm, pmq, q);
                acc0[i].y += fp_mulmod_any(yz, u52_to_double(bb.y), q, qinv) + fp_mulmod(u52_to_double(zz.y), pm, pmq, q);
                acc1[i].x += fp_mulmod_any(yx, u52_to_double(aa.x), q, qinv);
                acc1[i].y += fp_mulmod_any(yz, u52_to_double(aa.y), q, qinv);
                if (NPARTS > 4 || ND == 0) {
                    acc0[i].x = fp_reduce(acc0[i].x, q, qinv);
                    acc0[i].y = fp_reduce(acc0[i].y, q, qinv);
                    acc1[i].x = fp_reduce(acc1[i].x, q, qinv);
                    acc1[i].y = fp_reduce(acc1[i].y, q, qinv);
                }
            }
        }
        t_mark = stm.add<0>(t_mark);
        int nd = ND;
        if (ND == 1) asm volatile("" : "+s"(nd));  // two digits: keep this a loop -- inlined into the client loop it costs 50 more spilled registers
#pragma unroll 1
        for (int u = 0; u < nd; ++u) {
            const bool last_u = u == ND - 1;
            const int dj = u < own ? u : u + 1;
            const u64 *e0 = ek + ((size_t)dj * 2 + 0) * a.D * n, *e1 = ek + ((size_t)dj * 2 + 1) * a.D * n;
            // the digit's eval-key tiles in ONE burst right after its transform: loaded pair by pair at the point of use
            // (the `last_u` branch between the pairs kept the compiler from hoisting them) every pair waited for its own
            // L2 round trip -- +0.9 % on the step (before the transform instead: +0.4 %, 61 registers in scratch; the next
            // client's c1 / c0 tiles loaded a phase ahead as well: -1.0 ... -2.2 %, 74-85 registers in scratch)
            ulong2 ebb[PAIRS], eaa[PAIRS];
            transform(x, last_u ? src_of(cl + 1, 0) : src_of(cl, u + 1));
            t_mark = stm.add<1>(t_mark);
#pragma unroll
            for (int i = 0; i < PAIRS; ++i) {
                const int e = row3_pair<LOGC>(c.g, c.t, i);
                ebb[i] = reinterpret_cast<const ulong2 *>(e0)[e];
                eaa[i] = reinterpret_cast<const ulong2 *>(e1)[e];
            }
            __builtin_amdgcn_sched_barrier(0);  // the burst stays where it is written
#pragma unroll
            // per client the sums grow by at most 0.97 q (own digit) + 0.82 q (c0 P) + 0.75 q per converted digit on top
            // of the 0.51 q carried over: < 3.8 q < 2^53 for up to 4 digits (5, 6 digits: the extra reduction above);
            // the last digit's products end with the reduction that brings them back to 0.51 q
            for (int i = 0; i < PAIRS; ++i) {
                const int e = row3_pair<LOGC>(c.g, c.t, i);
                const int xx = (2 * e) % R;
                const ulong2 bb = ebb[i], aa = eaa[i];
                const double yx = bitsd(lds[TL::at(c.g, xx)]), yz = bitsd(lds[TL::at(c.g, xx + 1)]);
                acc0[i].x += fp_mulmod_any(yx, u52_to_double(bb.x), q, qinv);
                acc0[i].y += fp_mulmod_any(yz, u52_to_double(bb.y), q, qinv);
                acc1[i].x += fp_mulmod_any(yx, u52_to_double(aa.x), q, qinv);
                acc1[i].y += fp_mulmod_any(yz, u52_to_double(aa.y), q, qinv);
            }
            if (last_u) {  // one block of products, then the reductions: no branch between the pairs
#pragma unroll
                for (int i = 0; i < PAIRS; ++i) {
                    acc0[i].x = fp_reduce(acc0[i].x, q, qinv);
                    acc0[i].y = fp_reduce(acc0[i].y, q, qinv);
                    acc1[i].x = fp_reduce(acc1[i].x, q, qinv);
                    acc1[i].y = fp_reduce(acc1[i].y, q, qinv);
                }
            }
            t_mark = stm.add<2>(t_mark);
        }
    }
#pragma unroll 1
    for (int comp = 0; comp < 2; ++comp) {  // the summed ModDown conversions
        transform(x, comp == 0 ? src_of(a.n_clients, 1) : nullptr);
#pragma unroll
        for (int i = 0; i < PAIRS; ++i) {
            const int xx = (2 * row3_pair<LOGC>(c.g, c.t, i)) % R;
            const double yx = bitsd(lds[TL::at(c.g, xx)]), yz = bitsd(lds[TL::at(c.g, xx + 1)]);
            if (comp == 0) {
                acc0[i].x -= yx;
                acc0[i].y -= yz;
            } else {
                acc1[i].x -= yx;
                acc1[i].y -= yz;
            }
        }
    }
    const double pi = bitsd(a.pq[4 * sl + 2]), piq = bitsd(a.pq[4 * sl + 3]);
#pragma unroll
    for (int i = 0; i < PAIRS; ++i) {
        const int e = row3_pair<LOGC>(c.g, c.t, i);
        ulong2 r0, r1v;
        r0.x = fp_to_canonical(fp_mulmod(acc0[i].x, pi, piq, q), q, qinv);
        r0.y = fp_to_canonical(fp_mulmod(acc0[i].y, pi, piq, q), q, qinv);
        r1v.x = fp_to_canonical(fp_mulmod(acc1[i].x, pi, piq, q), q, qinv);
        r1v.y = fp_to_canonical(fp_mulmod(acc1[i].y, pi, piq, q), q, qinv);
        reinterpret_cast<ulong2 *>(o0)[e] = r0;
        reinterpret_cast<ulong2 *>(o1)[e] = r1v;
    }
    t_mark = stm.add<3>(t_mark);
    stm.add<4>(t_begin);
    stm.flush<2>(T.stamps, false);
}

}  // namespace mk
