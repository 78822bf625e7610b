// ntt_kernels.hpp -- negacyclic NTT / INTT over RNS limbs for gfx950.
//
// Stands in for DCRTPolyImpl::SwitchFormat -> ChineseRemainderTransformFTTNat::
// {ForwardTransformToBitReverse, InverseTransformFromBitReverse} ([upstream]
// transformnat-impl.h).  Same mathematical map (convention P4:
// natural-order input, bit-reversed output, psi-power table in bit-reversed order), so
// outputs are bit-identical; the schedule is MI355X-first:
//
//   N = R1 x R2.  A 2^16-point limb (512 KiB) does not fit the 160 KiB LDS of a CU, so
//   a transform is two launches, each streaming the limb once in coalesced segments:
//   - column pass: stages 0..log R1-1 couple elements R2 apart.  A workgroup owns a
//     [R1 rows][16 columns] tile (16 x 8 B = one 128-B segment per row), transforms the
//     16 columns in LDS; every column uses the same log R1 levels of twiddles.
//   - row pass: stages log R1..log N-1 act inside contiguous rows of R2 elements.  A
//     workgroup owns TILE/R2 consecutive rows (contiguous in HBM); row r uses twiddle
//     index 2^s'(R1 + r) + group at local stage s'.
//   The inverse (Gentleman-Sande) runs the same two passes in the opposite order with
//   the inverse table, then scales by N^-1 (optionally times a folded per-limb constant).
//
//   Butterflies are Harvey lazy: forward keeps values below 8q (4q in these generic kernels), inverse below 2q;
//   only the last pass reduces to the canonical [0,q).  The generic LDS-stage kernels in this file are the fallback
//   for ring sizes the register kernels of ntt_radix.hpp do not cover (odd log N).
#pragma once
#include "modarith.hpp"

namespace mk {

constexpr int NTT_THREADS = 256;
constexpr int NTT_TILE = 4096;  // elements per workgroup tile (32 KiB of LDS)
constexpr int NTT_COLS = 16;    // columns per column-pass tile: 16 x 8 B = 128-B segments

struct NttTables {
    const LimbConst *limb;  // [D]
    const u64 *tw, *tw_sh;  // [D][N] forward psi^bitrev(k) + Shoup companions
    const u64 *itw, *itw_sh;
    // round-B twiddles of the two-round row kernels, re-laid out so that one wave instruction reads contiguous memory:
    // [D][r1 rows][H chunks][H threads] pairs (w, companion); chunk i < H-1 is twiddle i of the round, chunk H-1 pads.
    // (From the plain tables a thread's 15 pairs sit up to 64 B apart per lane: 32 cache lines per wave instruction.)
    const u64 *twb, *itwb;
    uint32_t log_n, log_r1, log_r2;
    uint32_t L;  // #Q limbs at full level (P limbs start at id L)
    uint32_t has_fp;  // some limbs run on the fp64 kernel instances (host launches both instances then)
    const unsigned char *h_fp_of;  // HOST pointer (never read on the device): per limb id, 1 = fp64 instance
    uint32_t int_pm;  // HOST decision: the integer limbs run on the AR_PM instances (all of them qualify), else AR_INT
    unsigned long long *stamps;  // diagnostic builds (-DMK_STAMP=1) with MKCKKS_STAMPS=1: per-wave phase stamps; else null
    uint32_t cu_affine;  // 1: workgroups that share operand tiles are placed on the same CU (ntt_radix.hpp: group_member)
};
// arithmetic of a radix-kernel instance (template parameter AR; AR_INT / AR_FP keep the values of the old bool)
constexpr int AR_INT = 0, AR_FP = 1, AR_PM = 2;

struct NttIo {
    const u64 *in;
    u64 *out;
    size_t in_stride, out_stride;  // words between consecutive polynomials
    uint32_t in_slot0, out_slot0;  // first limb slot touched inside a polynomial
    uint32_t vslot0;               // virtual slot of the first limb (decides the limb id)
    uint32_t nslots;               // limbs per polynomial handled by this launch
    uint32_t nl;                   // #Q limbs of the polynomial (slots >= nl are P limbs)
    // digit buffers [item][part][ext][N]: polynomial p belongs to digit p % skip_nparts, whose own limbs
    // [part*skip_alpha, min(nl, (part+1)*skip_alpha)) are not transformed (0 = transform everything)
    uint32_t skip_nparts = 0, skip_alpha = 0;
    // the radix kernels come in an integer and an fp64 instance; each is launched over the slots of its class
    // only: bit b of slot_mask = slot (in_slot0 + b) belongs to this launch, nsel = popcount(slot_mask)
    unsigned long long slot_mask = ~0ull;
    uint32_t nsel = 0;
    // two-level input addressing (inverse row pass of the n-client flow, whose polynomials are the c1 of
    // cts[client][index]): polynomial p is read at in + (p / in_group) * in_gstride + (p % in_group) * in_stride;
    // in_group == 0: plain in + p * in_stride
    uint32_t in_group = 0;
    size_t in_gstride = 0;
};
MK_D size_t ntt_in_offset(const NttIo &io, uint32_t poly) {
    return io.in_group ? (size_t)(poly / io.in_group) * io.in_gstride + (size_t)(poly % io.in_group) * io.in_stride
                       : (size_t)poly * io.in_stride;
}

MK_D uint32_t nth_set_bit(unsigned long long mask, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i) mask &= mask - 1;
    return (uint32_t)__builtin_ctzll(mask);
}

MK_D bool ntt_slot_skipped(const NttIo &io, uint32_t poly, uint32_t vslot) {
    if (!io.skip_nparts) return false;
    const uint32_t lo = (poly % io.skip_nparts) * io.skip_alpha;
    const uint32_t hi = lo + io.skip_alpha < io.nl ? lo + io.skip_alpha : io.nl;
    return vslot >= lo && vslot < hi;
}

MK_D uint32_t limb_id_of(uint32_t vslot, uint32_t nl, uint32_t L) {
    return vslot < nl ? vslot : L + (vslot - nl);
}

// forward CT butterfly: x' = x + w y, y' = x - w y (lazy).  One conditional subtraction of 2q on x keeps
// every value below 8q when the inputs are below 8q (below 4q when they are below 4q); q < 2^60.
MK_D void ct_butterfly(u64 &x, u64 &y, u64 w, u64 wp, u64 q, u64 q2) {
    u64 u = csub(x, q2);
    u64 v = shoup_lazy(y, w, wp, q);
    x = u + v;
    y = u - v + q2;
}
// the two halves of the every-other-stage schedule used by the register kernels (values < 8q throughout):
// "c4": x in [0,8q) is first brought to [0,4q); outputs < 6q.   "nc": x in [0,6q) as is; outputs < 8q.
MK_D void ct_butterfly_c4(u64 &x, u64 &y, u64 w, u64 wp, u64 q, u64 q2, u64 q4) {
    u64 u = csub(x, q4);
    u64 v = shoup_lazy(y, w, wp, q);
    x = u + v;
    y = u - v + q2;
}
MK_D void ct_butterfly_nc(u64 &x, u64 &y, u64 w, u64 wp, u64 q, u64 q2) {
    u64 v = shoup_lazy(y, w, wp, q);
    u64 u = x;
    x = u + v;
    y = u - v + q2;
}
// inverse GS butterfly on (x, y) in [0,2q): x' = x + y, y' = (x - y) w (lazy)
MK_D void gs_butterfly(u64 &x, u64 &y, u64 w, u64 wp, u64 q, u64 q2) {
    u64 s = x + y;
    u64 d = x + q2 - y;
    x = csub(s, q2);
    y = shoup_lazy(d, w, wp, q);
}
// pseudo-Mersenne butterflies (LimbConst::pm, U = 2^k; bounds of pm_lazy / pm_fold in modarith.hpp).
// forward, "f": x is folded below U + 2^30 first; with y < 8U: v < 2.375U, x' < 3.376U, y' = u - v + 3q < 4.001U.
// forward, "n": x < 4.001U as is: x' < 6.376U, y' < 7.001U < 8U = the multiplier's input bound.
// A round alternates f, n, f, n: its outputs stay below 7.001U < 8q, which is what the integer consumers (canon8, the
// next round's first stage) are written for.
MK_D void ct_butterfly_pm_f(u64 &x, u64 &y, u64 w, u64 wx, const PmK &P) {
    const u64 u = pm_fold(x, P);
    const u64 v = pm_lazy(y, w, wx, P);
    x = u + v;
    y = u - v + P.q3;
}
MK_D void ct_butterfly_pm_n(u64 &x, u64 &y, u64 w, u64 wx, const PmK &P) {
    const u64 v = pm_lazy(y, w, wx, P);
    const u64 u = x;
    x = u + v;
    y = u - v + P.q3;
}
// inverse: x, y < 2.375U -> x' = fold(x + y) < 1.001U, d = x - y + 3q < 5.375U, y' = d w < 2.375U
MK_D void gs_butterfly_pm(u64 &x, u64 &y, u64 w, u64 wx, const PmK &P) {
    const u64 s = x + y;
    const u64 d = x - y + P.q3;
    x = pm_fold(s, P);
    y = pm_lazy(d, w, wx, P);
}
MK_D u64 canon8(u64 v, u64 q, u64 q2) {  // [0,8q) -> [0,q)
    return csub(csub(csub(v, q2 + q2), q2), q);
}

// Transform `nsub` independent sub-NTTs of size 2^log_r held in LDS.
// Element k of sub-transform g lives at lds[g*gs + k*ks].  Twiddle index at local stage s'
// is ((base_of(g)) << s') + group.  INV selects GS order (stages high->low).
template <bool INV, typename BaseFn>
MK_D void lds_stages(u64 *lds, int nsub, int log_r, int gs, int ks, bool sub_fast, const u64 *tw,
                     const u64 *tw_sh, u64 q, u64 q2, BaseFn base_of) {
    const int half = 1 << (log_r - 1);
    const int total = nsub * half;
    for (int st = 0; st < log_r; ++st) {
        const int s = INV ? (log_r - 1 - st) : st;
        const int log_tr = log_r - 1 - s;
        const int tr = 1 << log_tr;
        for (int b = threadIdx.x; b < total; b += NTT_THREADS) {
            int g, k;
            if (sub_fast) {  // consecutive threads walk sub-transforms (columns) first
                g = b % nsub;
                k = b / nsub;
            } else {  // consecutive threads walk butterflies of one sub-transform (row)
                g = b >> (log_r - 1);
                k = b & (half - 1);
            }
            const int grp = k >> log_tr;
            const int pos = k & (tr - 1);
            const int r0 = (grp << (log_tr + 1)) + pos;
            const uint32_t ti = ((uint32_t)base_of(g) << s) + (uint32_t)grp;
            const u64 w = tw[ti], wp = tw_sh[ti];
            u64 *p0 = lds + g * gs + r0 * ks;
            u64 *p1 = p0 + tr * ks;
            u64 x = *p0, y = *p1;
            if (INV) gs_butterfly(x, y, w, wp, q, q2);
            else ct_butterfly(x, y, w, wp, q, q2);
            *p0 = x;
            *p1 = y;
        }
        __syncthreads();
    }
}

}  // namespace mk
