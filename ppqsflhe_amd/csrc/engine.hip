// engine.hip -- gfx950 kernels + launch sequences for the PRE / aggregation hot path.
//
// Reference call sites replaced (paths relative to /root/reference; OpenFHE internals
// are [upstream]):
//   changeCipherDomain.cpp:74,89,105        cc->ReEncrypt       -> Engine::reencrypt
//   aggregateEncryptedWeights.cpp:82,91,106 cc->EvalAdd         -> Engine::eval_add / eval_sum
//   aggregateEncryptedWeights.cpp:83,92,107 cc->EvalMult(.,0.5) -> Engine::rescale (+ factors)
//   encryptModelWeights.cpp:83,91,110       cc->Encrypt         -> Engine::encrypt (+ lift_ntt)
//   decryptModelWeights.cpp:81,90,108       cc->Decrypt         -> Engine::decrypt
//   keyGen.cpp:33 / REkeyGen.cpp:52         KeyGen / ReKeyGen   -> Engine::keygen / rekeygen
//
// Data layout in HBM: limb-major u64, one limb = N contiguous words, so every kernel
// streams 8-B (or 16-B) words with unit stride per lane; the limb (modulus) is uniform per
// workgroup (blockIdx.y), so all per-limb constants sit in SGPRs.
#include "engine.hpp"
#include "comm.hpp"
#include "ntt_radix.hpp"
#include "qsum_kernels.hpp"
#include "codec_kernels.hpp"
#include "sampler_kernels.hpp"

#include <cmath>
#include <type_traits>
#include <cstdlib>
#include <cstring>

namespace mk {

// =====================================================================================
// NTT kernels (schedule described in ntt_kernels.hpp)
// =====================================================================================

template <bool INV>
__global__ __launch_bounds__(NTT_THREADS) void k_ntt_col(NttIo io, NttTables T, const u64 *scale,
                                                         const u64 *scale_sh) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const uint32_t poly = blockIdx.y / io.nslots, s = blockIdx.y % io.nslots;
    if (ntt_slot_skipped(io, poly, io.vslot0 + s)) return;  // block-uniform
    const uint32_t id = limb_id_of(io.vslot0 + s, io.nl, T.L);
    const LimbConst lc = T.limb[id];
    const uint32_t n = 1u << T.log_n, r1 = 1u << T.log_r1, r2 = 1u << T.log_r2;
    const u64 *src = io.in + (size_t)poly * io.in_stride + (size_t)(io.in_slot0 + s) * n;
    u64 *dst = io.out + (size_t)poly * io.out_stride + (size_t)(io.out_slot0 + s) * n;
    const uint32_t c0 = blockIdx.x * NTT_COLS;
    const uint32_t pairs = r1 * (NTT_COLS / 2);
    for (uint32_t e = threadIdx.x; e < pairs; e += NTT_THREADS) {
        const uint32_t r = e / (NTT_COLS / 2), cp = (e % (NTT_COLS / 2)) * 2;
        const ulong2 v = *reinterpret_cast<const ulong2 *>(src + (size_t)r * r2 + c0 + cp);
        lds[r * NTT_COLS + cp] = v.x;
        lds[r * NTT_COLS + cp + 1] = v.y;
    }
    __syncthreads();
    const u64 *tw = (INV ? T.itw : T.tw) + (size_t)id * n;
    const u64 *tw_sh = (INV ? T.itw_sh : T.tw_sh) + (size_t)id * n;
    lds_stages<INV>(lds, NTT_COLS, (int)T.log_r1, 1, NTT_COLS, true, tw, tw_sh, lc.q, lc.q2,
                    [](int) { return 1; });
    u64 sc = 0, sc_sh = 0;
    if (INV) {
        sc = scale ? scale[id] : lc.ninv;
        sc_sh = scale ? scale_sh[id] : lc.ninv_sh;
    }
    for (uint32_t e = threadIdx.x; e < pairs; e += NTT_THREADS) {
        const uint32_t r = e / (NTT_COLS / 2), cp = (e % (NTT_COLS / 2)) * 2;
        ulong2 v;
        v.x = lds[r * NTT_COLS + cp];
        v.y = lds[r * NTT_COLS + cp + 1];
        if (INV) {  // last pass of the inverse: scale by N^-1 (x folded constant), canonical
            v.x = shoup_mul(v.x, sc, sc_sh, lc.q);
            v.y = shoup_mul(v.y, sc, sc_sh, lc.q);
        }
        *reinterpret_cast<ulong2 *>(dst + (size_t)r * r2 + c0 + cp) = v;
    }
}

template <bool INV>
__global__ __launch_bounds__(NTT_THREADS) void k_ntt_row(NttIo io, NttTables T) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const uint32_t poly = blockIdx.y / io.nslots, s = blockIdx.y % io.nslots;
    if (ntt_slot_skipped(io, poly, io.vslot0 + s)) return;  // block-uniform
    const uint32_t id = limb_id_of(io.vslot0 + s, io.nl, T.L);
    const LimbConst lc = T.limb[id];
    const uint32_t n = 1u << T.log_n, r1 = 1u << T.log_r1, r2 = 1u << T.log_r2;
    const uint32_t tile = n < (uint32_t)NTT_TILE ? n : (uint32_t)NTT_TILE;
    const uint32_t rows = tile >> T.log_r2, row0 = blockIdx.x * rows;
    const u64 *src = io.in + (size_t)poly * io.in_stride + (size_t)(io.in_slot0 + s) * n + (size_t)row0 * r2;
    u64 *dst = io.out + (size_t)poly * io.out_stride + (size_t)(io.out_slot0 + s) * n + (size_t)row0 * r2;
    for (uint32_t e = threadIdx.x; e < tile / 2; e += NTT_THREADS) {
        const ulong2 v = reinterpret_cast<const ulong2 *>(src)[e];
        lds[2 * e] = v.x;
        lds[2 * e + 1] = v.y;
    }
    __syncthreads();
    const u64 *tw = (INV ? T.itw : T.tw) + (size_t)id * n;
    const u64 *tw_sh = (INV ? T.itw_sh : T.tw_sh) + (size_t)id * n;
    const int base0 = (int)(r1 + row0);
    lds_stages<INV>(lds, (int)rows, (int)T.log_r2, (int)r2, 1, false, tw, tw_sh, lc.q, lc.q2,
                    [base0](int g) { return base0 + g; });
    for (uint32_t e = threadIdx.x; e < tile / 2; e += NTT_THREADS) {
        ulong2 v;
        v.x = lds[2 * e];
        v.y = lds[2 * e + 1];
        if (!INV) {  // last pass of the forward transform: [0,8q) -> [0,q)
            v.x = canon8(v.x, lc.q, lc.q2);
            v.y = canon8(v.y, lc.q, lc.q2);
        }
        reinterpret_cast<ulong2 *>(dst)[e] = v;
    }
}

// =====================================================================================
// coefficient-wise kernels.  grid = (N / (2*256), slots, items); one modulus per block.
// =====================================================================================

constexpr int EW_THREADS = 256;

struct EwGeom {
    uint32_t n, nl, L;
};

#define EW_PROLOGUE(nslots_)                                                   \
    const uint32_t slot = blockIdx.y, item = blockIdx.z;                       \
    const uint32_t idx = (blockIdx.x * EW_THREADS + threadIdx.x) * 2;          \
    if (idx >= g.n) return;                                                    \
    (void)slot; (void)item;

__device__ __forceinline__ ulong2 ld2(const u64 *p) { return *reinterpret_cast<const ulong2 *>(p); }
__device__ __forceinline__ void st2(u64 *p, ulong2 v) { *reinterpret_cast<ulong2 *>(p) = v; }

// out = a + b over [items][slots][N]   (EvalAddCore)
__global__ void k_add(const u64 *a, const u64 *b, u64 *out, EwGeom g, const LimbConst *limb, uint32_t slots) {
    EW_PROLOGUE(slots)
    const u64 q = limb[limb_id_of(slot % g.nl, g.nl, g.L)].q;
    const size_t off = ((size_t)item * slots + slot) * g.n + idx;
    ulong2 x = ld2(a + off), y = ld2(b + off);
    x.x = add_mod(x.x, y.x, q);
    x.y = add_mod(x.y, y.y, q);
    st2(out + off, x);
}

// out[item][slot] = sum_k in[k][item][slot]  (n-ary EvalAdd; lazy u64 sum of < 2^61 terms, one reduction)
__global__ void k_sum(const u64 *in, u64 *out, EwGeom g, const LimbConst *limb, uint32_t slots,
                      uint32_t n_clients, size_t client_stride) {
    EW_PROLOGUE(slots)
    const LimbConst lc = limb[limb_id_of(slot % g.nl, g.nl, g.L)];
    const size_t off = ((size_t)item * slots + slot) * g.n + idx;
    u64 s0 = 0, s1 = 0;
    uint32_t pending = 0;
    for (uint32_t k = 0; k < n_clients; ++k) {
        const ulong2 v = ld2(in + (size_t)k * client_stride + off);
        s0 += v.x;
        s1 += v.y;
        if (++pending == 4) {  // 4 canonical terms + 1 reduced carry < 5 * 2^61 < 2^64
            s0 = reduce_word(s0, lc);
            s1 = reduce_word(s1, lc);
            pending = 1;
        }
    }
    ulong2 r;
    r.x = reduce_word(s0, lc);
    r.y = reduce_word(s1, lc);
    st2(out + off, r);
}

// in-place word-wise reduction after an integer-sum collective
__global__ void k_reduce(u64 *ct, EwGeom g, const LimbConst *limb, uint32_t slots) {
    EW_PROLOGUE(slots)
    const LimbConst lc = limb[limb_id_of(slot % g.nl, g.nl, g.L)];
    const size_t off = ((size_t)item * slots + slot) * g.n + idx;
    ulong2 v = ld2(ct + off);
    v.x = reduce_word(v.x, lc);
    v.y = reduce_word(v.y, lc);
    st2(ct + off, v);
}

// ct[item][slot] *= f[slot % nl]   (EvalMultCoreInPlace with a per-limb integer constant)
__global__ void k_mul_const(u64 *ct, EwGeom g, const LimbConst *limb, uint32_t slots, const u64 *f,
                            const u64 *f_sh) {
    EW_PROLOGUE(slots)
    const uint32_t l = slot % g.nl;
    const u64 q = limb[limb_id_of(l, g.nl, g.L)].q;
    const size_t off = ((size_t)item * slots + slot) * g.n + idx;
    ulong2 v = ld2(ct + off);
    v.x = shoup_mul(v.x, f[l], f_sh[l], q);
    v.y = shoup_mul(v.y, f[l], f_sh[l], q);
    st2(ct + off, v);
}

// NativeVectorT::SwitchModulus of the dropped limb (COEFFICIENT format) into limb `slot`:
// centred lift, v > floor(q_last/2) is negative.  last: [items][N]; out: [items][nl-1][N]
__global__ void k_switch_modulus(const u64 *last, u64 *out, EwGeom g, const LimbConst *limb, u64 q_last) {
    EW_PROLOGUE(g.nl - 1)
    const LimbConst lc = limb[slot];
    const u64 half = q_last >> 1;
    const u64 ql_mod = reduce_word(q_last, lc);
    const ulong2 v = ld2(last + (size_t)item * g.n + idx);
    ulong2 r;
    u64 a = reduce_word(v.x, lc);
    r.x = v.x > half ? sub_mod(a, ql_mod, lc.q) : a;
    a = reduce_word(v.y, lc);
    r.y = v.y > half ? sub_mod(a, ql_mod, lc.q) : a;
    st2(out + ((size_t)item * (g.nl - 1) + slot) * g.n + idx, r);
}

// DropLastElementAndScale tail: out[item][i] = (in[item][i] - tmp[item][i]) * c[i],
// c = q_last^-1 (times the EvalMult constant when the two are fused).  in has nl slots, out nl-1.
__global__ void k_sub_mul(const u64 *in, const u64 *tmp, u64 *out, EwGeom g, const LimbConst *limb,
                          const u64 *c, const u64 *c_sh) {
    EW_PROLOGUE(g.nl - 1)
    const u64 q = limb[slot].q;
    const ulong2 x = ld2(in + ((size_t)item * g.nl + slot) * g.n + idx);
    const size_t off = ((size_t)item * (g.nl - 1) + slot) * g.n + idx;
    const ulong2 t = ld2(tmp + off);
    ulong2 r;
    r.x = shoup_mul(sub_mod(x.x, t.x, q), c[slot], c_sh[slot], q);
    r.y = shoup_mul(sub_mod(x.y, t.y, q), c[slot], c_sh[slot], q);
    st2(out + off, r);
}

// ApproxModDown tail: out[item][i] = (in[item][i] - conv[item][i]) * Pinv[i] (+ add[item/2][i] on even items).
// in has ext = nl+K slots per item, conv/out have nl.
__global__ void k_moddown_tail(const u64 *in, const u64 *conv, u64 *out, EwGeom g, const LimbConst *limb,
                               uint32_t ext, const u64 *pinv, const u64 *pinv_sh, const u64 *add,
                               size_t add_stride, size_t out_item_stride, int accumulate) {
    EW_PROLOGUE(g.nl)
    const u64 q = limb[slot].q;
    const ulong2 x = ld2(in + ((size_t)item * ext + slot) * g.n + idx);
    const ulong2 t = ld2(conv + ((size_t)item * g.nl + slot) * g.n + idx);
    ulong2 r;
    r.x = shoup_mul(sub_mod(x.x, t.x, q), pinv[slot], pinv_sh[slot], q);
    r.y = shoup_mul(sub_mod(x.y, t.y, q), pinv[slot], pinv_sh[slot], q);
    if (add && (item & 1) == 0) {
        const ulong2 c = ld2(add + (size_t)(item >> 1) * add_stride + (size_t)slot * g.n + idx);
        r.x = add_mod(r.x, c.x, q);
        r.y = add_mod(r.y, c.y, q);
    }
    u64 *o = out + (size_t)item * out_item_stride + (size_t)slot * g.n + idx;
    if (accumulate) {
        const ulong2 a = ld2(o);
        r.x = add_mod(r.x, a.x, q);
        r.y = add_mod(r.y, a.y, q);
    }
    st2(o, r);
}

// copy selected limb slots between strided polynomial arrays
__global__ void k_copy_slots(const u64 *in, size_t in_stride, uint32_t in_slot0, u64 *out, size_t out_stride,
                             uint32_t out_slot0, uint32_t n) {
    const uint32_t slot = blockIdx.y, item = blockIdx.z;
    const uint32_t idx = (blockIdx.x * EW_THREADS + threadIdx.x) * 2;
    if (idx >= n) return;
    st2(out + (size_t)item * out_stride + (size_t)(out_slot0 + slot) * n + idx,
        ld2(in + (size_t)item * in_stride + (size_t)(in_slot0 + slot) * n + idx));
}

// =====================================================================================
// hybrid key switching kernels
// =====================================================================================

// ApproxSwitchCRTBasis: one thread per coefficient.  in: [items][*][N] COEFFICIENT, canonical.
template <int N_IN>
__global__ void k_baseconv(const u64 *in, size_t in_stride, u64 *out, size_t out_stride, DevConv cv,
                           const LimbConst *limb, uint32_t n, int prescaled) {
    const uint32_t idx = blockIdx.x * EW_THREADS + threadIdx.x;
    const uint32_t item = blockIdx.y;
    if (idx >= n) return;
    u64 t[N_IN];
#pragma unroll
    for (int i = 0; i < N_IN; ++i) {
        const u64 x = in[(size_t)item * in_stride + (size_t)cv.src_slot[i] * n + idx];
        // prescaled: the inverse transform already folded [(S/s_i)^-1]_{s_i} into its N^-1 scaling
        t[i] = prescaled ? x : shoup_mul(x, cv.hatinv[i], cv.hatinv_sh[i], limb[cv.src_id[i]].q);
    }
    for (uint32_t j = 0; j < cv.n_out; ++j) {
        u64 hi = 0, lo = 0;
#pragma unroll
        for (int i = 0; i < N_IN; ++i) mac128(hi, lo, t[i], cv.hat[i * cv.n_out + j]);
        out[(size_t)item * out_stride + (size_t)cv.dst_slot[j] * n + idx] = reduce_wide(hi, lo, limb[cv.dst_id[j]]);
    }
}

// EvalFastKeySwitchCoreExt, batch form: one workgroup owns a (limb, coefficient range) and keeps the eval-key
// words b_j, a_j of all NPARTS digits in registers while it walks the `items` ciphertexts of the batch, so the
// eval key is streamed from HBM once per launch instead of once per ciphertext.  The digit's own limbs are read
// straight from c1 (no copy into the digit buffer).
template <int NPARTS>
__global__ void k_inner_product_b(const u64 *digits, const u64 *c1, size_t c1_stride, const u64 *evk, u64 *ctilde,
                                  EwGeom g, const LimbConst *limb, uint32_t ext, uint32_t D, uint32_t alpha,
                                  uint32_t items, unsigned long long slot_mask) {
    const uint32_t slot = nth_set_bit(slot_mask, blockIdx.y);  // the slots this launch covers
    const uint32_t idx = (blockIdx.x * EW_THREADS + threadIdx.x) * 2;
    if (idx >= g.n) return;
    const uint32_t id = limb_id_of(slot, g.nl, g.L);
    const LimbConst lc = limb[id];
    const int own = slot < g.nl ? (int)(slot / alpha) : -1;
    ulong2 b[NPARTS], a[NPARTS];
#pragma unroll
    for (int j = 0; j < NPARTS; ++j) {
        b[j] = ld2(evk + (((size_t)j * 2 + 0) * D + id) * g.n + idx);
        a[j] = ld2(evk + (((size_t)j * 2 + 1) * D + id) * g.n + idx);
    }
    for (uint32_t item = 0; item < items; ++item) {
        u64 h0x = 0, l0x = 0, h0y = 0, l0y = 0, h1x = 0, l1x = 0, h1y = 0, l1y = 0;
#pragma unroll
        for (int j = 0; j < NPARTS; ++j) {
            const ulong2 d = (j == own) ? ld2(c1 + (size_t)item * c1_stride + (size_t)slot * g.n + idx)
                                        : ld2(digits + (((size_t)item * NPARTS + j) * ext + slot) * g.n + idx);
            mac128(h0x, l0x, d.x, b[j].x);
            mac128(h0y, l0y, d.y, b[j].y);
            mac128(h1x, l1x, d.x, a[j].x);
            mac128(h1y, l1y, d.y, a[j].y);
        }
        ulong2 r0, r1;
        if (NPARTS <= 4) {
            r0.x = reduce_sum4(h0x, l0x, lc);
            r0.y = reduce_sum4(h0y, l0y, lc);
            r1.x = reduce_sum4(h1x, l1x, lc);
            r1.y = reduce_sum4(h1y, l1y, lc);
        } else {
            r0.x = reduce_wide(h0x, l0x, lc);
            r0.y = reduce_wide(h0y, l0y, lc);
            r1.x = reduce_wide(h1x, l1x, lc);
            r1.y = reduce_wide(h1y, l1y, lc);
        }
        st2(ctilde + (((size_t)item * 2 + 0) * ext + slot) * g.n + idx, r0);
        st2(ctilde + (((size_t)item * 2 + 1) * ext + slot) * g.n + idx, r1);
    }
}

// =====================================================================================
// key generation / encryption / decryption kernels
// =====================================================================================

__device__ __forceinline__ u64 lift_value(int8_t v, const LimbConst &lc) {
    return v == 0 ? 0 : (v > 0 ? 1 : lc.q - 1);
}
__device__ __forceinline__ u64 lift_value(int32_t v, const LimbConst &lc) {
    const u64 m = reduce_word((u64)(v < 0 ? -(int64_t)v : (int64_t)v), lc);
    return (v < 0 && m != 0) ? lc.q - m : m;
}
// exact residue of round(x) for a double x (|x| < 2^120): mantissa * 2^e reduced in 128 bits
__device__ __forceinline__ u64 lift_value(double x, const LimbConst &lc) {
    const double r = round(x);
    const bool neg = r < 0;
    const double a = fabs(r);
    u64 m;
    if (a < 9223372036854775808.0) {
        m = reduce_word((u64)a, lc);
    } else {
        int e;
        const double f = frexp(a, &e);                 // a = f * 2^e, f in [0.5,1)
        const u64 mant = (u64)ldexp(f, 53);            // 53-bit integer mantissa
        const int sh = e - 53;                         // a = mant * 2^sh, 11 <= sh <= 67
        const u64 hi = sh >= 64 ? (mant << (sh - 64)) : (mant >> (64 - sh));
        const u64 lo = sh >= 64 ? 0 : (mant << sh);
        m = reduce_wide(hi, lo, lc);
    }
    return (neg && m != 0) ? lc.q - m : m;
}

// signed coefficient vectors -> residues: coef [items][N] -> out [items][ext][N]
template <typename T>
__global__ void k_lift(const T *coef, u64 *out, EwGeom g, const LimbConst *limb, uint32_t ext) {
    EW_PROLOGUE(ext)
    const LimbConst lc = limb[limb_id_of(slot, g.nl, g.L)];
    const T *c = coef + (size_t)item * g.n + idx;
    ulong2 r;
    r.x = lift_value(c[0], lc);
    r.y = lift_value(c[1], lc);
    st2(out + ((size_t)item * ext + slot) * g.n + idx, r);
}

// out = x*y + z (+ w*wc) over limb ids; every operand addressed by (base, item stride, slot->offset rule)
struct Opnd {
    const u64 *p;
    size_t item_stride;  // words between items (0 = broadcast)
    uint32_t by_id;      // 1: limb offset = limb id * N (keys over QP); 0: offset = slot * N
};
__device__ __forceinline__ ulong2 ld_op(const Opnd &o, uint32_t item, uint32_t slot, uint32_t id, uint32_t n,
                                        uint32_t idx) {
    return ld2(o.p + (size_t)item * o.item_stride + (size_t)(o.by_id ? id : slot) * n + idx);
}
__global__ void k_fma(Opnd x, Opnd y, Opnd z, Opnd w, const u64 *wc, int negate_xy, u64 *out,
                      size_t out_item_stride, uint32_t out_by_id, EwGeom g, const LimbConst *limb, uint32_t slots) {
    EW_PROLOGUE(slots)
    const uint32_t id = limb_id_of(slot, g.nl, g.L);
    const LimbConst lc = limb[id];
    const ulong2 a = ld_op(x, item, slot, id, g.n, idx), b = ld_op(y, item, slot, id, g.n, idx);
    ulong2 r;
    r.x = mul_mod(a.x, b.x, lc);
    r.y = mul_mod(a.y, b.y, lc);
    if (negate_xy) {
        r.x = r.x ? lc.q - r.x : 0;
        r.y = r.y ? lc.q - r.y : 0;
    }
    if (z.p) {
        const ulong2 c = ld_op(z, item, slot, id, g.n, idx);
        r.x = add_mod(r.x, c.x, lc.q);
        r.y = add_mod(r.y, c.y, lc.q);
    }
    if (w.p) {
        const ulong2 d = ld_op(w, item, slot, id, g.n, idx);
        const u64 k = wc ? wc[id] : 1;
        r.x = add_mod(r.x, wc ? mul_mod(d.x, k, lc) : d.x, lc.q);
        r.y = add_mod(r.y, wc ? mul_mod(d.y, k, lc) : d.y, lc.q);
    }
    st2(out + (size_t)item * out_item_stride + (size_t)(out_by_id ? id : slot) * g.n + idx, r);
}

// =====================================================================================
// Engine
// =====================================================================================

static int fast_log_h(uint32_t log_r, uint32_t other_extent);

// packed round-B twiddle table of the two-round row kernels (layout: NttTables::twb)
__global__ void k_pack_rowb(const u64 *tw, const u64 *tw_sh, u64 *out, uint32_t log_n, uint32_t log_r1, uint32_t log_h,
                            uint32_t limbs) {
    const uint32_t n = 1u << log_n, r1 = 1u << log_r1, H = 1u << log_h;
    const size_t total = (size_t)limbs * n, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one (w, wp) pair each
    if (t >= total) return;
    const uint32_t j = (uint32_t)(t % H), i = (uint32_t)((t / H) % H), row = (uint32_t)((t / ((size_t)H * H)) % r1);
    const uint32_t limb = (uint32_t)(t / n);
    u64 w = 0, wp = 0;
    if (i < H - 1) {
        const uint32_t s = 31u - (uint32_t)__clz(i + 1), gq = i - ((1u << s) - 1);
        const uint32_t idx = ((((r1 + row) << log_h) + j) << s) + gq;
        w = tw[(size_t)limb * n + idx];
        wp = tw_sh[(size_t)limb * n + idx];
    }
    out[2 * t] = w;
    out[2 * t + 1] = wp;
}


static int fast_row(uint32_t log_r2, uint32_t rows);

static dim3 ew_grid(uint32_t n, uint32_t slots, uint32_t items) {
    return dim3((n / 2 + EW_THREADS - 1) / EW_THREADS, slots, items);
}

// "no HIP device" has one non-obvious cause worth naming: two copies of the HIP runtime in one process (PyTorch-ROCm
// ships its own libamdhip64 and loads it by path; this library links /opt/rocm's).  The copy initialised second finds no
// device.  /proc/self/maps tells.
static std::string hip_runtime_diagnosis() {
    std::vector<std::string> paths;
    if (FILE *f = std::fopen("/proc/self/maps", "r")) {
        char line[4096];
        while (std::fgets(line, sizeof line, f)) {
            const char *p = std::strstr(line, "libamdhip64");
            if (!p) continue;
            const char *start = std::strchr(line, '/');
            if (!start) continue;
            std::string path(start);
            while (!path.empty() && (path.back() == '\n' || path.back() == ' ')) path.pop_back();
            bool seen = false;
            for (const std::string &q : paths) seen = seen || q == path;
            if (!seen) paths.push_back(path);
        }
        std::fclose(f);
    }
    if (paths.size() < 2) return "";
    std::string msg = "; this process carries " + std::to_string(paths.size()) + " HIP runtimes (";
    for (size_t i = 0; i < paths.size(); ++i) msg += (i ? ", " : "") + paths[i];
    return msg + "): the one initialised second sees no device -- load the library that owns the device first (import torch "
                 "before libmkckks_hip.so, or link both against the same libamdhip64)";
}

// Switches are read ONCE, when a context is created (tests and A/B runs create a context under the switch).
static bool env_flag(const char *name, bool dflt) {
    const char *e = std::getenv(name);
    return e ? std::atoi(e) != 0 : dflt;
}
Knobs Knobs::from_env() {
    Knobs k;
    if (const char *e = std::getenv("MKCKKS_CHUNK")) {
        const int v = std::atoi(e);
        if (v >= 1 && v <= 64) k.chunk = (uint32_t)v;
    }
    if (const char *e = std::getenv("MKCKKS_QSUM_GROUP")) {
        const int v = std::atoi(e);
        if (v >= 1 && v <= 64) k.qsum_group = (uint32_t)v;
    }
    k.cu_affine = env_flag("MKCKKS_CU_AFFINE", k.cu_affine);
    k.generic_ntt = env_flag("MKCKKS_GENERIC_NTT", k.generic_ntt);
    k.no_pm = env_flag("MKCKKS_NO_PM", k.no_pm);
    k.no_fp64 = env_flag("MKCKKS_NO_FP64", k.no_fp64);
    return k;
}

Engine::Engine(const ParamSet &ps, int device) : ps_(ps), device_(device), knobs_(Knobs::from_env()) {
    if (device_ < 0) return;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= device_)
        throw NoDevice("no HIP device " + std::to_string(device_) + " (found " + std::to_string(count) + ")" +
                       hip_runtime_diagnosis());
    MK_HIP(hipSetDevice(device_));
    const uint32_t D = ps_.D, n = ps_.n;
    // N = R1 x R2 with R1 <= R2; even splits of even log N land on the radix-H kernels (16/64/256)
    tabs_.log_n = ps_.log_n;
    tabs_.log_r1 = (ps_.log_n % 2 == 0) ? ((ps_.log_n / 2) & ~1u) : ps_.log_n / 2;
    tabs_.log_r2 = ps_.log_n - tabs_.log_r1;
    if (knobs_.generic_ntt) { tabs_.log_r1 = ps_.log_n / 2; tabs_.log_r2 = ps_.log_n - tabs_.log_r1; }
    // fp64 limbs: only when BOTH passes run on the radix kernels (the generic LDS-stage kernels are integer-only)
    // and the modulus is below 1.25 * 2^50 (bounds in ntt_radix.hpp).  MKCKKS_NO_FP64=1 keeps everything integer.
    const bool radix_both = fast_log_h(tabs_.log_r1, 1u << tabs_.log_r2) && fast_row(tabs_.log_r2, 1u << tabs_.log_r1);
    const bool fp_ok = radix_both && !knobs_.no_fp64;
    tabs_.has_fp = 0;
    fp_of_.assign(D, 0);
    for (uint32_t i = 0; i < D; ++i) {
        ps_.limb[i].fp = (fp_ok && ps_.moduli[i] < (5ull << 48)) ? 1u : 0u;
        fp_of_[i] = (uint8_t)ps_.limb[i].fp;
        tabs_.has_fp |= ps_.limb[i].fp;
    }
    // integer limbs of the form 2^k - c (the 60-bit q_0 and P limbs OpenFHE generates): pseudo-Mersenne butterflies, if
    // EVERY integer limb of the context qualifies and both passes run on the radix kernels (the generic LDS-stage kernels
    // are Shoup-only) -- one integer arithmetic per context, chosen at launch (AR_PM / AR_INT kernel instances).
    // MKCKKS_NO_PM=1 keeps Shoup's.
    bool all_pm = radix_both && !knobs_.no_pm;
    for (uint32_t i = 0; i < D; ++i)
        if (!ps_.limb[i].fp && !pm_eligible(ps_.moduli[i])) all_pm = false;
    tabs_.int_pm = all_pm ? 1u : 0u;
    for (uint32_t i = 0; i < D; ++i) {
        const bool pm = all_pm && !ps_.limb[i].fp;
        ps_.limb[i].pm = pm ? 1u : 0u;
        ps_.limb[i].pm_c = pm ? (uint32_t)(((u64)1 << ps_.limb[i].k) - ps_.moduli[i]) : 0u;
    }
    tabs_.h_fp_of = fp_of_.data();
    tabs_.cu_affine = knobs_.cu_affine ? 1u : 0u;
    tabs_.stamps = nullptr;
    if (MK_STAMP && env_flag("MKCKKS_STAMPS", false)) {  // diagnostic build: phase stamps of the hot kernels (tools/stamps.py)
        MK_HIP(hipMalloc(&d_stamps_, (size_t)STAMP_REGIONS * STAMP_REGION * sizeof(unsigned long long)));
        MK_HIP(hipMemset(d_stamps_, 0, (size_t)STAMP_REGIONS * STAMP_REGION * sizeof(unsigned long long)));
        tabs_.stamps = d_stamps_;
    }
    MK_HIP(hipMalloc(&d_limb_, D * sizeof(LimbConst)));
    MK_HIP(hipMemcpy(d_limb_, ps_.limb.data(), D * sizeof(LimbConst), hipMemcpyHostToDevice));
    const size_t tbytes = (size_t)D * n * sizeof(u64);
    MK_HIP(hipMalloc(&d_tw_, tbytes));
    MK_HIP(hipMalloc(&d_tw_sh_, tbytes));
    MK_HIP(hipMalloc(&d_itw_, tbytes));
    MK_HIP(hipMalloc(&d_itw_sh_, tbytes));
    std::vector<u64> w, wsh;
    auto as_fp = [&](uint32_t i) {  // (w, Shoup companion) -> (double w, double w/q) bit patterns
        const long double q = (long double)ps_.moduli[i];
        for (uint32_t k = 0; k < n; ++k) {
            const double wd = (double)w[k], wq = (double)((long double)w[k] / q);
            std::memcpy(&w[k], &wd, 8);
            std::memcpy(&wsh[k], &wq, 8);
        }
    };
    auto as_pm = [&](uint32_t i) {  // (w, Shoup companion) -> (w << t, (w * 2^32 mod q) << t), see pm_lazy
        for (uint32_t k = 0; k < n; ++k) {
            wsh[k] = pm_tw_companion(w[k], ps_.limb[i]);
            w[k] = pm_tw(w[k], ps_.limb[i]);
        }
    };
    for (uint32_t i = 0; i < D; ++i) {
        ps_.twiddles(i, false, w, wsh);
        if (ps_.limb[i].fp) as_fp(i);
        if (ps_.limb[i].pm) as_pm(i);
        MK_HIP(hipMemcpy(d_tw_ + (size_t)i * n, w.data(), n * sizeof(u64), hipMemcpyHostToDevice));
        MK_HIP(hipMemcpy(d_tw_sh_ + (size_t)i * n, wsh.data(), n * sizeof(u64), hipMemcpyHostToDevice));
        ps_.twiddles(i, true, w, wsh);
        if (ps_.limb[i].fp) as_fp(i);
        if (ps_.limb[i].pm) as_pm(i);
        MK_HIP(hipMemcpy(d_itw_ + (size_t)i * n, w.data(), n * sizeof(u64), hipMemcpyHostToDevice));
        MK_HIP(hipMemcpy(d_itw_sh_ + (size_t)i * n, wsh.data(), n * sizeof(u64), hipMemcpyHostToDevice));
    }
    {   // encoding tables: rotation group 5^j mod 2N and the 2N-th roots of unity (fp64, computed on the host)
        const uint32_t slots = n / 2, M = 2 * n;
        std::vector<uint32_t> rot(slots);
        uint64_t pw = 1;
        for (uint32_t j = 0; j < slots; ++j) { rot[j] = (uint32_t)pw; pw = pw * 5 % M; }
        std::vector<double2> ksi(M + 1);
        const double pi = std::acos(-1.0);
        for (uint32_t k = 0; k <= M; ++k) {
            const double ang = 2.0 * pi * (double)k / (double)M;
            ksi[k] = double2{std::cos(ang), std::sin(ang)};
        }
        MK_HIP(hipMalloc(&d_rot_, slots * sizeof(uint32_t)));
        MK_HIP(hipMalloc(&d_ksi_, (M + 1) * sizeof(double2)));
        MK_HIP(hipMemcpy(d_rot_, rot.data(), slots * sizeof(uint32_t), hipMemcpyHostToDevice));
        MK_HIP(hipMemcpy(d_ksi_, ksi.data(), (M + 1) * sizeof(double2), hipMemcpyHostToDevice));
    }
    tabs_.twb = tabs_.itwb = nullptr;
    if (const int lh = fast_log_h(tabs_.log_r2, 1u << tabs_.log_r1)) {  // two-round row kernels: packed round-B tables
        MK_HIP(hipMalloc(&d_twb_, 2 * tbytes));
        MK_HIP(hipMalloc(&d_itwb_, 2 * tbytes));
        const size_t total = (size_t)D * n;
        const dim3 grid((unsigned)((total + 255) / 256));
        k_pack_rowb<<<grid, 256>>>(d_tw_, d_tw_sh_, d_twb_, ps_.log_n, tabs_.log_r1, (uint32_t)lh, D);
        k_pack_rowb<<<grid, 256>>>(d_itw_, d_itw_sh_, d_itwb_, ps_.log_n, tabs_.log_r1, (uint32_t)lh, D);
        MK_HIP(hipGetLastError());
        MK_HIP(hipDeviceSynchronize());
        tabs_.twb = d_twb_;
        tabs_.itwb = d_itwb_;
    }
    tabs_.limb = d_limb_;
    tabs_.tw = d_tw_; tabs_.tw_sh = d_tw_sh_; tabs_.itw = d_itw_; tabs_.itw_sh = d_itw_sh_;
    tabs_.L = ps_.L;
}

Engine::~Engine() {
    if (device_ < 0) return;
    (void)hipSetDevice(device_);
    (void)hipDeviceSynchronize();
    for (void *p : {(void *)d_limb_, (void *)d_tw_, (void *)d_tw_sh_, (void *)d_itw_, (void *)d_itw_sh_, (void *)ws_,
                    (void *)d_twb_, (void *)d_itwb_, (void *)d_stamps_,
                    (void *)d_rot_, (void *)d_ksi_})
        if (p) (void)hipFree(p);
    for (void *p : owned_) (void)hipFree(p);
    if (up_stream_) {
        (void)hipStreamDestroy(up_stream_);
        (void)hipStreamDestroy(down_stream_);
        for (uint32_t i = 0; i < COPY_RING; ++i) (void)hipEventDestroy(copy_ev_[i]);
        (void)hipEventDestroy(ev_fence_);
    }
}

Lanes Engine::lanes() const { return Lanes{stream_}; }

void Engine::need_device() const {
    if (device_ < 0) throw NoDevice("host-only context: no device operations");
}

void Engine::sync() {
    need_device();
    MK_HIP(hipStreamSynchronize(stream_));
}
void *Engine::dev_alloc(size_t bytes) {
    need_device();
    void *p = nullptr;
    MK_HIP(hipMalloc(&p, bytes ? bytes : 8));
    return p;
}
void Engine::dev_free(void *p) {
    need_device();
    MK_HIP(hipStreamSynchronize(stream_));
    MK_HIP(hipFree(p));
}
void Engine::upload(void *d, const void *h, size_t bytes) {
    need_device();
    MK_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, stream_));
    MK_HIP(hipStreamSynchronize(stream_));
}
void Engine::download(void *h, const void *d, size_t bytes) {
    need_device();
    MK_HIP(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, stream_));
    MK_HIP(hipStreamSynchronize(stream_));
}

// ---- I/O pipeline -------------------------------------------------------------------
void Engine::copy_streams() {
    if (up_stream_) return;
    MK_HIP(hipStreamCreateWithFlags(&up_stream_, hipStreamNonBlocking));
    MK_HIP(hipStreamCreateWithFlags(&down_stream_, hipStreamNonBlocking));
    for (uint32_t i = 0; i < COPY_RING; ++i) MK_HIP(hipEventCreateWithFlags(&copy_ev_[i], hipEventDisableTiming));
    MK_HIP(hipEventCreateWithFlags(&ev_fence_, hipEventDisableTiming));
}
void *Engine::host_alloc(size_t bytes) {
    need_device();
    void *p = nullptr;
    MK_HIP(hipHostMalloc(&p, bytes ? bytes : 8, hipHostMallocDefault));
    return p;
}
void Engine::host_free(void *p) {
    need_device();
    if (p) MK_HIP(hipHostFree(p));
}
uint64_t Engine::enqueue_copy(void *dst, const void *src, size_t bytes, bool up) {
    need_device();
    copy_streams();
    const uint64_t t = next_ticket_;
    if (t > COPY_RING) {  // the ring slot's previous copy (ticket t - COPY_RING) must be over before its event is re-recorded
        MK_HIP(hipEventSynchronize(copy_ev_[t % COPY_RING]));
        done_ticket_ = std::max(done_ticket_, t - COPY_RING);
    }
    hipStream_t s = up ? up_stream_ : down_stream_;
    MK_HIP(hipMemcpyAsync(dst, src, bytes, up ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, s));
    MK_HIP(hipEventRecord(copy_ev_[t % COPY_RING], s));
    ++next_ticket_;
    return t;
}
uint64_t Engine::upload_async(void *d, const void *h, size_t bytes) { return enqueue_copy(d, h, bytes, true); }
uint64_t Engine::download_async(void *h, const void *d, size_t bytes) { return enqueue_copy(h, d, bytes, false); }
bool Engine::copy_done(uint64_t ticket) {
    need_device();
    if (ticket == 0 || ticket >= next_ticket_) throw std::invalid_argument("unknown copy ticket");
    if (ticket <= done_ticket_ || ticket + COPY_RING < next_ticket_) return true;  // slot already recycled: waited for then
    const hipError_t e = hipEventQuery(copy_ev_[ticket % COPY_RING]);
    if (e == hipErrorNotReady) return false;
    MK_HIP(e);
    return true;
}
void Engine::copy_wait(uint64_t ticket) {
    need_device();
    if (ticket == 0 || ticket >= next_ticket_) throw std::invalid_argument("unknown copy ticket");
    if (ticket <= done_ticket_ || ticket + COPY_RING < next_ticket_) return;
    MK_HIP(hipEventSynchronize(copy_ev_[ticket % COPY_RING]));
}
void Engine::fence_uploads() {
    need_device();
    if (!up_stream_) return;
    MK_HIP(hipEventRecord(ev_fence_, up_stream_));
    MK_HIP(hipStreamWaitEvent(stream_, ev_fence_, 0));
}
void Engine::fence_compute() {
    need_device();
    copy_streams();
    MK_HIP(hipEventRecord(ev_fence_, stream_));
    MK_HIP(hipStreamWaitEvent(down_stream_, ev_fence_, 0));
}

// residues at or above their modulus, counted per workgroup and added to *bad (untrusted ciphertext files: the kernels
// assume canonical residues)
__global__ void k_count_noncanonical(const u64 *ct, EwGeom g, const LimbConst *limb, uint32_t slots, unsigned long long *bad) {
    EW_PROLOGUE(slots)
    const u64 q = limb[limb_id_of(slot % g.nl, g.nl, g.L)].q;
    const ulong2 v = ld2(ct + ((size_t)item * slots + slot) * g.n + idx);
    const unsigned long long wx = __ballot(v.x >= q), wy = __ballot(v.y >= q);  // one atomic per wave, and only for bad data
    if ((wx | wy) != 0 && (threadIdx.x & 63) == 0) atomicAdd(bad, (unsigned long long)(__popcll(wx) + __popcll(wy)));
}
uint64_t Engine::count_noncanonical(const u64 *ct, uint32_t n_ct, uint32_t nl) {
    need_device();
    check_nl(nl);
    if (!n_ct) return 0;
    unsigned long long *d_bad = reinterpret_cast<unsigned long long *>(workspace(1));
    MK_HIP(hipMemsetAsync(d_bad, 0, 8, stream_));
    EwGeom g{ps_.n, nl, ps_.L};
    k_count_noncanonical<<<ew_grid(ps_.n, 2 * nl, n_ct), EW_THREADS, 0, stream_>>>(ct, g, d_limb_, 2 * nl, d_bad);
    MK_HIP(hipGetLastError());
    unsigned long long h = 0;
    MK_HIP(hipMemcpyAsync(&h, d_bad, 8, hipMemcpyDeviceToHost, stream_));
    MK_HIP(hipStreamSynchronize(stream_));
    return h;
}

size_t Engine::debug_stamps(unsigned long long *h_out, uint32_t region) {
    need_device();
    if (!d_stamps_ || region >= STAMP_REGIONS) return 0;
    MK_HIP(hipDeviceSynchronize());
    MK_HIP(hipMemcpy(h_out, d_stamps_ + (size_t)region * STAMP_REGION, (size_t)STAMP_REGION * sizeof(unsigned long long),
                     hipMemcpyDeviceToHost));
    MK_HIP(hipMemset(d_stamps_ + (size_t)region * STAMP_REGION, 0, (size_t)STAMP_REGION * sizeof(unsigned long long)));
    return STAMP_REGION;
}

void Engine::check_nl(uint32_t nl) const {
    if (nl < 1 || nl > ps_.L) throw std::invalid_argument("nl out of range");
}

u64 *Engine::workspace(size_t words) {
    if (words > ws_words_) {
        MK_HIP(hipStreamSynchronize(stream_));
        if (ws_) MK_HIP(hipFree(ws_));
        ws_ = nullptr;
        ws_words_ = 0;
        MK_HIP(hipMalloc(&ws_, words * sizeof(u64)));
        ws_words_ = words;
    }
    return ws_;
}

const u64 *Engine::limb_vector(const std::string &key, const std::vector<u64> &vals) {
    auto it = vec_cache_.find(key);
    if (it != vec_cache_.end()) return it->second;
    u64 *d = nullptr;
    MK_HIP(hipMalloc(&d, vals.size() * sizeof(u64)));
    MK_HIP(hipMemcpy(d, vals.data(), vals.size() * sizeof(u64), hipMemcpyHostToDevice));
    owned_.push_back(d);
    vec_cache_[key] = d;
    return d;
}

static DevConv to_dev(const BaseConvTable &t, const u64 *d_hat, const u64 *d_hat_fp) {
    if (t.src.size() > (size_t)MAX_CONV_IN || t.dst.size() > (size_t)MAX_CONV_OUT)
        throw std::invalid_argument("base conversion larger than supported");
    DevConv c{};
    c.n_in = (uint32_t)t.src.size();
    c.n_out = (uint32_t)t.dst.size();
    for (size_t i = 0; i < t.src.size(); ++i) {
        c.src_id[i] = t.src[i];
        c.hatinv[i] = t.hatinv[i];
        c.hatinv_sh[i] = t.hatinv_sh[i];
    }
    for (size_t j = 0; j < t.dst.size(); ++j) c.dst_id[j] = t.dst[j];
    c.hat = d_hat;
    c.hat_d = reinterpret_cast<const double *>(d_hat_fp);
    c.hatq_d = c.hat_d + t.hat.size();
    return c;
}

// [S/s_i]_t and [S/s_i]_t / t as doubles (bit patterns), [n_in][n_out] each: the fp64 form of the conversion matrix
static std::vector<u64> hat_as_doubles(const BaseConvTable &t, const std::vector<u64> &moduli) {
    const size_t cnt = t.hat.size(), n_out = t.dst.size();
    std::vector<u64> v(2 * cnt);
    for (size_t e = 0; e < cnt; ++e) {
        const long double q = (long double)moduli[t.dst[e % n_out]];
        const double h = (double)t.hat[e], hq = (double)((long double)t.hat[e] / q);
        std::memcpy(&v[e], &h, 8);
        std::memcpy(&v[cnt + e], &hq, 8);
    }
    return v;
}

const DevConv &Engine::modup_conv(uint32_t nl, uint32_t part) {
    auto key = std::make_pair(nl, part);
    auto it = modup_cache_.find(key);
    if (it != modup_cache_.end()) return it->second;
    BaseConvTable t = ps_.modup_table(nl, part);
    const std::string tag = std::to_string(nl) + "_" + std::to_string(part);
    const u64 *d_hat = limb_vector("modup_hat_" + tag, t.hat);
    DevConv c = to_dev(t, d_hat, limb_vector("modup_hatd_" + tag, hat_as_doubles(t, ps_.moduli)));
    const uint32_t lo = part * ps_.alpha;
    for (uint32_t i = 0; i < c.n_in; ++i) c.src_slot[i] = lo + i;  // slot inside the nl-limb input polynomial
    // targets: all slots of the extended polynomial except the digit's own
    uint32_t w = 0;
    for (uint32_t s = 0; s < nl + ps_.K; ++s)
        if (s < lo || s >= lo + c.n_in) c.dst_slot[w++] = s;
    return modup_cache_.emplace(key, c).first->second;
}

const DevConv &Engine::moddown_conv(uint32_t nl) {
    auto it = moddown_cache_.find(nl);
    if (it != moddown_cache_.end()) return it->second;
    BaseConvTable t = ps_.moddown_table(nl);
    const u64 *d_hat = limb_vector("moddown_hat_" + std::to_string(nl), t.hat);
    DevConv c = to_dev(t, d_hat, limb_vector("moddown_hatd_" + std::to_string(nl), hat_as_doubles(t, ps_.moduli)));
    for (uint32_t k = 0; k < c.n_in; ++k) c.src_slot[k] = k;
    for (uint32_t i = 0; i < c.n_out; ++i) c.dst_slot[i] = i;
    return moddown_cache_.emplace(nl, c).first->second;
}

// source-class pattern of a conversion for k_conv_col: 0 = all integer, 1 = all fp64, 2 = source 0 integer and
// the rest fp64; -1 = none of these (the caller then interchanges every source as packed 30-bit halves)
int Engine::conv_src_mode(const DevConv &cv) const {
    bool all_fp = true, all_int = true, rest_fp = true;
    for (uint32_t i = 0; i < cv.n_in; ++i) {
        const bool fp = fp_of_[cv.src_id[i]] != 0;
        all_fp = all_fp && fp;
        all_int = all_int && !fp;
        if (i > 0) rest_fp = rest_fp && fp;
    }
    if (all_int) return 0;
    if (all_fp) return 1;
    return (!fp_of_[cv.src_id[0]] && rest_fp) ? 2 : -1;
}

template <int N_IN>
static void launch_baseconv_n(const u64 *in, size_t in_stride, u64 *out, size_t out_stride, const DevConv &cv,
                              const LimbConst *limb, uint32_t n, uint32_t items, int prescaled, hipStream_t s) {
    dim3 grid((n + EW_THREADS - 1) / EW_THREADS, items);
    k_baseconv<N_IN><<<grid, EW_THREADS, 0, s>>>(in, in_stride, out, out_stride, cv, limb, n, prescaled);
}
static void launch_baseconv(const u64 *in, size_t in_stride, u64 *out, size_t out_stride, const DevConv &cv,
                            const LimbConst *limb, uint32_t n, uint32_t items, int prescaled, hipStream_t s) {
    switch (cv.n_in) {
        case 1: launch_baseconv_n<1>(in, in_stride, out, out_stride, cv, limb, n, items, prescaled, s); break;
        case 2: launch_baseconv_n<2>(in, in_stride, out, out_stride, cv, limb, n, items, prescaled, s); break;
        case 3: launch_baseconv_n<3>(in, in_stride, out, out_stride, cv, limb, n, items, prescaled, s); break;
        case 4: launch_baseconv_n<4>(in, in_stride, out, out_stride, cv, limb, n, items, prescaled, s); break;
        case 5: launch_baseconv_n<5>(in, in_stride, out, out_stride, cv, limb, n, items, prescaled, s); break;
        case 6: launch_baseconv_n<6>(in, in_stride, out, out_stride, cv, limb, n, items, prescaled, s); break;
        case 7: launch_baseconv_n<7>(in, in_stride, out, out_stride, cv, limb, n, items, prescaled, s); break;
        case 8: launch_baseconv_n<8>(in, in_stride, out, out_stride, cv, limb, n, items, prescaled, s); break;
        default: throw std::invalid_argument("base conversion fan-in unsupported");
    }
    MK_HIP(hipGetLastError());
}

// ---- transforms ---------------------------------------------------------------------

// radix-H register kernels exist for sub-transform sizes 16, 64, 256 (H = 4, 8, 16) and need at least
// S = 256/H columns (rows) for the column (row) pass; everything else takes the generic LDS-stage kernels.
static int fast_log_h(uint32_t log_r, uint32_t other_extent) {
    if (log_r != 4 && log_r != 6 && log_r != 8) return 0;
    const uint32_t h = 1u << (log_r / 2);
    return (256u / h) <= other_extent ? (int)(log_r / 2) : 0;
}
// row pass only: 512-point rows take the three-round radix-8 kernel (code 9)
static int fast_row(uint32_t log_r2, uint32_t rows) {
    if (log_r2 == 9 && rows >= 4) return 9;
    return fast_log_h(log_r2, rows);
}

// The integer and the fp64 instance of a pass touch disjoint limbs and are launched one after the other on the same
// stream (forking them onto two streams was measured slower: -5 %, event round trips).
template <typename FInt, typename FFp>
static void launch_two_classes(const Lanes &ln, bool has_int, bool has_fp, FInt &&launch_int, FFp &&launch_fp) {
    if (has_int) launch_int(ln.main);
    if (has_fp) launch_fp(ln.main);
}

// slots of `io` whose limb runs on the fp64 (want_fp) or the integer instance; fp_of: per-limb-id class (host copy)
// the integer instance of a radix kernel in the context's integer arithmetic: f(integral_constant<int, AR_PM | AR_INT>)
template <typename F>
static void with_int_arith(const NttTables &T, F &&f) {
    if (T.int_pm) f(std::integral_constant<int, AR_PM>{});
    else f(std::integral_constant<int, AR_INT>{});
}

static unsigned long long class_mask(const NttIo &io, const unsigned char *fp_of, uint32_t L, bool want_fp) {
    unsigned long long m = 0;
    for (uint32_t b = 0; b < io.nslots; ++b) {
        const uint32_t v = io.vslot0 + b, id = v < io.nl ? v : L + (v - io.nl);
        if ((fp_of[id] != 0) == want_fp) m |= 1ull << b;
    }
    return m;
}

template <bool INV>
static void launch_col(const NttIo &io0, const NttTables &T, uint32_t n_polys, const u64 *scale, const u64 *scale_sh,
                       const Lanes &ln, int pack = 0) {
    const uint32_t r1 = 1u << T.log_r1, r2 = 1u << T.log_r2;
    NttIo io = io0, iof = io0;
    io.slot_mask = class_mask(io0, T.h_fp_of, T.L, false);
    iof.slot_mask = class_mask(io0, T.h_fp_of, T.L, true);
    io.nsel = (uint32_t)__builtin_popcountll(io.slot_mask);
    iof.nsel = (uint32_t)__builtin_popcountll(iof.slot_mask);
    const uint32_t items = n_polys * io.nsel, itemsf = n_polys * iof.nsel;
    switch (fast_log_h(T.log_r1, r2)) {
        case 4:
            launch_two_classes(ln, items != 0, itemsf != 0,
                [&](hipStream_t s) { with_int_arith(T, [&](auto ar) {
                    k_ntt_col_r<4, INV, decltype(ar)::value>
                        <<<dim3(r2 / 16, items), NTT_THREADS, 0, s>>>(io, T, scale, scale_sh, pack);
                }); },
                [&](hipStream_t s) { k_ntt_col_r<4, INV, AR_FP><<<dim3(r2 / 16, itemsf), NTT_THREADS, 0, s>>>(iof, T, scale, scale_sh, pack); });
            break;
        case 3:
            launch_two_classes(ln, items != 0, itemsf != 0,
                [&](hipStream_t s) { with_int_arith(T, [&](auto ar) {
                    k_ntt_col_r<3, INV, decltype(ar)::value>
                        <<<dim3(r2 / 32, items), NTT_THREADS, 0, s>>>(io, T, scale, scale_sh, pack);
                }); },
                [&](hipStream_t s) { k_ntt_col_r<3, INV, AR_FP><<<dim3(r2 / 32, itemsf), NTT_THREADS, 0, s>>>(iof, T, scale, scale_sh, pack); });
            break;
        case 2:
            launch_two_classes(ln, items != 0, itemsf != 0,
                [&](hipStream_t s) { with_int_arith(T, [&](auto ar) {
                    k_ntt_col_r<2, INV, decltype(ar)::value>
                        <<<dim3(r2 / 64, items), NTT_THREADS, 0, s>>>(io, T, scale, scale_sh, pack);
                }); },
                [&](hipStream_t s) { k_ntt_col_r<2, INV, AR_FP><<<dim3(r2 / 64, itemsf), NTT_THREADS, 0, s>>>(iof, T, scale, scale_sh, pack); });
            break;
        default:
            if (pack) throw std::logic_error("packed output needs the radix column kernel");
            k_ntt_col<INV><<<dim3(r2 / NTT_COLS, n_polys * io0.nslots), NTT_THREADS, (size_t)r1 * NTT_COLS * sizeof(u64), ln.main>>>(
                io0, T, scale, scale_sh);
    }
}

static bool row_tail_supported(const NttTables &T) { return fast_row(T.log_r2, 1u << T.log_r1) != 0; }

template <bool INV>
static void launch_row(const NttIo &io0, const NttTables &T, uint32_t n_polys, const TailArgs &tail, const Lanes &ln,
                       unsigned classes = 3) {  // bit 0: integer limbs, bit 1: fp64 limbs
    const uint32_t n = 1u << T.log_n, r1 = 1u << T.log_r1;
    NttIo io = io0, iof = io0;
    io.slot_mask = class_mask(io0, T.h_fp_of, T.L, false);
    iof.slot_mask = class_mask(io0, T.h_fp_of, T.L, true);
    io.nsel = (uint32_t)__builtin_popcountll(io.slot_mask);
    iof.nsel = (uint32_t)__builtin_popcountll(iof.slot_mask);
    const uint32_t items = (classes & 1) ? n_polys * io.nsel : 0, itemsf = (classes & 2) ? n_polys * iof.nsel : 0;
    switch (fast_row(T.log_r2, r1)) {
        case 9:
            launch_two_classes(ln, items != 0, itemsf != 0,
                [&](hipStream_t s) { with_int_arith(T, [&](auto ar) {
                    k_ntt_row3<INV, decltype(ar)::value>
                        <<<dim3((r1 / 4) * items), NTT_THREADS, 0, s>>>(io, T, tail);
                }); },
                [&](hipStream_t s) { k_ntt_row3<INV, AR_FP><<<dim3((r1 / 4) * itemsf), NTT_THREADS, 0, s>>>(iof, T, tail); });
            break;
        case 4:
            launch_two_classes(ln, items != 0, itemsf != 0,
                [&](hipStream_t s) { with_int_arith(T, [&](auto ar) {
                    k_ntt_row_r<4, INV, decltype(ar)::value>
                        <<<dim3((r1 / 16) * items), NTT_THREADS, 0, s>>>(io, T, tail);
                }); },
                [&](hipStream_t s) { k_ntt_row_r<4, INV, AR_FP><<<dim3((r1 / 16) * itemsf), NTT_THREADS, 0, s>>>(iof, T, tail); });
            break;
        case 3:
            launch_two_classes(ln, items != 0, itemsf != 0,
                [&](hipStream_t s) { with_int_arith(T, [&](auto ar) {
                    k_ntt_row_r<3, INV, decltype(ar)::value>
                        <<<dim3((r1 / 32) * items), NTT_THREADS, 0, s>>>(io, T, tail);
                }); },
                [&](hipStream_t s) { k_ntt_row_r<3, INV, AR_FP><<<dim3((r1 / 32) * itemsf), NTT_THREADS, 0, s>>>(iof, T, tail); });
            break;
        case 2:
            launch_two_classes(ln, items != 0, itemsf != 0,
                [&](hipStream_t s) { with_int_arith(T, [&](auto ar) {
                    k_ntt_row_r<2, INV, decltype(ar)::value>
                        <<<dim3((r1 / 64) * items), NTT_THREADS, 0, s>>>(io, T, tail);
                }); },
                [&](hipStream_t s) { k_ntt_row_r<2, INV, AR_FP><<<dim3((r1 / 64) * itemsf), NTT_THREADS, 0, s>>>(iof, T, tail); });
            break;
        default: {
            if (tail.enabled) throw std::logic_error("fused tail needs the radix row kernel");
            const uint32_t tile = n < (uint32_t)NTT_TILE ? n : (uint32_t)NTT_TILE;
            k_ntt_row<INV><<<dim3(n / tile, n_polys * io0.nslots), NTT_THREADS, (size_t)tile * sizeof(u64), ln.main>>>(io0, T);
        }
    }
}

// two-pass launcher: reads `io.in`, leaves the result in `io.out` (may be the same buffer)
static void ntt_passes(NttIo io, const NttTables &T, uint32_t n_polys, bool inverse, const u64 *scale,
                       const u64 *scale_sh, const Lanes &s, int pack = 0) {
    if (n_polys == 0 || io.nslots == 0) return;
    NttIo second = io;  // second pass runs in place on the output
    second.in_group = 0;
    second.in = io.out;
    second.in_stride = io.out_stride;
    second.in_slot0 = io.out_slot0;
    const TailArgs none{};
    if (!inverse) {
        launch_col<false>(io, T, n_polys, nullptr, nullptr, s);
        launch_row<false>(second, T, n_polys, none, s);
    } else {
        launch_row<true>(io, T, n_polys, none, s);
        launch_col<true>(second, T, n_polys, scale, scale_sh, s, pack);
    }
    MK_HIP(hipGetLastError());
}

// base conversion fused into the forward column pass of every converted limb (k_conv_col); false when the
// column pass of this ring size has no radix kernel (the caller then runs k_baseconv + a plain column pass)
template <int LOG_H, int N_IN, int SRCMODE>
static void launch_conv_col_n(const ConvIo &io, const ConvIo &iof, const dim3 &grid, const dim3 &gridf, const NttTables &T,
                              const DevConv &cv, const Lanes &ln) {
    if constexpr (LOG_H >= 3) {
        // two targets of a class per workgroup (k_conv_col2): half the source traffic through each CU's L1; the last
        // workgroup of an odd target count carries one
        const uint32_t per = grid.x / (io.nsel ? io.nsel : 1), perf = gridf.x / (iof.nsel ? iof.nsel : 1);
        if (io.nsel) with_int_arith(T, [&](auto ar) {
            k_conv_col2<LOG_H, N_IN, decltype(ar)::value, DevConv, SRCMODE>
                <<<dim3(per * ((io.nsel + 1) / 2)), NTT_THREADS, 0, ln.main>>>(io, T, cv);
        });
        if (iof.nsel)
            k_conv_col2<LOG_H, N_IN, AR_FP, DevConv, SRCMODE><<<dim3(perf * ((iof.nsel + 1) / 2)), NTT_THREADS, 0, ln.main>>>(iof, T, cv);
    } else {
        launch_two_classes(ln, io.nsel != 0, iof.nsel != 0,
            [&](hipStream_t s) { with_int_arith(T, [&](auto ar) {
                k_conv_col<LOG_H, N_IN, decltype(ar)::value, DevConv, SRCMODE>
                    <<<grid, NTT_THREADS, 0, s>>>(io, T, cv);
            }); },
            [&](hipStream_t s) { k_conv_col<LOG_H, N_IN, AR_FP, DevConv, SRCMODE><<<gridf, NTT_THREADS, 0, s>>>(iof, T, cv); });
    }
}
template <int LOG_H, int N_IN>
static void launch_conv_col_m(const ConvIo &io, const ConvIo &iof, const dim3 &grid, const dim3 &gridf, const NttTables &T,
                              const DevConv &cv, const Lanes &ln, int srcmode) {
    if constexpr (N_IN <= 4 && LOG_H >= 3) {
        if (srcmode == 1) return launch_conv_col_n<LOG_H, N_IN, 1>(io, iof, grid, gridf, T, cv, ln);
        if (srcmode == 2) return launch_conv_col_n<LOG_H, N_IN, (N_IN > 1 ? 2 : 0)>(io, iof, grid, gridf, T, cv, ln);
    }
    if (srcmode != 0) throw std::logic_error("double conversion sources: unsupported shape");
    launch_conv_col_n<LOG_H, N_IN, 0>(io, iof, grid, gridf, T, cv, ln);
}
// srcmode: form of the sources in io.in (see src_is_double): 0 = packed 30-bit halves, 1 / 2 = doubles
template <int LOG_H>
static void launch_conv_col_h(const ConvIo &io0, const NttTables &T, const DevConv &cv, const Lanes &ln, int srcmode,
                              unsigned classes) {  // classes: bit 0 integer-class targets, bit 1 fp64-class targets
    const uint32_t tiles = (1u << T.log_r2) / (256u >> LOG_H);
    ConvIo io = io0, iof = io0;
    io.target_mask = iof.target_mask = 0;
    for (uint32_t j = 0; j < cv.n_out; ++j) {
        const bool fp = T.h_fp_of[cv.dst_id[j]] != 0;
        if (fp && (classes & 2)) iof.target_mask |= 1ull << j;
        if (!fp && (classes & 1)) io.target_mask |= 1ull << j;
    }
    io.nsel = (uint32_t)__builtin_popcountll(io.target_mask);
    iof.nsel = (uint32_t)__builtin_popcountll(iof.target_mask);
    const dim3 grid(io.items * tiles * io.nsel), gridf(io.items * tiles * iof.nsel);
    switch (cv.n_in) {
        case 1: launch_conv_col_m<LOG_H, 1>(io, iof, grid, gridf, T, cv, ln, srcmode); break;
        case 2: launch_conv_col_m<LOG_H, 2>(io, iof, grid, gridf, T, cv, ln, srcmode); break;
        case 3: launch_conv_col_m<LOG_H, 3>(io, iof, grid, gridf, T, cv, ln, srcmode); break;
        case 4: launch_conv_col_m<LOG_H, 4>(io, iof, grid, gridf, T, cv, ln, srcmode); break;
        case 5: launch_conv_col_m<LOG_H, 5>(io, iof, grid, gridf, T, cv, ln, srcmode); break;
        case 6: launch_conv_col_m<LOG_H, 6>(io, iof, grid, gridf, T, cv, ln, srcmode); break;
        case 7: launch_conv_col_m<LOG_H, 7>(io, iof, grid, gridf, T, cv, ln, srcmode); break;
        case 8: launch_conv_col_m<LOG_H, 8>(io, iof, grid, gridf, T, cv, ln, srcmode); break;
        default: throw std::invalid_argument("base conversion fan-in unsupported");
    }
}
// the group's ONE ModDown conversion (k_conv_col_psum), one instance per arithmetic class of the targets
template <int LOG_H, int N_IN>
static void launch_conv_col_psum_n(const ConvIo &io, const ConvIo &iof, uint32_t tiles, const NttTables &T, const DevConv &cv,
                                   hipStream_t s) {
    if (io.nsel) with_int_arith(T, [&](auto ar) {
        k_conv_col_psum<LOG_H, N_IN, decltype(ar)::value, DevConv>
            <<<dim3(io.items * tiles * io.nsel), NTT_THREADS, 0, s>>>(io, T, cv);
    });
    if (iof.nsel)  // fp64-class targets two per workgroup (+0.2 % on the step at C3, +0.9 % at N = 2^17, L = 20)
        k_conv_col_psum2<LOG_H, N_IN, DevConv><<<dim3(iof.items * tiles * ((iof.nsel + 1) / 2)), NTT_THREADS, 0, s>>>(iof, T, cv);
}
template <int LOG_H>
static void launch_conv_col_psum_h(const ConvIo &io0, const NttTables &T, const DevConv &cv, hipStream_t s) {
    const uint32_t tiles = (1u << T.log_r2) / (256u >> LOG_H);
    ConvIo io = io0, iof = io0;
    io.target_mask = iof.target_mask = 0;
    for (uint32_t j = 0; j < cv.n_out; ++j) (T.h_fp_of[cv.dst_id[j]] ? iof.target_mask : io.target_mask) |= 1ull << j;
    io.nsel = (uint32_t)__builtin_popcountll(io.target_mask);
    iof.nsel = (uint32_t)__builtin_popcountll(iof.target_mask);
    switch (cv.n_in) {
        case 1: launch_conv_col_psum_n<LOG_H, 1>(io, iof, tiles, T, cv, s); break;
        case 2: launch_conv_col_psum_n<LOG_H, 2>(io, iof, tiles, T, cv, s); break;
        case 3: launch_conv_col_psum_n<LOG_H, 3>(io, iof, tiles, T, cv, s); break;
        case 4: launch_conv_col_psum_n<LOG_H, 4>(io, iof, tiles, T, cv, s); break;
        case 5: launch_conv_col_psum_n<LOG_H, 5>(io, iof, tiles, T, cv, s); break;
        case 6: launch_conv_col_psum_n<LOG_H, 6>(io, iof, tiles, T, cv, s); break;
        case 7: launch_conv_col_psum_n<LOG_H, 7>(io, iof, tiles, T, cv, s); break;
        case 8: launch_conv_col_psum_n<LOG_H, 8>(io, iof, tiles, T, cv, s); break;
        default: throw std::invalid_argument("base conversion fan-in unsupported");
    }
}
static void launch_conv_col_psum(const ConvIo &io, const NttTables &T, const DevConv &cv, hipStream_t s) {
    switch (fast_log_h(T.log_r1, 1u << T.log_r2)) {
        case 4: launch_conv_col_psum_h<4>(io, T, cv, s); break;
        case 3: launch_conv_col_psum_h<3>(io, T, cv, s); break;
        default: throw std::logic_error("summed conversion needs 64- or 256-point columns");
    }
    MK_HIP(hipGetLastError());
}
// inverse column pass of the P limbs of every client of a group, summed over the clients as integers (k_icol_sum)
static void launch_icol_sum(const u64 *pc, u64 *psum, const NttTables &T, const u64 *scale, const u64 *scale_sh, uint32_t K,
                            uint32_t n_polys, uint32_t n_clients, size_t in_cstride, hipStream_t s) {
    const uint32_t r2 = 1u << T.log_r2;
    switch (fast_log_h(T.log_r1, r2)) {
        case 4: with_int_arith(T, [&](auto ar) {
            k_icol_sum<4, decltype(ar)::value>
                <<<dim3(r2 / 16, n_polys * K), NTT_THREADS, 0, s>>>(pc, psum, T, scale, scale_sh, K, n_clients, in_cstride);
        }); break;
        case 3: with_int_arith(T, [&](auto ar) {
            k_icol_sum<3, decltype(ar)::value>
                <<<dim3(r2 / 32, n_polys * K), NTT_THREADS, 0, s>>>(pc, psum, T, scale, scale_sh, K, n_clients, in_cstride);
        }); break;
        default: throw std::logic_error("summed conversion needs 64- or 256-point columns");
    }
    MK_HIP(hipGetLastError());
}
static bool launch_conv_col(const ConvIo &io, const NttTables &T, const DevConv &cv, const Lanes &s, int srcmode = 0,
                            unsigned classes = 3) {
    switch (fast_log_h(T.log_r1, 1u << T.log_r2)) {
        case 4: launch_conv_col_h<4>(io, T, cv, s, srcmode, classes); break;
        case 3: launch_conv_col_h<3>(io, T, cv, s, srcmode, classes); break;
        case 2: launch_conv_col_h<2>(io, T, cv, s, srcmode, classes); break;
        default: return false;
    }
    MK_HIP(hipGetLastError());
    return true;
}

void Engine::ntt_launch(u64 *d, uint32_t n_polys, uint32_t nl, uint32_t ext, bool inverse, const u64 *scale,
                        const u64 *scale_sh) {
    NttIo io{d, d, (size_t)ext * ps_.n, (size_t)ext * ps_.n, 0, 0, 0, ext, nl};
    ntt_passes(io, tabs_, n_polys, inverse, scale, scale_sh, lanes());
}

void Engine::ntt_forward(u64 *d, uint32_t n_polys, uint32_t nl, bool with_p) {
    need_device();
    if (nl > ps_.L || (nl == 0 && !with_p)) throw std::invalid_argument("nl out of range");
    ntt_launch(d, n_polys, nl, nl + (with_p ? ps_.K : 0), false, nullptr, nullptr);
}
void Engine::ntt_inverse(u64 *d, uint32_t n_polys, uint32_t nl, bool with_p) {
    need_device();
    if (nl > ps_.L || (nl == 0 && !with_p)) throw std::invalid_argument("nl out of range");
    ntt_launch(d, n_polys, nl, nl + (with_p ? ps_.K : 0), true, nullptr, nullptr);
}

// ---- aggregation --------------------------------------------------------------------

void Engine::eval_add(const u64 *a, const u64 *b, u64 *out, uint32_t n_ct, uint32_t nl) {
    need_device();
    check_nl(nl);
    if (!n_ct) return;
    EwGeom g{ps_.n, nl, ps_.L};
    k_add<<<ew_grid(ps_.n, 2 * nl, n_ct), EW_THREADS, 0, stream_>>>(a, b, out, g, d_limb_, 2 * nl);
    MK_HIP(hipGetLastError());
}

void Engine::eval_sum(const u64 *in, u64 *out, uint32_t n_clients, uint32_t n_ct, uint32_t nl) {
    need_device();
    check_nl(nl);
    if (!n_ct || !n_clients) return;
    EwGeom g{ps_.n, nl, ps_.L};
    k_sum<<<ew_grid(ps_.n, 2 * nl, n_ct), EW_THREADS, 0, stream_>>>(in, out, g, d_limb_, 2 * nl, n_clients,
                                                                 (size_t)n_ct * 2 * nl * ps_.n);
    MK_HIP(hipGetLastError());
}

void Engine::reduce_mod(u64 *ct, uint32_t n_ct, uint32_t nl, uint32_t n_terms) {
    need_device();
    check_nl(nl);
    if (n_terms > 8) throw std::invalid_argument("integer-sum collective supports at most 8 terms (q < 2^61)");
    if (!n_ct) return;
    EwGeom g{ps_.n, nl, ps_.L};
    k_reduce<<<ew_grid(ps_.n, 2 * nl, n_ct), EW_THREADS, 0, stream_>>>(ct, g, d_limb_, 2 * nl);
    MK_HIP(hipGetLastError());
}

void Engine::reduce_scatter_sum_mod(void *comm, const u64 *partial, u64 *shard, uint32_t n_ct_shard, uint32_t nl,
                                    uint32_t n_ranks) {
    need_device();
    check_nl(nl);
    if (n_ranks < 1 || n_ranks > 8) throw std::invalid_argument("integer-sum collective supports 1..8 ranks (q < 2^61)");
    if (!n_ct_shard) return;
    comm_reduce_scatter_u64(comm, partial, shard, (size_t)n_ct_shard * 2 * nl * ps_.n, stream_);
    reduce_mod(shard, n_ct_shard, nl, n_ranks);
}

void Engine::mult_const(u64 *ct, uint32_t n_ct, uint32_t nl, const std::vector<u64> &factors) {
    need_device();
    check_nl(nl);
    if (!n_ct) return;
    std::vector<u64> both(factors);
    for (uint32_t i = 0; i < nl; ++i) both.push_back(h_shoup(factors[i], ps_.moduli[i]));
    u64 *d_f = workspace(2 * nl);
    MK_HIP(hipMemcpyAsync(d_f, both.data(), both.size() * sizeof(u64), hipMemcpyHostToDevice, stream_));
    MK_HIP(hipStreamSynchronize(stream_));  // `both` is a stack-local staging buffer
    EwGeom g{ps_.n, nl, ps_.L};
    k_mul_const<<<ew_grid(ps_.n, 2 * nl, n_ct), EW_THREADS, 0, stream_>>>(ct, g, d_limb_, 2 * nl, d_f, d_f + nl);
    MK_HIP(hipGetLastError());
}

template <int LOG_H>
static void launch_switch_col(const u64 *last, u64 *out, const NttTables &T, uint32_t n_targets, uint32_t items, u64 q_last,
                              hipStream_t s) {
    unsigned long long mi = 0, mf = 0;
    for (uint32_t i = 0; i < n_targets; ++i) (T.h_fp_of[i] ? mf : mi) |= 1ull << i;
    const uint32_t tiles = (1u << T.log_r2) / (256u >> LOG_H);
    const uint32_t ni = (uint32_t)__builtin_popcountll(mi), nf = (uint32_t)__builtin_popcountll(mf);
    if (ni) with_int_arith(T, [&](auto ar) {
        k_switch_col<LOG_H, decltype(ar)::value>
            <<<dim3(tiles, ni, items), NTT_THREADS, 0, s>>>(last, out, T, n_targets, q_last, mi);
    });
    if (nf) k_switch_col<LOG_H, AR_FP><<<dim3(tiles, nf, items), NTT_THREADS, 0, s>>>(last, out, T, n_targets, q_last, mf);
}

void Engine::rescale(const u64 *in, u64 *out, uint32_t n_ct, uint32_t nl, const std::vector<u64> *factors) {
    need_device();
    check_nl(nl);
    if (nl < 2) throw std::invalid_argument("rescale needs at least 2 limbs");
    if (!n_ct) return;
    const uint32_t n = ps_.n, last = nl - 1, items = 2 * n_ct;
    // constants c_i = q_last^-1 (* EvalMult constant) mod q_i, cached on the device per (level, constant): no host
    // synchronisation inside a server step
    std::string key = "resc_" + std::to_string(nl);
    if (factors)
        for (uint32_t i = 0; i < last; ++i) key += "_" + std::to_string((*factors)[i]);
    const u64 *d_c = nullptr;
    {
        auto it = vec_cache_.find(key);
        if (it != vec_cache_.end()) {
            d_c = it->second;
        } else {
            std::vector<u64> c(2 * last);
            for (uint32_t i = 0; i < last; ++i) {
                u64 v = ps_.q_inv_mod(last, i);
                if (factors) v = h_mulmod(v, (*factors)[i], ps_.moduli[i]);
                c[i] = v;
                c[last + i] = h_shoup(v, ps_.moduli[i]);
            }
            if (vec_cache_.size() > 4096) throw std::runtime_error("too many distinct rescale constants cached");
            d_c = limb_vector(key, c);
        }
    }
    const size_t w_last = (size_t)items * n, w_tmp = (size_t)items * last * n;
    u64 *ws = workspace(w_last + w_tmp);
    u64 *d_last = ws, *d_tmp = ws + w_last;
    // 1. dropped limb -> COEFFICIENT format
    NttIo io{in, d_last, (size_t)nl * n, (size_t)n, last, 0, last, 1, nl};
    ntt_passes(io, tabs_, items, true, nullptr, nullptr, lanes());
    const int col_h = fast_log_h(tabs_.log_r1, 1u << tabs_.log_r2);
    if (col_h >= 2 && row_tail_supported(tabs_)) {
        // 2. centred switch of modulus fused into the forward column pass of every remaining limb; 3. row pass with
        // 4. (c_i - tmp_i) * q_last^-1 [* constant] in its copy-out (the ModDown tail with til = the input ciphertext)
        switch (col_h) {
            case 4: launch_switch_col<4>(d_last, d_tmp, tabs_, last, items, ps_.moduli[last], stream_); break;
            case 3: launch_switch_col<3>(d_last, d_tmp, tabs_, last, items, ps_.moduli[last], stream_); break;
            default: launch_switch_col<2>(d_last, d_tmp, tabs_, last, items, ps_.moduli[last], stream_); break;
        }
        MK_HIP(hipGetLastError());
        NttIo row{d_tmp, out, (size_t)last * n, (size_t)last * n, 0, 0, 0, last, last};
        TailArgs tail{in, nullptr, d_c, d_c + last, 0, nl, 1, 0u};
        launch_row<false>(row, tabs_, items, tail, lanes());
        MK_HIP(hipGetLastError());
        return;
    }
    // ring sizes without the radix kernels: the four steps as separate kernels
    EwGeom g{n, nl, ps_.L};
    k_switch_modulus<<<ew_grid(n, last, items), EW_THREADS, 0, stream_>>>(d_last, d_tmp, g, d_limb_, ps_.moduli[last]);
    MK_HIP(hipGetLastError());
    ntt_launch(d_tmp, items, last, last, false, nullptr, nullptr);
    k_sub_mul<<<ew_grid(n, last, items), EW_THREADS, 0, stream_>>>(in, d_tmp, out, g, d_limb_, d_c, d_c + last);
    MK_HIP(hipGetLastError());
}

// ---- hybrid key switching -----------------------------------------------------------

// N^-1 * [(Q_j/q_i)^-1]_{q_i} per Q limb (its digit at this level) and N^-1 * [(P/p_k)^-1]_{p_k} per P limb:
// the inverse transforms ahead of ModUp / ModDown scale by these, so the conversions take their inputs as is.
const u64 *Engine::folded_scale(uint32_t nl) {
    const std::string key = "fold_" + std::to_string(nl);
    auto it = vec_cache_.find(key);
    if (it != vec_cache_.end()) return it->second;
    const uint32_t D = ps_.D;
    std::vector<u64> v(2 * D, 0);
    auto put = [&](uint32_t id, u64 hatinv) {
        const u64 q = ps_.moduli[id];
        const u64 c = h_mulmod(ps_.limb[id].ninv, hatinv, q);
        if (ps_.limb[id].fp) {  // the inverse column pass of an fp limb scales with (double c, double c/q)
            const double cd = (double)c, cq = (double)((long double)c / (long double)q);
            std::memcpy(&v[id], &cd, 8);
            std::memcpy(&v[D + id], &cq, 8);
        } else {
            v[id] = c;
            v[D + id] = h_shoup(c, q);
        }
    };
    for (uint32_t part = 0; part < ps_.num_parts(nl); ++part) {
        BaseConvTable t = ps_.modup_table(nl, part);
        for (size_t i = 0; i < t.src.size(); ++i) put(t.src[i], t.hatinv[i]);
    }
    BaseConvTable t = ps_.moddown_table(nl);
    for (size_t i = 0; i < t.src.size(); ++i) put(t.src[i], t.hatinv[i]);
    return limb_vector(key, v);
}

const u64 *Engine::p_inverse(uint32_t nl) {
    const std::string key = "pinv_" + std::to_string(nl);
    auto it = vec_cache_.find(key);
    if (it != vec_cache_.end()) return it->second;
    std::vector<u64> pinv(2 * nl);
    for (uint32_t i = 0; i < nl; ++i) {
        pinv[i] = ps_.p_inv_mod(i);
        pinv[nl + i] = h_shoup(pinv[i], ps_.moduli[i]);
    }
    return limb_vector(key, pinv);
}

// EvalKeySwitchPrecomputeCore on `cnt` polynomials c1 (items ct_stride apart): fills the converted limbs of
// dig[item][part][ext][N] in EVALUATION format; the digits' own limbs are NOT copied (readers take them from c1).
void Engine::modup_core(const u64 *c1, size_t c1_stride, u64 *coef, u64 *dig, uint32_t cnt, uint32_t nl,
                        bool rows_int_only, uint32_t in_group, size_t in_gstride) {
    const uint32_t n = ps_.n, ext = nl + ps_.K, nparts = ps_.num_parts(nl), D = ps_.D;
    const u64 *fold = folded_scale(nl);
    // the fused conversion kernel exists when the column pass has a radix kernel; it reads packed 30-bit halves
    bool fused = fast_log_h(tabs_.log_r1, 1u << tabs_.log_r2) != 0;
    // S1: c1 -> COEFFICIENT format, scaled by N^-1 * Qhat_inv
    NttIo s1{c1, coef, c1_stride, (size_t)nl * n, 0, 0, 0, nl, nl};
    s1.in_group = in_group;  // n-client flow: polynomial p is c1 of cts[p / in_group][p % in_group]
    s1.in_gstride = in_gstride;
    const size_t dstride = (size_t)nparts * ext * n;
    // interchange format of the coefficient-form digits: fp64-class limbs as canonical doubles when every digit is
    // "all fp64-class" or "q_0 first, then fp64-class" (k_conv_col's SRCMODE 1 / 2), else packed halves throughout
    bool doubles = fused && tabs_.has_fp && fast_log_h(tabs_.log_r1, 1u << tabs_.log_r2) >= 3;
    for (uint32_t part = 0; part < nparts && doubles; ++part) {
        const DevConv &cv = modup_conv(nl, part);
        if (conv_src_mode(cv) < 0 || (conv_src_mode(cv) != 0 && cv.n_in > 4)) doubles = false;
    }
    ntt_passes(s1, tabs_, cnt, true, fold, fold + D, lanes(), fused ? (doubles ? 2 : 1) : 0);
    for (uint32_t part = 0; part < nparts && fused; ++part) {
        // S2+S3a: base conversion fused into the column pass of each complement limb
        ConvIo io{coef, dig + (size_t)part * ext * n, (size_t)nl * n, dstride, cnt, 0, 0};
        const DevConv &cv = modup_conv(nl, part);
        launch_conv_col(io, tabs_, cv, lanes(), doubles ? conv_src_mode(cv) : 0);
    }
    if (fused) {
        // S3b: one row pass over every converted limb of every digit (own limbs skipped)
        NttIo row{dig, dig, (size_t)ext * n, (size_t)ext * n, 0, 0, 0, ext, nl, nparts, ps_.alpha};
        // (rows_int_only: the fp64 limbs finish their transform inside the fused inner-product kernel)
        if (!skip_rows_) launch_row<false>(row, tabs_, cnt * nparts, TailArgs{}, lanes(), rows_int_only ? 1u : 3u);
        MK_HIP(hipGetLastError());
        return;
    }
    if (rows_int_only) throw std::logic_error("fused inner product needs the radix kernels");
    for (uint32_t part = 0; part < nparts; ++part) {  // ring sizes without a radix column kernel
        const DevConv &cv = modup_conv(nl, part);
        const uint32_t lo = part * ps_.alpha, hi = lo + cv.n_in;
        u64 *d = dig + (size_t)part * ext * n;
        launch_baseconv(coef, (size_t)nl * n, d, dstride, cv, d_limb_, n, cnt, 1, stream_);
        if (lo > 0) {
            NttIo a{d, d, dstride, dstride, 0, 0, 0, lo, nl};
            ntt_passes(a, tabs_, cnt, false, nullptr, nullptr, lanes());
        }
        if (hi < ext) {
            NttIo b{d, d, dstride, dstride, hi, hi, hi, ext - hi, nl};
            ntt_passes(b, tabs_, cnt, false, nullptr, nullptr, lanes());
        }
    }
}

// first half of ApproxModDown on `cnt` polynomials: INTT of the P limbs of til (scaled by N^-1 * Phat^-1), conversion
// P -> Q_l, forward column pass of the converted limbs -> conv [cnt][nl][N] (the row pass + tail follow).  pc is the
// scratch of the P limbs between the passes; with rows_done the inverse ROW pass already happened inside the fused
// inner-product kernel and pc holds its output.
void Engine::moddown_convert(const u64 *til, u64 *pc, u64 *conv, uint32_t cnt, uint32_t nl, bool rows_done) {
    const uint32_t n = ps_.n, K = ps_.K, ext = nl + K, D = ps_.D;
    const u64 *fold = folded_scale(nl);
    NttIo s5{til, pc, (size_t)ext * n, (size_t)K * n, nl, 0, nl, K, nl};
    const DevConv &cv = moddown_conv(nl);
    if (!rows_done) {
        ntt_passes(s5, tabs_, cnt, true, fold, fold + D, lanes(), 1);
    } else {
        NttIo second = s5;
        second.in = s5.out;
        second.in_stride = s5.out_stride;
        second.in_slot0 = s5.out_slot0;
        launch_col<true>(second, tabs_, cnt, fold, fold + D, lanes(), 1);
        MK_HIP(hipGetLastError());
    }
    ConvIo io{pc, conv, (size_t)K * n, (size_t)nl * n, cnt, 0, 0};
    launch_conv_col(io, tabs_, cv, lanes());
}

// ApproxModDown on `cnt` polynomials til[item][ext][N] -> out[item] (items out_stride apart, nl limbs each);
// add (optional): ciphertext array whose c0 is added on even items (KeySwitchInPlace: c0 += ...).
void Engine::moddown_core(const u64 *til, u64 *pc, u64 *conv, u64 *out, size_t out_stride, const u64 *add,
                          size_t add_stride, uint32_t cnt, uint32_t nl, bool accumulate, bool p_rows_done) {
    const uint32_t n = ps_.n, K = ps_.K, ext = nl + K, D = ps_.D;
    const u64 *fold = folded_scale(nl), *pinv = p_inverse(nl);
    const bool conv_fused = fast_log_h(tabs_.log_r1, 1u << tabs_.log_r2) != 0;  // conversion inside the column pass
    const bool tail_fused = conv_fused && row_tail_supported(tabs_);               // tail inside the row pass
    const DevConv &cv = moddown_conv(nl);
    EwGeom g{n, nl, ps_.L};
    if (tail_fused) {
        moddown_convert(til, pc, conv, cnt, nl, p_rows_done);
        // row pass of the converted limbs with the (ctilde_Q - conv) * P^-1 (+ c0) tail in its copy-out
        NttIo row{conv, out, (size_t)nl * n, out_stride, 0, 0, 0, nl, nl};
        TailArgs tail{til, add, pinv, pinv + nl, add_stride, ext, 1, accumulate ? 1u : 0u};
        launch_row<false>(row, tabs_, cnt, tail, lanes());
        MK_HIP(hipGetLastError());
        return;
    }
    if (conv_fused) {  // fused conversion + column pass, plain row pass, tail as its own kernel
        moddown_convert(til, pc, conv, cnt, nl, p_rows_done);
        NttIo row{conv, conv, (size_t)nl * n, (size_t)nl * n, 0, 0, 0, nl, nl};
        launch_row<false>(row, tabs_, cnt, TailArgs{}, lanes());
    } else {
        if (p_rows_done) throw std::logic_error("fused P-limb inverse needs the radix kernels");
        NttIo s5{til, pc, (size_t)ext * n, (size_t)K * n, nl, 0, nl, K, nl};
        ntt_passes(s5, tabs_, cnt, true, fold, fold + D, lanes(), 0);
        launch_baseconv(pc, (size_t)K * n, conv, (size_t)nl * n, cv, d_limb_, n, cnt, 1, stream_);
        ntt_launch(conv, cnt, nl, nl, false, nullptr, nullptr);
    }
    k_moddown_tail<<<ew_grid(n, nl, cnt), EW_THREADS, 0, stream_>>>(til, conv, out, g, d_limb_, ext, pinv, pinv + nl,
                                                                 add, add_stride, out_stride, accumulate ? 1 : 0);
    MK_HIP(hipGetLastError());
}

void Engine::modup(const u64 *c1, u64 *digits, uint32_t cnt, uint32_t nl) {
    need_device();
    check_nl(nl);
    if (!cnt) return;
    const uint32_t n = ps_.n, ext = nl + ps_.K, nparts = ps_.num_parts(nl);
    u64 *coef = workspace((size_t)cnt * nl * n);
    modup_core(c1, (size_t)nl * n, coef, digits, cnt, nl);
    // the public digit layout carries the own limbs too (EvalKeySwitchPrecomputeCore's partsCtExt)
    const size_t dstride = (size_t)nparts * ext * n;
    for (uint32_t part = 0; part < nparts; ++part) {
        const uint32_t lo = part * ps_.alpha, hi = std::min(nl, lo + ps_.alpha);
        k_copy_slots<<<ew_grid(n, hi - lo, cnt), EW_THREADS, 0, stream_>>>(c1, (size_t)nl * n, lo,
                                                                         digits + (size_t)part * ext * n, dstride, lo, n);
    }
    MK_HIP(hipGetLastError());
}

void Engine::moddown(const u64 *in, u64 *out, uint32_t cnt, uint32_t nl) {
    need_device();
    check_nl(nl);
    if (!cnt) return;
    const uint32_t n = ps_.n, K = ps_.K;
    const size_t w_pc = (size_t)cnt * K * n, w_conv = (size_t)cnt * nl * n;
    u64 *ws = workspace(w_pc + w_conv);
    moddown_core(in, ws, ws + w_pc, out, (size_t)nl * n, nullptr, 0, cnt, nl, false);
}

template <int NPARTS>
static void launch_inner(const u64 *dig, const u64 *c1, size_t c1_stride, const u64 *evk, u64 *til, EwGeom g,
                         const LimbConst *limb, uint32_t ext, uint32_t D, uint32_t alpha, uint32_t items,
                         unsigned long long slot_mask, hipStream_t s) {
    const uint32_t nsel = (uint32_t)__builtin_popcountll(slot_mask);
    if (!nsel) return;
    k_inner_product_b<NPARTS><<<dim3((g.n / 2 + EW_THREADS - 1) / EW_THREADS, nsel), EW_THREADS, 0, s>>>(
        dig, c1, c1_stride, evk, til, g, limb, ext, D, alpha, items, slot_mask);
}

template <int LOG_H, int NPARTS>
static void launch_row_inner_fp(const InnerArgs &a, const NttTables &T, hipStream_t s) {
    const uint32_t tiles = (1u << T.log_r1) / (256u >> LOG_H);
    // 2 waves per SIMD: 196 VGPRs, no spills; measured 18.35 k ct/s against 17.9 k at 3 waves (168 VGPRs, 28 spilled)
    const dim3 grid(tiles * a.nsel * a.items);
    k_row_inner_fp<LOG_H, NPARTS, 2><<<grid, NTT_THREADS, 0, s>>>(a, T);
}
template <int LOG_H>
static void launch_row_inner_fp_n(const InnerArgs &a, const NttTables &T, uint32_t nparts, hipStream_t s) {
    switch (nparts) {
        case 1: launch_row_inner_fp<LOG_H, 1>(a, T, s); break;
        case 2: launch_row_inner_fp<LOG_H, 2>(a, T, s); break;
        case 3: launch_row_inner_fp<LOG_H, 3>(a, T, s); break;
        case 4: launch_row_inner_fp<LOG_H, 4>(a, T, s); break;
        case 5: launch_row_inner_fp<LOG_H, 5>(a, T, s); break;
        case 6: launch_row_inner_fp<LOG_H, 6>(a, T, s); break;
        default: throw std::invalid_argument("more than 6 key-switch digits unsupported");
    }
}

// 256-point rows have two implementations of the fused kernels: two rounds of radix 16 (16 words per thread, 2 waves
// per SIMD) and three rounds 8 x 8 x 4 (8 words per thread, 3 waves per SIMD); Knobs::row3x picks

template <int LOGC>
static void launch_row3_inner_fp_n(const InnerArgs &a, const NttTables &T, uint32_t nparts, hipStream_t s) {
    const dim3 grid(((1u << T.log_r1) / RowT<LOGC>::ROWS) * a.nsel * a.items);
    switch (nparts) {
        case 1: k_row3_inner_fp<1, LOGC><<<grid, NTT_THREADS, 0, s>>>(a, T); break;
        case 2: k_row3_inner_fp<2, LOGC><<<grid, NTT_THREADS, 0, s>>>(a, T); break;
        case 3: k_row3_inner_fp<3, LOGC><<<grid, NTT_THREADS, 0, s>>>(a, T); break;
        case 4: k_row3_inner_fp<4, LOGC><<<grid, NTT_THREADS, 0, s>>>(a, T); break;
        case 5: k_row3_inner_fp<5, LOGC><<<grid, NTT_THREADS, 0, s>>>(a, T); break;
        case 6: k_row3_inner_fp<6, LOGC><<<grid, NTT_THREADS, 0, s>>>(a, T); break;
        default: throw std::invalid_argument("more than 6 key-switch digits unsupported");
    }
}

template <int LOGC, bool INVP>
static void launch_row3_inner_int_k(const InnerArgs &a, const NttTables &T, uint32_t nparts, uint32_t L, u64 *pc,
                                    uint32_t K, hipStream_t s) {
    if (!a.nsel) return;
    const dim3 grid(((1u << T.log_r1) / RowT<LOGC>::ROWS) * a.nsel * a.items);
    switch (nparts) {
        case 1: with_int_arith(T, [&](auto ar) {
            k_row3_inner_int<1, LOGC, INVP, decltype(ar)::value>
                <<<grid, NTT_THREADS, 0, s>>>(a, T, L, pc, K);
        }); break;
        case 2: with_int_arith(T, [&](auto ar) {
            k_row3_inner_int<2, LOGC, INVP, decltype(ar)::value>
                <<<grid, NTT_THREADS, 0, s>>>(a, T, L, pc, K);
        }); break;
        case 3: with_int_arith(T, [&](auto ar) {
            k_row3_inner_int<3, LOGC, INVP, decltype(ar)::value>
                <<<grid, NTT_THREADS, 0, s>>>(a, T, L, pc, K);
        }); break;
        case 4: with_int_arith(T, [&](auto ar) {
            k_row3_inner_int<4, LOGC, INVP, decltype(ar)::value>
                <<<grid, NTT_THREADS, 0, s>>>(a, T, L, pc, K);
        }); break;
        case 5: with_int_arith(T, [&](auto ar) {
            k_row3_inner_int<5, LOGC, INVP, decltype(ar)::value>
                <<<grid, NTT_THREADS, 0, s>>>(a, T, L, pc, K);
        }); break;
        case 6: with_int_arith(T, [&](auto ar) {
            k_row3_inner_int<6, LOGC, INVP, decltype(ar)::value>
                <<<grid, NTT_THREADS, 0, s>>>(a, T, L, pc, K);
        }); break;
        default: throw std::invalid_argument("more than 6 key-switch digits unsupported");
    }
}
// integer slots of the extended basis: Q slots (q0) keep their accumulators in til; P slots continue into the inverse
// row pass when `pc` is given (its own kernel instance: different epilogue, different register budget)
template <int LOGC>
static void launch_row3_inner_int_n(const InnerArgs &a, const NttTables &T, uint32_t nparts, uint32_t L, u64 *pc,
                                    uint32_t K, hipStream_t s) {
    if (!pc) {
        launch_row3_inner_int_k<LOGC, false>(a, T, nparts, L, nullptr, K, s);
        return;
    }
    const unsigned long long q_slots = a.nl >= 64 ? ~0ull : ((1ull << a.nl) - 1);
    InnerArgs aq = a, ap = a;
    aq.slot_mask = a.slot_mask & q_slots;
    ap.slot_mask = a.slot_mask & ~q_slots;
    aq.nsel = (uint32_t)__builtin_popcountll(aq.slot_mask);
    ap.nsel = (uint32_t)__builtin_popcountll(ap.slot_mask);
    launch_row3_inner_int_k<LOGC, true>(ap, T, nparts, L, pc, K, s);
    launch_row3_inner_int_k<LOGC, false>(aq, T, nparts, L, nullptr, K, s);
}

// S1-S4 of the hybrid key switch for `cnt` ciphertexts: ModUp digits of c1 (EvalKeySwitchPrecomputeCore) and their
// inner product with the eval key over Q_l P (EvalFastKeySwitchCoreExt) -> til [cnt][2][ext][N].
// With the radix kernels the fp64 Q limbs finish their forward transform inside k_row_inner_fp (digits stay on chip);
// integer limbs (q0, P) take k_row3_inner_int, or the row pass + k_inner_product_b on ring sizes without it.
bool Engine::keyswitch_digits(const u64 *c1, size_t ct_stride, const u64 *evk, u64 *coef, u64 *dig, u64 *til, u64 *pc,
                              uint32_t cnt, uint32_t nl) {
    const uint32_t n = ps_.n, ext = nl + ps_.K, nparts = ps_.num_parts(nl), D = ps_.D;
    const int row_h = fast_row(tabs_.log_r2, 1u << tabs_.log_r1);
    unsigned long long fp_mask = 0, all_mask = ext >= 64 ? ~0ull : ((1ull << ext) - 1);
    for (uint32_t i = 0; i < nl; ++i)
        if (tabs_.h_fp_of[i]) fp_mask |= 1ull << i;
    const bool fuse = fp_mask != 0 && (row_h == 3 || row_h == 4 || row_h == 9) &&
                      fast_log_h(tabs_.log_r1, 1u << tabs_.log_r2) != 0;
    const bool fuse_int = fuse && (row_h == 4 || row_h == 9);  // +1.3 % at C3
    {
        struct Reset {  // cleared on every exit path: the public ModUp entry point must never inherit it
            bool &flag;
            ~Reset() { flag = false; }
        } reset{skip_rows_};
        skip_rows_ = fuse_int;  // no separate row pass at all: both classes finish inside the fused kernels
        modup_core(c1, ct_stride, coef, dig, cnt, nl, fuse);
    }
    if (fuse) {
        InnerArgs a{dig, c1, evk, til, ct_stride, nl, ext, D, ps_.alpha, cnt, fp_mask,
                    (uint32_t)__builtin_popcountll(fp_mask)};
        if (row_h == 9) launch_row3_inner_fp_n<3>(a, tabs_, nparts, stream_);
        else if (row_h == 4) launch_row_inner_fp_n<4>(a, tabs_, nparts, stream_);
        else launch_row_inner_fp_n<3>(a, tabs_, nparts, stream_);
    }
    const unsigned long long mask = fuse ? (all_mask & ~fp_mask) : all_mask;
    if (fuse_int) {  // integer limbs: row pass + inner product in one three-round kernel as well
        InnerArgs a{dig, c1, evk, til, ct_stride, nl, ext, D, ps_.alpha, cnt, mask, (uint32_t)__builtin_popcountll(mask)};
        u64 *pc_fused = pc;
        if (row_h == 9) launch_row3_inner_int_n<3>(a, tabs_, nparts, ps_.L, pc_fused, ps_.K, stream_);
        else launch_row3_inner_int_n<2>(a, tabs_, nparts, ps_.L, pc_fused, ps_.K, stream_);
        MK_HIP(hipGetLastError());
        return pc_fused != nullptr;
    }
    EwGeom g{n, nl, ps_.L};
    switch (nparts) {
        case 1: launch_inner<1>(dig, c1, ct_stride, evk, til, g, d_limb_, ext, D, ps_.alpha, cnt, mask, stream_); break;
        case 2: launch_inner<2>(dig, c1, ct_stride, evk, til, g, d_limb_, ext, D, ps_.alpha, cnt, mask, stream_); break;
        case 3: launch_inner<3>(dig, c1, ct_stride, evk, til, g, d_limb_, ext, D, ps_.alpha, cnt, mask, stream_); break;
        case 4: launch_inner<4>(dig, c1, ct_stride, evk, til, g, d_limb_, ext, D, ps_.alpha, cnt, mask, stream_); break;
        case 5: launch_inner<5>(dig, c1, ct_stride, evk, til, g, d_limb_, ext, D, ps_.alpha, cnt, mask, stream_); break;
        case 6: launch_inner<6>(dig, c1, ct_stride, evk, til, g, d_limb_, ext, D, ps_.alpha, cnt, mask, stream_); break;
        default: throw std::invalid_argument("more than 6 key-switch digits unsupported");
    }
    MK_HIP(hipGetLastError());
    return false;
}

void Engine::reencrypt_chunk(const u64 *ct, const u64 *evk, u64 *out, uint32_t cnt, uint32_t nl, bool accumulate) {
    const uint32_t n = ps_.n, K = ps_.K, ext = nl + K, nparts = ps_.num_parts(nl);
    const size_t ct_stride = (size_t)2 * nl * n;
    const size_t w_coef = (size_t)cnt * nl * n, w_dig = (size_t)cnt * nparts * ext * n;
    const size_t w_til = (size_t)cnt * 2 * ext * n, w_pc = (size_t)cnt * 2 * K * n, w_conv = (size_t)cnt * 2 * nl * n;
    u64 *ws = workspace(w_coef + w_dig + w_til + w_pc + w_conv);
    u64 *coef = ws, *dig = coef + w_coef, *til = dig + w_dig, *pc = til + w_til, *conv = pc + w_pc;
    const u64 *c1 = ct + (size_t)nl * n;  // component 1 of item 0; items are ct_stride apart

    // S1-S4: ModUp digits of c1 and their inner product with the eval key
    const bool p_rows = keyswitch_digits(c1, ct_stride, evk, coef, dig, til, pc, cnt, nl);
    // S5: ApproxModDown of both components (2*cnt polynomials of ext limbs), + c0 on component 0
    moddown_core(til, pc, conv, out, (size_t)nl * n, ct, ct_stride, 2 * cnt, nl, accumulate, p_rows);
}

template <int LOGC, int MINW>
static void launch_qsum3_fp(const QSumArgs &a, const NttTables &T, uint32_t nparts, hipStream_t s) {
    const uint32_t tiles = (1u << T.log_r1) / RowT<LOGC>::ROWS;
    const dim3 grid(tiles * a.nsel * a.cnt);
    switch (nparts) {
        case 1: k_qsum3_fp<1, LOGC, MINW><<<grid, NTT_THREADS, 0, s>>>(a, T); break;
        case 2: k_qsum3_fp<2, LOGC, MINW><<<grid, NTT_THREADS, 0, s>>>(a, T); break;
        case 3: k_qsum3_fp<3, LOGC, MINW><<<grid, NTT_THREADS, 0, s>>>(a, T); break;
        case 4: k_qsum3_fp<4, LOGC, MINW><<<grid, NTT_THREADS, 0, s>>>(a, T); break;
        case 5: k_qsum3_fp<5, LOGC, MINW><<<grid, NTT_THREADS, 0, s>>>(a, T); break;
        case 6: k_qsum3_fp<6, LOGC, MINW><<<grid, NTT_THREADS, 0, s>>>(a, T); break;
        default: throw std::invalid_argument("more than 6 key-switch digits unsupported");
    }
}

// per Q limb: P mod q, (P mod q)/q, P^-1 mod q, (P^-1 mod q)/q as doubles (fp64-class limbs; zeros elsewhere)
const u64 *Engine::p_doubles() {
    auto it = vec_cache_.find("p_doubles");
    if (it != vec_cache_.end()) return it->second;
    std::vector<u64> v(4 * (size_t)ps_.L, 0);
    for (uint32_t i = 0; i < ps_.L; ++i) {
        if (!ps_.limb[i].fp) continue;
        const long double q = (long double)ps_.moduli[i];
        const u64 pm = ps_.p_mod(i), pi = ps_.p_inv_mod(i);
        const double d[4] = {(double)pm, (double)((long double)pm / q), (double)pi, (double)((long double)pi / q)};
        std::memcpy(&v[4 * (size_t)i], d, sizeof(d));
    }
    return limb_vector("p_doubles", v);
}

// the merged n-client flow needs: radix column kernels (64- or 256-point columns), 256- or 512-point rows (k_qsum3_fp,
// k_row3_inner_int and the fused tail + sum kernels exist for them), fp64-class Q limbs, integer-class P limbs
bool Engine::qsum_ok(uint32_t nl) const {
    const int row_h = fast_row(tabs_.log_r2, 1u << tabs_.log_r1);
    if ((row_h != 4 && row_h != 9) || fast_log_h(tabs_.log_r1, 1u << tabs_.log_r2) < 3) return false;
    for (uint32_t k = 0; k < ps_.K; ++k)
        if (tabs_.h_fp_of[ps_.L + k]) return false;  // k_icol_sum sums the P-limb coefficients as 64-bit integers
    for (uint32_t i = 0; i < nl; ++i)
        if (tabs_.h_fp_of[i]) return true;
    return false;
}

// sum over clients of ReEncrypt(ct_c[b], evk_c).  Merged flow (N = 2^14, 2^16 with fp64-class limbs): per chunk of
// ciphertext indices and group of clients, every phase is ONE launch over all (client, index) items:
//   ModUp of c1 -> converted digits (column-passed)                          modup_core
//   P limbs: row pass + eval-key inner product + inverse row pass            k_row3_inner_int<.., true>
//   ApproxModDown's conversion P -> Q_l (column-passed)                      moddown_convert
//   integer-class Q limbs (q_0): inner product, then row pass + tail + sum   k_row3_inner_int<.., false>, k_row3_tail_once
//   fp64-class Q limbs: everything that is left, summed over clients        k_qsum3_fp
void Engine::reencrypt_sum_merged(const u64 *cts, const u64 *evks, u64 *out, uint32_t n_clients, uint32_t n_ct,
                                  uint32_t nl) {
    const uint32_t n = ps_.n, K = ps_.K, ext = nl + K, nparts = ps_.num_parts(nl), D = ps_.D;
    const size_t ct_words = (size_t)2 * nl * n, evk_words = (size_t)ps_.beta * 2 * D * n;
    const u64 *pinv = p_inverse(nl), *pq = p_doubles();
    unsigned long long fp_mask = 0, intq_mask = 0, p_mask = 0;
    for (uint32_t i = 0; i < nl; ++i) (tabs_.h_fp_of[i] ? fp_mask : intq_mask) |= 1ull << i;
    for (uint32_t i = nl; i < ext; ++i) p_mask |= 1ull << i;
    const uint32_t n_intq = (uint32_t)__builtin_popcountll(intq_mask);
    // clients per pass: the group's P-limb coefficients are summed as 64-bit integers (k_icol_sum), so a group holds at
    // most floor((2^64 - 1) / max p_k) clients (15 for 60-bit P limbs)
    u64 max_p = 1;
    for (uint32_t k = 0; k < K; ++k) max_p = std::max(max_p, ps_.moduli[ps_.L + k]);
    const uint32_t group = std::min({n_clients, knobs_.qsum_group, (uint32_t)std::min<u64>(64, ~0ull / max_p)});
    const bool wide_rows = fast_row(tabs_.log_r2, 1u << tabs_.log_r1) == 9;  // N = 2^17: 512-point rows, three-round kernels
    // One stream: running the memory-bound sums of group g on a second stream beside the multiply-bound phases of group
    // g+1 was measured neutral to slightly negative (20.47 k against 20.59 k ct/s at groups of 4): both kinds of kernel
    // fill the register file of a CU, so the hardware time-slices them instead of co-scheduling.
    const uint32_t max_items = group * std::min(knobs_.chunk, n_ct);
    const size_t w_coef = (size_t)max_items * nl * n, w_dig = (size_t)max_items * nparts * ext * n;
    const size_t w_pc = (size_t)max_items * 2 * K * n;
    const size_t w_til = (size_t)max_items * 2 * n_intq * n;
    const size_t w_csum = (size_t)std::min(knobs_.chunk, n_ct) * 2 * nl * n;  // conversions summed over a group's clients
    const size_t w_psum = (size_t)std::min(knobs_.chunk, n_ct) * 2 * K * n;   // P-limb coefficients summed over a group's clients
    u64 *ws = workspace(w_coef + w_dig + w_pc + w_til + w_csum + w_psum);
    hipStream_t main = stream_;
    for (uint32_t b0 = 0; b0 < n_ct; b0 += knobs_.chunk) {
        const uint32_t cnt = std::min(knobs_.chunk, n_ct - b0);
        for (uint32_t g0 = 0; g0 < n_clients; g0 += group) {
            const uint32_t gc = std::min(group, n_clients - g0), items = gc * cnt;
            u64 *coef = ws, *dig = coef + w_coef, *pc = dig + w_dig, *til = pc + w_pc, *convsum = til + w_til;
            u64 *psum = convsum + w_csum;
            const u64 *ct0 = cts + ((size_t)g0 * n_ct + b0) * ct_words;  // client g0, index b0
            const u64 *c1 = ct0 + (size_t)nl * n, *evk0 = evks + (size_t)g0 * evk_words;
            const size_t ct_cstride = (size_t)n_ct * ct_words;
            {   // ModUp of every item's c1: column-passed converted limbs; the row passes happen in the consumers
                struct Reset {
                    bool &flag;
                    ~Reset() { flag = false; }
                } reset{skip_rows_};
                skip_rows_ = true;
                modup_core(c1, ct_words, coef, dig, items, nl, true, cnt, ct_cstride);
            }
            InnerArgs ia{dig, c1, evk0, til, ct_words, nl, ext, D, ps_.alpha, items, 0, 0};
            ia.ipc = cnt;
            ia.c1_gstride = ct_cstride;
            ia.evk_cstride = evk_words;
            {   // P limbs: accumulators straight through the inverse row pass into pc
                InnerArgs ap = ia;
                ap.slot_mask = p_mask;
                ap.nsel = K;
                if (wide_rows) launch_row3_inner_int_k<3, true>(ap, tabs_, nparts, ps_.L, pc, K, main);
                else launch_row3_inner_int_k<2, true>(ap, tabs_, nparts, ps_.L, pc, K, main);
            }
            {   // ApproxModDown: inverse column pass of every item's P limbs, summed over the group's clients as integers
                // (k_icol_sum), then ONE conversion P -> Q_l and forward column pass per (index, component, target limb)
                const u64 *fold = folded_scale(nl);
                launch_icol_sum(pc, psum, tabs_, fold, fold + D, K, 2 * cnt, gc, (size_t)cnt * 2 * K * n, main);
                ConvIo cs{psum, convsum, (size_t)K * n, (size_t)nl * n, 2 * cnt, 0, 0};
                launch_conv_col_psum(cs, tabs_, moddown_conv(nl), main);
            }
            if (n_intq) {  // integer-class Q limbs: accumulators through a compact til
                InnerArgs aq = ia;
                aq.slot_mask = intq_mask;
                aq.nsel = n_intq;
                aq.til_compact = 1;
                if (wide_rows) launch_row3_inner_int_k<3, false>(aq, tabs_, nparts, ps_.L, nullptr, K, main);
                else launch_row3_inner_int_k<2, false>(aq, tabs_, nparts, ps_.L, nullptr, K, main);
            }
            MK_HIP(hipGetLastError());
            if (n_intq) {  // q_0: row pass of the summed conversion, tail, sum over the group's clients
                TailOnceArgs ta{convsum, til, ct0, out + (size_t)b0 * ct_words, pinv, pinv + nl,
                                (size_t)cnt * 2 * n_intq * n, ct_cstride, ct_words, gc, nl, 2 * cnt, intq_mask, n_intq,
                                g0 != 0 ? 1u : 0u};
                const uint32_t tiles = (1u << tabs_.log_r1) / (wide_rows ? RowT<3>::ROWS : RowT<2>::ROWS);
                const dim3 grid(tiles * n_intq * 2 * cnt);
                if (wide_rows) with_int_arith(tabs_, [&](auto ar) {
                    k_row3_tail_once<3, decltype(ar)::value>
                        <<<grid, NTT_THREADS, 0, main>>>(ta, tabs_);
                });
                else with_int_arith(tabs_, [&](auto ar) {
                    k_row3_tail_once<2, decltype(ar)::value>
                        <<<grid, NTT_THREADS, 0, main>>>(ta, tabs_);
                });
            }
            QSumArgs qa{dig, convsum, ct0, evk0, out + (size_t)b0 * ct_words, pq, ct_cstride, ct_words, evk_words, ct_words,
                        gc, cnt, nl, ext, D, ps_.alpha, fp_mask, (uint32_t)__builtin_popcountll(fp_mask), g0 != 0 ? 1u : 0u};
            // 256-point rows: 2 workgroups per CU -- at N = 2^16, L = 12 the kernel's 5 632 workgroups then fill the 512 resident
            // slots exactly 11 times (7.33 times 768 at 3 per CU: a last round a third full) and nothing is parked in scratch;
            // +0.45 % on the step after the eval-key bursts (it was +-0 before them)
            if (wide_rows) launch_qsum3_fp<3, 3>(qa, tabs_, nparts, main);
            else launch_qsum3_fp<2, 2>(qa, tabs_, nparts, main);
            MK_HIP(hipGetLastError());
        }
    }
}

void Engine::reencrypt_sum(const u64 *cts, const u64 *evks, u64 *out, uint32_t n_clients, uint32_t n_ct, uint32_t nl) {
    need_device();
    check_nl(nl);
    if (!n_clients || !n_ct) return;
    const size_t ct_words = (size_t)2 * nl * ps_.n, evk_words = (size_t)ps_.beta * 2 * ps_.D * ps_.n;
    if (qsum_ok(nl)) {  // N = 2^14, 2^16, 2^17 with fp64-class Q limbs: every phase one launch over all (client, index) items
        reencrypt_sum_merged(cts, evks, out, n_clients, n_ct, nl);
        return;
    }
    // other ring sizes / arithmetic classes: one re-encryption per client with the accumulating tail
    for (uint32_t c = 0; c < n_clients; ++c)
        reencrypt(cts + (size_t)c * n_ct * ct_words, evks + (size_t)c * evk_words, out, n_ct, nl, c != 0);
}

void Engine::reencrypt(const u64 *ct, const u64 *evk, u64 *out, uint32_t n_ct, uint32_t nl, bool accumulate) {
    need_device();
    check_nl(nl);
    if (accumulate && ct == out) throw std::invalid_argument("accumulating re-encryption cannot run in place");
    const size_t ct_stride = (size_t)2 * nl * ps_.n;
    for (uint32_t done = 0; done < n_ct; done += knobs_.chunk) {
        const uint32_t cnt = n_ct - done < knobs_.chunk ? n_ct - done : knobs_.chunk;
        reencrypt_chunk(ct + done * ct_stride, evk, out + done * ct_stride, cnt, nl, accumulate);
    }
}

// ---- keys / client endpoints --------------------------------------------------------

void Engine::keygen(const int8_t *s, const u64 *a, const int32_t *e, u64 *pk, u64 *sk) {
    need_device();
    const uint32_t n = ps_.n, D = ps_.D, L = ps_.L;
    u64 *ee = workspace((size_t)D * n);
    EwGeom g{n, L, L};
    k_lift<int8_t><<<ew_grid(n, D, 1), EW_THREADS, 0, stream_>>>(s, sk, g, d_limb_, D);
    k_lift<int32_t><<<ew_grid(n, D, 1), EW_THREADS, 0, stream_>>>(e, ee, g, d_limb_, D);
    MK_HIP(hipGetLastError());
    ntt_launch(sk, 1, L, D, false, nullptr, nullptr);
    ntt_launch(ee, 1, L, D, false, nullptr, nullptr);
    // b = e - a*s ; a copied
    Opnd x{a, 0, 0}, y{sk, 0, 0}, z{ee, 0, 0}, none{nullptr, 0, 0};
    k_fma<<<ew_grid(n, D, 1), EW_THREADS, 0, stream_>>>(x, y, z, none, nullptr, 1, pk, 0, 0, g, d_limb_, D);
    MK_HIP(hipGetLastError());
    MK_HIP(hipMemcpyAsync(pk + (size_t)D * n, a, (size_t)D * n * sizeof(u64), hipMemcpyDeviceToDevice, stream_));
}

void Engine::rekeygen(const int8_t *s_old, const u64 *pk_new, const int8_t *u, const int32_t *e0,
                      const int32_t *e1, u64 *evk) {
    need_device();
    const uint32_t n = ps_.n, D = ps_.D, L = ps_.L, beta = ps_.beta;
    const size_t poly = (size_t)D * n;
    u64 *ws = workspace(poly * (1 + 3 * (size_t)beta) + D);
    u64 *se = ws, *ue = se + poly, *e0e = ue + poly * beta, *e1e = e0e + poly * beta, *d_gadget = e1e + poly * beta;
    EwGeom g{n, L, L};
    k_lift<int8_t><<<ew_grid(n, D, 1), EW_THREADS, 0, stream_>>>(s_old, se, g, d_limb_, D);
    k_lift<int8_t><<<ew_grid(n, D, beta), EW_THREADS, 0, stream_>>>(u, ue, g, d_limb_, D);
    k_lift<int32_t><<<ew_grid(n, D, beta), EW_THREADS, 0, stream_>>>(e0, e0e, g, d_limb_, D);
    k_lift<int32_t><<<ew_grid(n, D, beta), EW_THREADS, 0, stream_>>>(e1, e1e, g, d_limb_, D);
    MK_HIP(hipGetLastError());
    ntt_launch(se, 1, L, D, false, nullptr, nullptr);
    ntt_launch(ue, 3 * beta, L, D, false, nullptr, nullptr);  // ue, e0e, e1e are contiguous
    for (uint32_t part = 0; part < beta; ++part) {
        // gadget P mod q_i on the digit's own limbs, 0 elsewhere (KeySwitchGenInternal)
        std::vector<u64> gad(D, 0);
        const uint32_t lo = part * ps_.alpha, hi = std::min(L, lo + ps_.alpha);
        for (uint32_t i = lo; i < hi; ++i) gad[i] = ps_.p_mod(i);
        MK_HIP(hipMemcpyAsync(d_gadget, gad.data(), D * sizeof(u64), hipMemcpyHostToDevice, stream_));
        MK_HIP(hipStreamSynchronize(stream_));
        Opnd p0{pk_new, 0, 0}, p1{pk_new + poly, 0, 0}, uu{ue + part * poly, 0, 0};
        Opnd z0{e0e + part * poly, 0, 0}, z1{e1e + part * poly, 0, 0}, w{se, 0, 0}, none{nullptr, 0, 0};
        k_fma<<<ew_grid(n, D, 1), EW_THREADS, 0, stream_>>>(p0, uu, z0, w, d_gadget, 0, evk + (size_t)(part * 2 + 0) * poly,
                                                         0, 0, g, d_limb_, D);
        k_fma<<<ew_grid(n, D, 1), EW_THREADS, 0, stream_>>>(p1, uu, z1, none, nullptr, 0, evk + (size_t)(part * 2 + 1) * poly,
                                                         0, 0, g, d_limb_, D);
        MK_HIP(hipGetLastError());
        MK_HIP(hipStreamSynchronize(stream_));  // d_gadget is rewritten next iteration
    }
}

void Engine::lift_ntt(const double *coef, u64 *out, uint32_t cnt, uint32_t nl) {
    need_device();
    check_nl(nl);
    if (!cnt) return;
    EwGeom g{ps_.n, nl, ps_.L};
    k_lift<double><<<ew_grid(ps_.n, nl, cnt), EW_THREADS, 0, stream_>>>(coef, out, g, d_limb_, nl);
    MK_HIP(hipGetLastError());
    ntt_launch(out, cnt, nl, nl, false, nullptr, nullptr);
}

void Engine::encrypt(const u64 *pk, const u64 *pt, const int8_t *v, const int32_t *e0, const int32_t *e1, u64 *ct,
                     uint32_t n_ct, uint32_t nl) {
    need_device();
    check_nl(nl);
    if (!n_ct) return;
    const uint32_t n = ps_.n, D = ps_.D;
    const size_t poly = (size_t)nl * n;
    u64 *ws = workspace(3 * poly * n_ct);
    u64 *ve = ws, *e0e = ve + poly * n_ct, *e1e = e0e + poly * n_ct;
    EwGeom g{n, nl, ps_.L};
    k_lift<int8_t><<<ew_grid(n, nl, n_ct), EW_THREADS, 0, stream_>>>(v, ve, g, d_limb_, nl);
    k_lift<int32_t><<<ew_grid(n, nl, n_ct), EW_THREADS, 0, stream_>>>(e0, e0e, g, d_limb_, nl);
    k_lift<int32_t><<<ew_grid(n, nl, n_ct), EW_THREADS, 0, stream_>>>(e1, e1e, g, d_limb_, nl);
    MK_HIP(hipGetLastError());
    ntt_launch(ve, 3 * n_ct, nl, nl, false, nullptr, nullptr);
    // c0 = pk0*v + e0 + m ; c1 = pk1*v + e1   (pk addressed by limb id: first nl of its D limbs)
    Opnd p0{pk, 0, 1}, p1{pk + (size_t)D * n, 0, 1}, vv{ve, poly, 0};
    Opnd z0{e0e, poly, 0}, z1{e1e, poly, 0}, m{pt, poly, 0}, none{nullptr, 0, 0};
    k_fma<<<ew_grid(n, nl, n_ct), EW_THREADS, 0, stream_>>>(p0, vv, z0, m, nullptr, 0, ct, 2 * poly, 0, g, d_limb_, nl);
    k_fma<<<ew_grid(n, nl, n_ct), EW_THREADS, 0, stream_>>>(p1, vv, z1, none, nullptr, 0, ct + poly, 2 * poly, 0, g,
                                                         d_limb_, nl);
    MK_HIP(hipGetLastError());
}

void Engine::decrypt(const u64 *ct, const u64 *sk, u64 *m, uint32_t n_ct, uint32_t nl) {
    need_device();
    check_nl(nl);
    if (!n_ct) return;
    const uint32_t n = ps_.n;
    const size_t poly = (size_t)nl * n;
    EwGeom g{n, nl, ps_.L};
    // b = c1*s + c0 in EVALUATION, then COEFFICIENT
    Opnd c1{ct + poly, 2 * poly, 0}, s{sk, 0, 1}, c0{ct, 2 * poly, 0}, none{nullptr, 0, 0};
    k_fma<<<ew_grid(n, nl, n_ct), EW_THREADS, 0, stream_>>>(c1, s, c0, none, nullptr, 0, m, poly, 0, g, d_limb_, nl);
    MK_HIP(hipGetLastError());
    ntt_launch(m, n_ct, nl, nl, true, nullptr, nullptr);
}

// ---- randomness ------------------------------------------------------------------------------------

static ChaChaKey load_key(const uint8_t *key32) {
    if (!key32) throw std::invalid_argument("null sampler key");
    ChaChaKey k;
    for (int i = 0; i < 8; ++i)  // little-endian words, RFC 8439
        k.k[i] = (uint32_t)key32[4 * i] | ((uint32_t)key32[4 * i + 1] << 8) | ((uint32_t)key32[4 * i + 2] << 16) |
                 ((uint32_t)key32[4 * i + 3] << 24);
    return k;
}

void Engine::chacha_block(uint32_t *d_out16, const uint8_t *key32, uint32_t counter, const uint32_t nonce[3]) {
    need_device();
    k_chacha_block<<<1, 64, 0, stream_>>>(d_out16, load_key(key32), counter, nonce[0], nonce[1], nonce[2]);
    MK_HIP(hipGetLastError());
}

void Engine::sample_ternary(int8_t *out, size_t count, const uint8_t *key32, uint32_t sid) {
    need_device();
    const ChaChaKey key = load_key(key32);
    if (!count) return;
    k_sample_ternary<<<(unsigned)((count + 255) / 256), 256, 0, stream_>>>(out, count, key, sid);
    MK_HIP(hipGetLastError());
}

void Engine::sample_gauss(int32_t *out, size_t count, double sigma, const uint8_t *key32, uint32_t sid) {
    need_device();
    const ChaChaKey key = load_key(key32);
    if (!count) return;
    if (!(sigma > 0) || 12.0 * sigma > GAUSS_TABLE - 1) throw std::invalid_argument("sigma out of range");
    GaussTable t{};
    t.count = (int)std::ceil(12.0 * sigma) + 1;
    std::vector<long double> w(t.count);
    long double total = 0;
    for (int k = 0; k < t.count; ++k) {
        w[k] = std::exp(-(long double)k * k / (2.0L * sigma * sigma)) * (k ? 2.0L : 1.0L);
        total += w[k];
    }
    long double acc = 0;
    for (int k = 0; k < t.count; ++k) {
        acc += w[k] / total;
        const long double scaled = acc * 18446744073709551616.0L;
        t.thr[k] = scaled >= 18446744073709551615.0L ? ~0ull : (u64)scaled;
    }
    t.thr[t.count - 1] = ~0ull;
    k_sample_gauss<<<(unsigned)((count + 255) / 256), 256, 0, stream_>>>(out, count, key, sid, t);
    MK_HIP(hipGetLastError());
}

void Engine::sample_uniform(u64 *out, uint32_t items, uint32_t nl, bool with_p, const uint8_t *key32, uint32_t sid) {
    need_device();
    const ChaChaKey key = load_key(key32);
    if (nl > ps_.L || (nl == 0 && !with_p)) throw std::invalid_argument("nl out of range");
    if (!items) return;
    const uint32_t slots = nl + (with_p ? ps_.K : 0);
    k_sample_uniform<<<dim3((ps_.n + 255) / 256, slots, items), 256, 0, stream_>>>(out, ps_.n, nl, ps_.L, d_limb_, key, sid);
    MK_HIP(hipGetLastError());
}

// ---- CKKS encode / decode (fp64 canonical embedding on the device) ---------------------------------

void Engine::encode(const double *vals, u64 *pt, uint32_t cnt, uint32_t nl, double scale) {
    need_device();
    check_nl(nl);
    if (!cnt) return;
    const uint32_t n = ps_.n, slots = n / 2;
    CodecTables t{d_rot_, reinterpret_cast<const double2 *>(d_ksi_), slots, ps_.log_n - 1, 2 * n};
    // arena: complex work array [cnt][slots] then coefficient doubles [cnt][N] (both 16*slots bytes per item)
    u64 *ws = workspace((size_t)cnt * n * 2 + (size_t)cnt * n);
    double2 *v = reinterpret_cast<double2 *>(ws);
    double *coef = reinterpret_cast<double *>(ws + (size_t)cnt * n * 2);
    const dim3 gs((slots + 255) / 256, cnt), gh((slots / 2 + 255) / 256, cnt);
    k_codec_load<<<gs, 256, 0, stream_>>>(vals, v, slots);
    for (uint32_t len = slots; len >= 2; len >>= 1) k_fft_special_inv_stage<<<gh, 256, 0, stream_>>>(v, t, len);
    k_codec_to_coef<<<gs, 256, 0, stream_>>>(v, coef, t, scale);
    MK_HIP(hipGetLastError());
    EwGeom g{n, nl, ps_.L};
    k_lift<double><<<ew_grid(n, nl, cnt), EW_THREADS, 0, stream_>>>(coef, pt, g, d_limb_, nl);
    MK_HIP(hipGetLastError());
    ntt_launch(pt, cnt, nl, nl, false, nullptr, nullptr);
}

void Engine::decode(const u64 *m, double *vals, uint32_t cnt, uint32_t nl, double scale) {
    need_device();
    check_nl(nl);
    if (!cnt) return;
    if (nl > (uint32_t)CRT_MAX_LIMBS) throw std::invalid_argument("decode supports at most 32 limbs");
    const uint32_t n = ps_.n, slots = n / 2;
    CodecTables t{d_rot_, reinterpret_cast<const double2 *>(d_ksi_), slots, ps_.log_n - 1, 2 * n};
    // Garner constants for the first nl limbs: inv[a] = (q_0..q_{a-1})^-1 mod q_a, G[a][k] = q_k mod q_a
    std::vector<u64> gar((size_t)nl + (size_t)nl * nl, 0);
    for (uint32_t a = 1; a < nl; ++a) {
        const u64 qa = ps_.moduli[a];
        u64 prod = 1;
        for (uint32_t k = 0; k < a; ++k) {
            gar[nl + (size_t)a * nl + k] = ps_.moduli[k] % qa;
            prod = h_mulmod(prod, ps_.moduli[k] % qa, qa);
        }
        gar[a] = h_invmod(prod, qa);
    }
    const u64 *d_gar = limb_vector("garner_" + std::to_string(nl), gar);
    double2 *v = reinterpret_cast<double2 *>(workspace((size_t)cnt * n * 2));
    const dim3 gs((slots + 255) / 256, cnt), gh((slots / 2 + 255) / 256, cnt);
    k_crt_to_complex<<<gs, 256, 0, stream_>>>(m, v, t, d_limb_, d_gar, nl, scale);
    for (uint32_t len = 2; len <= slots; len <<= 1) k_fft_special_stage<<<gh, 256, 0, stream_>>>(v, t, len);
    k_codec_store_real<<<gs, 256, 0, stream_>>>(v, vals, slots);
    MK_HIP(hipGetLastError());
}

void Engine::host_twiddles(uint32_t limb, bool inverse, std::vector<u64> &out) const {
    std::vector<u64> sh;
    ps_.twiddles(limb, inverse, out, sh);
}

}  // namespace mk
