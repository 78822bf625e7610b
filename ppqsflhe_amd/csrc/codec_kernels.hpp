// codec_kernels.hpp -- the floating-point half of CKKS Encode / Decode on the GPU (fp64).
//
// Stands in for CKKSPackedEncoding::Encode / Decode and DiscreteFourierTransform::FFTSpecialInv / FFTSpecial
// ([upstream] pke/lib/encoding/ckkspackedencoding.cpp, core/lib/math/dftransform.cpp), reached from
// client/src/encryptModelWeights.cpp:82,90,109 and client/src/decryptModelWeights.cpp:83,92,109.
// Full packing: N/2 complex slots, slot j at zeta^(5^j).  One launch per butterfly stage (15 at N = 2^16): these are
// client-side, per-round operations on a few dozen ciphertexts, far off the server's hot loop.
#pragma once
#include "modarith.hpp"

namespace mk {

struct CodecTables {
    const uint32_t *rot;   // 5^j mod 2N, j < N/2
    const double2 *ksi;    // exp(2 pi i k / 2N), k <= 2N
    uint32_t slots, log_slots, m;  // N/2, log2, 2N
};

MK_D double2 cmul(double2 a, double2 b) { return double2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
MK_D uint32_t brev(uint32_t x, uint32_t bits) { return __brev(x) >> (32 - bits); }

// decode direction (FFTSpecial): one stage of length `len` on v[items][slots]
__global__ void k_fft_special_stage(double2 *v, CodecTables t, uint32_t len) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= t.slots / 2) return;
    double2 *p = v + (size_t)blockIdx.y * t.slots;
    const uint32_t half = len >> 1, quad = len << 2, gap = t.m / quad;
    const uint32_t i = (b / half) * len, j = b % half;
    const double2 w = t.ksi[(t.rot[j] % quad) * gap];
    const double2 u = p[i + j], x = cmul(p[i + j + half], w);
    p[i + j] = double2{u.x + x.x, u.y + x.y};
    p[i + j + half] = double2{u.x - x.x, u.y - x.y};
}
// encode direction (FFTSpecialInv), before the bit reversal and the 1/size scaling
__global__ void k_fft_special_inv_stage(double2 *v, CodecTables t, uint32_t len) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= t.slots / 2) return;
    double2 *p = v + (size_t)blockIdx.y * t.slots;
    const uint32_t half = len >> 1, quad = len << 2, gap = t.m / quad;
    const uint32_t i = (b / half) * len, j = b % half;
    const double2 w = t.ksi[(quad - (t.rot[j] % quad)) * gap];
    const double2 a = p[i + j], c = p[i + j + half];
    p[i + j] = double2{a.x + c.x, a.y + c.y};
    p[i + j + half] = cmul(double2{a.x - c.x, a.y - c.y}, w);
}
// real slot values -> complex work array
__global__ void k_codec_load(const double *vals, double2 *v, uint32_t slots) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= slots) return;
    v[(size_t)blockIdx.y * slots + i] = double2{vals[(size_t)blockIdx.y * slots + i], 0.0};
}
// bit reversal + 1/size + scale; real parts -> coef[0..slots), imaginary parts -> coef[slots..N)
__global__ void k_codec_to_coef(const double2 *v, double *coef, CodecTables t, double scale) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.slots) return;
    const double2 x = v[(size_t)blockIdx.y * t.slots + brev(i, t.log_slots)];
    const double f = scale / (double)t.slots;
    double *c = coef + (size_t)blockIdx.y * 2 * t.slots;
    c[i] = x.x * f;
    c[i + t.slots] = x.y * f;
}
__global__ void k_codec_store_real(const double2 *v, double *vals, uint32_t slots) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= slots) return;
    vals[(size_t)blockIdx.y * slots + i] = v[(size_t)blockIdx.y * slots + i].x;
}

// CRT interpolation of a decrypted polynomial (Garner mixed radix, centred lift) -> value / scale as fp64,
// written bit-reversed into the complex work array (folds FFTSpecial's leading bit reversal).
// m: [items][nl][N] COEFFICIENT-format residues.  garner: inv[nl] then G[nl][nl] with G[i][k] = q_k mod q_i.
constexpr int CRT_MAX_LIMBS = 32;
__global__ void k_crt_to_complex(const u64 *m, double2 *v, CodecTables t, const LimbConst *limb, const u64 *garner,
                                 uint32_t nl, double scale) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.slots) return;
    const uint32_t n = 2 * t.slots;
    const u64 *mp = m + (size_t)blockIdx.y * nl * n;
    double part[2];
    for (int h = 0; h < 2; ++h) {
        const uint32_t j = i + h * t.slots;
        u64 dig[CRT_MAX_LIMBS];
        dig[0] = mp[j];
        for (uint32_t a = 1; a < nl; ++a) {
            const LimbConst la = limb[a];
            const u64 *G = garner + nl + (size_t)a * nl;
            u64 acc = reduce_word(dig[a - 1], la);
            for (int k = (int)a - 2; k >= 0; --k)
                acc = add_mod(mul_mod(acc, G[k], la), reduce_word(dig[k], la), la.q);
            dig[a] = mul_mod(sub_mod(mp[(size_t)a * n + j], acc, la.q), garner[a], la);
        }
        bool neg = false;  // above (Q-1)/2 ?  digits compared with (q_a - 1)/2 from the top
        for (int a = (int)nl - 1; a >= 0; --a) {
            const u64 half = (limb[a].q - 1) >> 1;
            if (dig[a] != half) { neg = dig[a] > half; break; }
        }
        double acc = 0.0;
        for (int a = (int)nl - 1; a >= 0; --a) {
            const u64 q = limb[a].q;
            acc = acc * (double)q + (double)(neg ? q - 1 - dig[a] : dig[a]);
        }
        part[h] = (neg ? -(acc + 1.0) : acc) / scale;
    }
    v[(size_t)blockIdx.y * t.slots + brev(i, t.log_slots)] = double2{part[0], part[1]};
}

}  // namespace mk
