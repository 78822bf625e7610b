// sampler_kernels.hpp -- encryption / key-generation randomness on the GPU.
//
// Stands in for OpenFHE's TernaryUniformGeneratorImpl, DiscreteGaussianGeneratorImpl (sigma = 3.19, CC.json "dp")
// and DiscreteUniformGeneratorImpl ([upstream] core/lib/math/*generator*; SURVEY 2.1 "sample_ternary/gauss/uniform"),
// consumed by KeyGen / ReKeyGen / Encrypt (keyGen.cpp:33, REkeyGen.cpp:52, encryptModelWeights.cpp:83).
// OpenFHE's PRNG stream (blake2-based) cannot be reproduced, so parity is distributional; what consumes the samples
// is bit-exact.  Generator: Philox4x32-10 (counter based): element i of stream `sid` under `seed` is a pure function
// of (seed, sid, i), independent of launch geometry.
#pragma once
#include "modarith.hpp"

namespace mk {

struct Philox {
    uint32_t c[4];
};
MK_D Philox philox4x32_10(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    Philox p{{c0, c1, c2, c3}};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t m0 = (uint64_t)0xD2511F53u * p.c[0], m1 = (uint64_t)0xCD9E8D57u * p.c[2];
        const uint32_t n0 = (uint32_t)(m1 >> 32) ^ p.c[1] ^ k0, n2 = (uint32_t)(m0 >> 32) ^ p.c[3] ^ k1;
        p.c[1] = (uint32_t)m1;
        p.c[3] = (uint32_t)m0;
        p.c[0] = n0;
        p.c[2] = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return p;
}
MK_D u64 philox_u64(uint64_t seed, uint32_t sid, uint64_t i, uint32_t attempt) {
    const Philox p = philox4x32_10(seed, (uint32_t)i, (uint32_t)(i >> 32), sid, attempt);
    return ((u64)p.c[1] << 32) | p.c[0];
}

// uniform over {-1, 0, 1}: 64-bit multiply-shift (bias < 2^-62)
__global__ void k_sample_ternary(int8_t *out, size_t n, uint64_t seed, uint32_t sid) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 r = philox_u64(seed, sid, i, 0);
    out[i] = (int8_t)((int)mulhi64(r, 3) - 1);
}

// discrete Gaussian D_{Z,sigma} by inversion of the cumulative table of |x| (thr[k] = 2^64 * P(|x| <= k)), sign from
// an independent bit; tail cut at GAUSS_TABLE-1 >= 12 sigma for sigma <= 3.2
constexpr int GAUSS_TABLE = 48;
struct GaussTable {
    u64 thr[GAUSS_TABLE];
    int count;
};
__global__ void k_sample_gauss(int32_t *out, size_t n, uint64_t seed, uint32_t sid, GaussTable t) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Philox p = philox4x32_10(seed, (uint32_t)i, (uint32_t)(i >> 32), sid, 0);
    const u64 r = ((u64)p.c[1] << 32) | p.c[0];
    int k = 0;
    while (k < t.count - 1 && r >= t.thr[k]) ++k;
    out[i] = (p.c[2] & 1) ? -k : k;
}

// uniform residues in [0, q) per limb by rejection (accept r < 2^64 - (2^64 mod q)); out [items][slots][N]
__global__ void k_sample_uniform(u64 *out, uint32_t n, uint32_t nl, uint32_t L, const LimbConst *limb, uint64_t seed,
                                 uint32_t sid) {
    const uint32_t slot = blockIdx.y, idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const LimbConst lc = limb[slot < nl ? slot : L + (slot - nl)];
    const size_t pos = ((size_t)blockIdx.z * gridDim.y + slot) * n + idx;
    const u64 limit = 0 - lc.c64;  // floor(2^64 / q) * q
    u64 r = 0;
    for (uint32_t attempt = 0; attempt < 64; ++attempt) {
        r = philox_u64(seed, sid, pos, attempt);
        if (r < limit) break;
    }
    out[pos] = reduce_word(r, lc);
}

}  // namespace mk
