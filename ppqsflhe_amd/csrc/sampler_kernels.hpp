// sampler_kernels.hpp -- encryption / key-generation randomness on the GPU.
//
// Stands in for OpenFHE's TernaryUniformGeneratorImpl, DiscreteGaussianGeneratorImpl (sigma = 3.19, CC.json "dp")
// and DiscreteUniformGeneratorImpl ([upstream] core/lib/math/*generator*; SURVEY 2.1 "sample_ternary/gauss/uniform"),
// consumed by KeyGen / ReKeyGen / Encrypt (keyGen.cpp:33, REkeyGen.cpp:52, encryptModelWeights.cpp:83).
// OpenFHE's PRNG stream (blake2-based) cannot be reproduced, so parity is distributional; what consumes the samples
// is bit-exact.  Generator: the ChaCha20 block function (RFC 8439) under a 256-bit key drawn from the OS by the hosts:
// a cryptographic PRF, so published outputs (the uniform polynomial a of a public key) say nothing about the other
// streams.  Counter based: block b of stream `sid`, attempt `att` is ChaCha20(key, counter = b mod 2^32,
// nonce = (b >> 32, sid, att)); element i takes 64-bit word i % 8 of block i / 8 -- a pure function of (key, sid, i),
// independent of launch geometry.
#pragma once
#include "modarith.hpp"

namespace mk {

struct ChaChaKey {
    uint32_t k[8];
};
MK_D uint32_t rotl32(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
#define MK_CHACHA_QR(a, b, c, d) \
    a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); \
    a += b; d ^= a; d = rotl32(d, 8);  c += d; b ^= c; b = rotl32(b, 7);
// the 16 output words of one block
MK_D void chacha20_block(const ChaChaKey &key, uint32_t counter, uint32_t n0, uint32_t n1, uint32_t n2, uint32_t (&out)[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key.k[0], key.k[1], key.k[2], key.k[3],
                      key.k[4],    key.k[5],    key.k[6],    key.k[7],    counter,  n0,       n1,       n2};
    uint32_t x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = s[i];
#pragma unroll 1
    for (int r = 0; r < 10; ++r) {
        MK_CHACHA_QR(x[0], x[4], x[8], x[12])
        MK_CHACHA_QR(x[1], x[5], x[9], x[13])
        MK_CHACHA_QR(x[2], x[6], x[10], x[14])
        MK_CHACHA_QR(x[3], x[7], x[11], x[15])
        MK_CHACHA_QR(x[0], x[5], x[10], x[15])
        MK_CHACHA_QR(x[1], x[6], x[11], x[12])
        MK_CHACHA_QR(x[2], x[7], x[8], x[13])
        MK_CHACHA_QR(x[3], x[4], x[9], x[14])
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) out[i] = x[i] + s[i];
}
#undef MK_CHACHA_QR
// 64-bit word `w` (0..7) of block b of stream sid
MK_D u64 chacha_u64(const ChaChaKey &key, uint32_t sid, uint64_t b, uint32_t att, int w) {
    uint32_t o[16];
    chacha20_block(key, (uint32_t)b, (uint32_t)(b >> 32), sid, att, o);
    u64 r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (i == w) r = ((u64)o[2 * i + 1] << 32) | o[2 * i];
    return r;
}

// raw block (known-answer test hook: RFC 8439 2.3.2)
__global__ void k_chacha_block(uint32_t *out, ChaChaKey key, uint32_t counter, uint32_t n0, uint32_t n1, uint32_t n2) {
    uint32_t o[16];
    chacha20_block(key, counter, n0, n1, n2, o);
    if (threadIdx.x == 0 && blockIdx.x == 0)
        for (int i = 0; i < 16; ++i) out[i] = o[i];
}

// uniform over {-1, 0, 1}: 64-bit multiply-shift (bias < 2^-62)
__global__ void k_sample_ternary(int8_t *out, size_t n, ChaChaKey key, uint32_t sid) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 r = chacha_u64(key, sid, i >> 3, 0, (int)(i & 7));
    out[i] = (int8_t)((int)mulhi64(r, 3) - 1);
}

// discrete Gaussian D_{Z,sigma} by inversion of the cumulative table of |x| (thr[k] = 2^64 * P(|x| <= k)), sign from
// an independent word; tail cut at GAUSS_TABLE-1 >= 12 sigma for sigma <= 3.2.  Element i: words 2(i%4), 2(i%4)+1 of
// block i/4.
constexpr int GAUSS_TABLE = 48;
struct GaussTable {
    u64 thr[GAUSS_TABLE];
    int count;
};
__global__ void k_sample_gauss(int32_t *out, size_t n, ChaChaKey key, uint32_t sid, GaussTable t) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t o[16];
    const uint64_t b = i >> 2;
    chacha20_block(key, (uint32_t)b, (uint32_t)(b >> 32), sid, 0, o);
    u64 r = 0;
    uint32_t sign = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j == (int)(i & 3)) {
            r = ((u64)o[4 * j + 1] << 32) | o[4 * j];
            sign = o[4 * j + 2] & 1u;
        }
    int k = 0;
    while (k < t.count - 1 && r >= t.thr[k]) ++k;
    out[i] = sign ? -k : k;
}

// uniform residues in [0, q) per limb by rejection (accept r < 2^64 - (2^64 mod q)); out [items][slots][N]
__global__ void k_sample_uniform(u64 *out, uint32_t n, uint32_t nl, uint32_t L, const LimbConst *limb, ChaChaKey key,
                                 uint32_t sid) {
    const uint32_t slot = blockIdx.y, idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const LimbConst lc = limb[slot < nl ? slot : L + (slot - nl)];
    const size_t pos = ((size_t)blockIdx.z * gridDim.y + slot) * n + idx;
    const u64 limit = 0 - lc.c64;  // floor(2^64 / q) * q
    u64 r = 0;
    for (uint32_t attempt = 0; attempt < 64; ++attempt) {
        r = chacha_u64(key, sid, pos >> 3, attempt, (int)(pos & 7));
        if (r < limit) break;
    }
    out[pos] = reduce_word(r, lc);
}

}  // namespace mk
