// codec.hpp -- the floating-point half of CKKS encoding/decoding, on the host.
//
// Stands in for CKKSPackedEncoding::Encode / Decode and DiscreteFourierTransform::FFTSpecialInv /
// FFTSpecial ([upstream] pke/lib/encoding/ckkspackedencoding.cpp, core/lib/math/dftransform.cpp), reached from
// client/src/encryptModelWeights.cpp:82,90,109 (MakeCKKSPackedPlaintext) and
// client/src/decryptModelWeights.cpp:83,92,109 (GetRealPackedValue).  Full packing: slots = N/2, slot j sits at
// the primitive 2N-th root zeta^(5^j).  The integer half (rounding, residues, NTT) runs on the GPU
// (mkckks_lift_ntt_batch); CRT interpolation of a decrypted polynomial is done here.
// Decode-time noise flooding of upstream Decode (a Gaussian added to hide the CKKS error) is NOT reproduced:
// decoded values are the exact canonical embedding of the decrypted polynomial.
#pragma once
#include <cmath>
#include <complex>
#include <cstdint>
#include <vector>

namespace mkh {

class Codec {
public:
    explicit Codec(uint32_t ring_dim) : n_(ring_dim), slots_(ring_dim / 2), m_(2 * ring_dim) {
        rot_.resize(slots_);
        uint64_t p = 1;
        for (uint32_t j = 0; j < slots_; ++j) { rot_[j] = (uint32_t)p; p = p * 5 % m_; }
        ksi_.resize(m_ + 1);
        const double pi = std::acos(-1.0);
        for (uint32_t k = 0; k <= m_; ++k) ksi_[k] = std::polar(1.0, 2.0 * pi * (double)k / (double)m_);
    }
    uint32_t slots() const { return slots_; }

    // real slot values (zero padded to N/2) -> N scaled real coefficients (not yet rounded)
    void encode(const double *vals, size_t count, double scale, double *coef_out) const {
        std::vector<std::complex<double>> v(slots_);
        for (size_t i = 0; i < count && i < slots_; ++i) v[i] = vals[i];
        inverse_embedding(v);
        for (uint32_t i = 0; i < slots_; ++i) {
            coef_out[i] = v[i].real() * scale;
            coef_out[i + slots_] = v[i].imag() * scale;
        }
    }

    // COEFFICIENT-format residues m[nl][N] of a decrypted polynomial -> N/2 real slot values (/ scale)
    void decode(const uint64_t *m, uint32_t nl, const uint64_t *moduli, double scale, double *vals_out) const {
        std::vector<long double> c(n_);
        crt_centered(m, nl, moduli, c.data());
        std::vector<std::complex<double>> v(slots_);
        for (uint32_t i = 0; i < slots_; ++i)
            v[i] = std::complex<double>((double)(c[i] / (long double)scale), (double)(c[i + slots_] / (long double)scale));
        embedding(v);
        for (uint32_t i = 0; i < slots_; ++i) vals_out[i] = v[i].real();
    }

    // Garner mixed-radix interpolation to the centred representative in (-Q/2, Q/2], as long double
    void crt_centered(const uint64_t *m, uint32_t nl, const uint64_t *q, long double *out) const {
        typedef unsigned __int128 u128;
        std::vector<uint64_t> inv(nl, 0);
        auto mulmod = [](uint64_t a, uint64_t b, uint64_t mod) { return (uint64_t)((u128)a * b % mod); };
        auto powmod = [&](uint64_t a, uint64_t e, uint64_t mod) {
            uint64_t r = 1; a %= mod;
            for (; e; e >>= 1) { if (e & 1) r = mulmod(r, a, mod); a = mulmod(a, a, mod); }
            return r;
        };
        for (uint32_t i = 1; i < nl; ++i) {
            uint64_t p = 1;
            for (uint32_t k = 0; k < i; ++k) p = mulmod(p, q[k] % q[i], q[i]);
            inv[i] = powmod(p, q[i] - 2, q[i]);
        }
        std::vector<uint64_t> dig(nl);
        for (uint32_t j = 0; j < n_; ++j) {
            dig[0] = m[j];
            for (uint32_t i = 1; i < nl; ++i) {
                const uint64_t qi = q[i];
                uint64_t acc = dig[i - 1] % qi;
                for (int k = (int)i - 2; k >= 0; --k) acc = (uint64_t)(((u128)acc * (q[k] % qi) + dig[k] % qi) % qi);
                uint64_t diff = m[(size_t)i * n_ + j] % qi;
                diff = diff >= acc ? diff - acc : diff + qi - acc;
                dig[i] = mulmod(diff, inv[i], qi);
            }
            bool neg = false;  // above (Q-1)/2 ?  compare digits with (q_i - 1)/2 from the top
            for (int i = (int)nl - 1; i >= 0; --i) {
                const uint64_t half = (q[i] - 1) / 2;
                if (dig[i] != half) { neg = dig[i] > half; break; }
            }
            long double acc = 0;
            for (int i = (int)nl - 1; i >= 0; --i)
                acc = acc * (long double)q[i] + (long double)(neg ? q[i] - 1 - dig[i] : dig[i]);
            out[j] = neg ? -(acc + 1) : acc;
        }
    }

private:
    static void bit_reverse(std::vector<std::complex<double>> &v) {
        const size_t n = v.size();
        for (size_t i = 1, j = 0; i < n; ++i) {
            size_t bit = n >> 1;
            for (; j & bit; bit >>= 1) j ^= bit;
            j ^= bit;
            if (i < j) std::swap(v[i], v[j]);
        }
    }
    void embedding(std::vector<std::complex<double>> &v) const {  // coefficients -> slots
        const uint32_t size = slots_;
        bit_reverse(v);
        for (uint32_t len = 2; len <= size; len <<= 1) {
            const uint32_t half = len >> 1, quad = len << 2, gap = m_ / quad;
            for (uint32_t i = 0; i < size; i += len)
                for (uint32_t j = 0; j < half; ++j) {
                    const std::complex<double> w = ksi_[(rot_[j] % quad) * gap];
                    const std::complex<double> u = v[i + j], t = v[i + j + half] * w;
                    v[i + j] = u + t;
                    v[i + j + half] = u - t;
                }
        }
    }
    void inverse_embedding(std::vector<std::complex<double>> &v) const {  // slots -> coefficients
        const uint32_t size = slots_;
        for (uint32_t len = size; len >= 2; len >>= 1) {
            const uint32_t half = len >> 1, quad = len << 2, gap = m_ / quad;
            for (uint32_t i = 0; i < size; i += len)
                for (uint32_t j = 0; j < half; ++j) {
                    const std::complex<double> w = ksi_[(quad - (rot_[j] % quad)) * gap];
                    const std::complex<double> u = v[i + j] + v[i + j + half];
                    const std::complex<double> t = (v[i + j] - v[i + j + half]) * w;
                    v[i + j] = u;
                    v[i + j + half] = t;
                }
        }
        bit_reverse(v);
        for (auto &x : v) x /= (double)size;
    }

    uint32_t n_, slots_, m_;
    std::vector<uint32_t> rot_;
    std::vector<std::complex<double>> ksi_;
};

}  // namespace mkh
