// sampler.hpp -- encryption / key-generation randomness on the host.
// Stands in for OpenFHE's TernaryUniformGeneratorImpl, DiscreteGaussianGeneratorImpl (sigma = 3.19, CC.json
// "dp") and DiscreteUniformGeneratorImpl ([upstream] core/lib/math/*generator*).  OpenFHE's PRNG stream cannot
// be reproduced, so parity here is distributional; the arithmetic that consumes the samples is bit-exact.
// Seeded from std::random_device unless MKCKKS_SEED is set (tests).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <random>
#include <vector>

namespace mkh {

class Sampler {
public:
    Sampler() {
        if (const char *e = std::getenv("MKCKKS_SEED")) {
            eng_.seed(std::strtoull(e, nullptr, 10));
        } else {
            std::random_device rd;
            std::seed_seq seq{rd(), rd(), rd(), rd(), rd(), rd(), rd(), rd()};
            eng_.seed(seq);
        }
    }
    void ternary(int8_t *out, size_t n) {  // uniform over {-1, 0, 1}
        std::uniform_int_distribution<int> d(-1, 1);
        for (size_t i = 0; i < n; ++i) out[i] = (int8_t)d(eng_);
    }
    void gaussian(int32_t *out, size_t n, double sigma = 3.19) {  // rounded normal
        std::normal_distribution<double> d(0.0, sigma);
        for (size_t i = 0; i < n; ++i) out[i] = (int32_t)std::llround(d(eng_));
    }
    void uniform(uint64_t *out, size_t n, uint64_t modulus) {  // uniform in [0, modulus)
        std::uniform_int_distribution<uint64_t> d(0, modulus - 1);
        for (size_t i = 0; i < n; ++i) out[i] = d(eng_);
    }

private:
    std::mt19937_64 eng_;
};

}  // namespace mkh
