// sampler.hpp -- seed for the device-side samplers (mkckks_sample_*: Philox4x32-10 streams in HBM).
// OpenFHE seeds its PRNG from the OS; so does this (std::random_device) unless MKCKKS_SEED is set (tests).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <random>

namespace mkh {

inline uint64_t fresh_seed() {
    if (const char *e = std::getenv("MKCKKS_SEED")) return std::strtoull(e, nullptr, 10);
    std::random_device rd;
    return ((uint64_t)rd() << 32) ^ (uint64_t)rd() ^ ((uint64_t)rd() << 16);
}

}  // namespace mkh
