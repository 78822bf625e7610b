// sampler.hpp -- keys for the device-side samplers (mkckks_sample_*: ChaCha20 streams in HBM).
// OpenFHE seeds its Blake2 PRNG from the OS; so does this: every sampler key is 256 bits straight from getrandom(2)
// (or /dev/urandom), and the secret, error and public streams of one program get INDEPENDENT keys.
// Deterministic keys exist only in builds made with -DMKCKKS_TEST_SEED (never the default): MKCKKS_SEED=<decimal>
// then expands to keys by counter; an unparsable value is an error, not seed 0.
#pragma once
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <sys/random.h>

namespace mkh {

struct SamplerKey {
    uint8_t bytes[32];
};

inline void os_random(void *dst, size_t len) {
    uint8_t *p = static_cast<uint8_t *>(dst);
    size_t got = 0;
    while (got < len) {
        const ssize_t r = getrandom(p + got, len - got, 0);
        if (r > 0) {
            got += (size_t)r;
            continue;
        }
        if (r < 0 && errno == EINTR) continue;
        break;
    }
    if (got < len) {  // kernels without getrandom: the urandom device
        FILE *f = std::fopen("/dev/urandom", "rb");
        if (!f || std::fread(p + got, 1, len - got, f) != len - got) {
            if (f) std::fclose(f);
            throw std::runtime_error("no operating-system randomness available");
        }
        std::fclose(f);
    }
}

// a fresh, independent 256-bit key per call
inline SamplerKey fresh_key() {
    SamplerKey k;
#ifdef MKCKKS_TEST_SEED
    if (const char *e = std::getenv("MKCKKS_SEED")) {
        char *end = nullptr;
        errno = 0;
        const unsigned long long seed = std::strtoull(e, &end, 10);
        if (errno || end == e || *end != '\0') throw std::runtime_error("MKCKKS_SEED is not a decimal number");
        static uint64_t counter = 0;
        std::memset(k.bytes, 0, sizeof k.bytes);
        std::memcpy(k.bytes, &seed, 8);
        std::memcpy(k.bytes + 8, &counter, 8);
        ++counter;
        return k;
    }
#endif
    os_random(k.bytes, sizeof k.bytes);
    return k;
}

}  // namespace mkh
