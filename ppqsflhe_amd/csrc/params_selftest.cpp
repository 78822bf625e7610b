// params_selftest -- params.cpp (parameter generation, CRT / base-conversion tables, twiddles) on the host only: the
// piece of the library that is plain C++ and can run under -fsanitize=address,undefined (`make asan`; GPU code cannot be
// sanitised on this pool).  Prints a checksum line per configuration; tests/test_sanitizers.py runs it.
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

#include "params.hpp"

int main() {
    struct Cfg { uint32_t log_n, depth, sbits, first, dnum; };
    const Cfg cfgs[] = {{10, 3, 40, 60, 2}, {12, 1, 40, 60, 2}, {14, 2, 40, 60, 2}, {12, 18, 50, 60, 3}, {16, 10, 50, 60, 3}};
    try {
        for (const Cfg &c : cfgs) {
            mk::ParamSet ps;
            ps.generate(c.log_n, c.depth, c.sbits, c.first, c.dnum, 60, 20);
            unsigned long long sum = 0;
            for (uint32_t nl = 1; nl <= ps.L; ++nl) {
                for (uint32_t part = 0; part < ps.num_parts(nl); ++part) {
                    const mk::BaseConvTable t = ps.modup_table(nl, part);
                    if (t.hat.size() != t.src.size() * t.dst.size()) throw std::runtime_error("modup table shape");
                    for (mk::u64 v : t.hat) sum += v;
                }
                const mk::BaseConvTable md = ps.moddown_table(nl);
                for (mk::u64 v : md.hat) sum += v;
                (void)ps.const_factors(nl, ps.L - nl, 0.5);
            }
            std::vector<mk::u64> w, wsh;
            for (uint32_t id : {0u, ps.L - 1, ps.D - 1}) {
                ps.twiddles(id, false, w, wsh);
                ps.twiddles(id, true, w, wsh);
                sum += w[1] + wsh[ps.n - 1];
                sum += ps.p_mod(id % ps.L) + ps.p_inv_mod(id % ps.L);
            }
            std::printf("ok log_n=%u L=%u K=%u alpha=%u beta=%u checksum=%llu\n", ps.log_n, ps.L, ps.K, ps.alpha, ps.beta, sum);
        }
        // pseudo-Mersenne arithmetic of modarith.hpp (host mirror of the device code): congruence and the stated output
        // bounds on boundary and random operands, for every eligible prime the configurations above produce plus 55-
        // and 58-bit first moduli
        {
            using mk::u64;
            typedef unsigned __int128 u128;
            unsigned long long checked = 0;
            u64 rng = 0x9E3779B97F4A7C15ull;
            auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
            for (const Cfg &c : {Cfg{14, 2, 40, 60, 2}, Cfg{16, 10, 50, 60, 3}, Cfg{14, 2, 40, 55, 2}, Cfg{14, 2, 40, 58, 2},
                                 Cfg{17, 19, 50, 60, 3}}) {
                mk::ParamSet ps;
                ps.generate(c.log_n, c.depth, c.sbits, c.first, c.dnum, 60, 20);
                for (uint32_t i = 0; i < ps.D; ++i) {
                    const u64 q = ps.moduli[i];
                    if (q < (5ull << 48) || !mk::pm_eligible(q)) continue;
                    mk::LimbConst lc = ps.limb[i];
                    lc.pm = 1;
                    lc.pm_c = (uint32_t)(((u64)1 << lc.k) - q);
                    const mk::PmK P = mk::pm_consts(lc);
                    const u64 U = (u64)1 << lc.k, amax = (U << 3) - 1;
                    const u64 as[] = {0, 1, q - 1, q, U - 1, U, 2 * q, amax, amax - 1, (U << 2) + 12345, 0xffffffffull,
                                      0x100000000ull, amax & ~0xffffffffull};
                    const u64 ws[] = {0, 1, q - 1, q / 2, 0xffffffffull, 0x100000000ull, q - 0xffffffffull};
                    auto check = [&](u64 a, u64 w) {
                        const u64 r = mk::pm_lazy(a, mk::pm_tw(w, lc), mk::pm_tw_companion(w, lc), P);
                        if (r % q != (u64)((u128)a * w % q)) throw std::runtime_error("pm_lazy: wrong residue");
                        if ((u128)r * 8 >= (u128)U * 19) throw std::runtime_error("pm_lazy: result not below 2.375 * 2^k");
                        if (r > P.q3) throw std::runtime_error("pm_lazy: result above 3q");
                        ++checked;
                    };
                    for (u64 a : as)
                        for (u64 w : ws) check(a, w);
                    for (int it = 0; it < 20000; ++it) check(next() & amax, next() % q);
                    // 128-bit accumulators of the eval-key inner product: up to 6 products (lazy word < 8U) x (residue < q)
                    for (int it = 0; it < 20000; ++it) {
                        u128 X = 0;
                        const int terms = 1 + it % 6;
                        for (int t = 0; t < terms; ++t) {
                            const u64 a = (it % 5 == 0) ? amax : (next() & amax), b = (it % 7 == 0) ? q - 1 : next() % q;
                            X += (u128)a * b;
                        }
                        const u64 r = mk::pm_reduce128((u64)(X >> 64), (u64)X, P, q);
                        if (r != (u64)(X % q)) throw std::runtime_error("pm_reduce128");
                        ++checked;
                    }
                    // column sums of up to 4 products of 60-bit numbers split in 30-bit halves (base conversion)
                    for (int it = 0; it < 20000; ++it) {
                        mk::Cols cs{0, 0, 0};
                        u128 X = 0;
                        for (int t = 0; t < 1 + it % 4; ++t) {
                            const u64 a = (it % 5 == 0) ? (1ull << 60) - 1 : next() >> 4, b = (it % 3 == 0) ? q - 1 : next() % q;
                            uint32_t a0, a1, b0, b1;
                            mk::split30(a, a0, a1);
                            mk::split30(b, b0, b1);
                            mk::mac_cols(cs, a0, a1, b0, b1);
                            X += (u128)a * b;
                        }
                        const u64 r = mk::pm_reduce_cols(cs, P);
                        if (r % q != (u64)(X % q) || r >= 2 * U + (U >> 3)) throw std::runtime_error("pm_reduce_cols");
                        ++checked;
                    }
                    const u64 xs[] = {0, q, U, ~0ull, ~0ull - 1, U - 1, amax};
                    for (u64 x : xs) {
                        const u64 f = mk::pm_fold(x, P);
                        if (f % q != x % q || f >= U + (1ull << 30)) throw std::runtime_error("pm_fold");
                    }
                    for (int it = 0; it < 20000; ++it) {
                        const u64 x = next(), f = mk::pm_fold(x, P);
                        if (f % q != x % q || f >= U + (1ull << 30)) throw std::runtime_error("pm_fold");
                    }
                }
            }
            if (!checked) throw std::runtime_error("no pseudo-Mersenne prime was exercised");
            std::printf("ok pseudo-Mersenne arithmetic (%llu products)\n", checked);
        }
        mk::ParamSet bad;
        try {
            bad.generate(30, 1, 40, 60, 2, 60, 20);
            std::printf("unexpected: log_n=30 accepted\n");
            return 1;
        } catch (const std::invalid_argument &) {
            std::printf("ok invalid parameters are refused\n");
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "ERROR: %s\n", e.what());
        return 1;
    }
    return 0;
}
