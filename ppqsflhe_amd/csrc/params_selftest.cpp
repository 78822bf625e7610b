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
