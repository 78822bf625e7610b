// params.hpp -- host-side CKKS/RNS parameter set and CRT tables.
//
// Product-side counterpart of what OpenFHE builds in GenCryptoContext
// (reference server/src/genCC.cpp:32-79) and rebuilds on every
// Serial::DeserializeFromFile(cc) (changeCipherDomain.cpp:33 etc.):
// [upstream] ckksrns-parametergeneration.cpp (moduli selection, FLEXIBLEAUTOEXT),
// rns-cryptoparameters.cpp::PrecomputeCRTTables (HYBRID digits, special primes,
// base-conversion tables), ckksrns-cryptoparameters.cpp (scaling factors),
// nbtheory (FirstPrime/LastPrime/PreviousPrime/NextPrime, minimal RootOfUnity).
#pragma once
#include <cstdint>
#include <map>
#include <vector>

#include "modarith.hpp"

namespace mk {

// Number theory on the host.
u64 h_mulmod(u64 a, u64 b, u64 m);
u64 h_powmod(u64 a, u64 e, u64 m);
u64 h_invmod(u64 a, u64 m);
bool h_is_prime(u64 n);
u64 h_shoup(u64 w, u64 q);
u64 h_min_primitive_root(u64 order, u64 q);

// Tables for one approximate base conversion (ApproxSwitchCRTBasis):
// source limbs S = {s_i}, target limbs T = {t_j}.
struct BaseConvTable {
    std::vector<uint32_t> src;   // limb ids (index into QP table)
    std::vector<uint32_t> dst;   // limb ids
    std::vector<u64> hatinv;     // [(S/s_i)^-1]_{s_i}
    std::vector<u64> hatinv_sh;  // Shoup companions
    std::vector<u64> hat;        // [S/s_i]_{t_j}, row-major [i][j]
};

struct ParamSet {
    uint32_t log_n = 0, n = 0;
    uint32_t L = 0, K = 0, D = 0;  // #Q limbs, #P limbs, L+K
    uint32_t alpha = 0, beta = 0;  // limbs per digit, digits at full level
    uint32_t mult_depth = 0, scaling_bits = 0, first_bits = 0, dnum = 0, aux_bits = 0, extra_bits = 0;
    std::vector<u64> moduli;       // D entries: Q then P
    std::vector<u64> roots;        // minimal primitive 2N-th roots
    std::vector<LimbConst> limb;   // per-limb reduction constants
    std::vector<double> sf, sf_big;

    // builds everything above; throws std::invalid_argument on unsupported input
    void generate(uint32_t log_n, uint32_t mult_depth, uint32_t scaling_bits, uint32_t first_bits,
                  uint32_t dnum, uint32_t aux_bits, uint32_t extra_bits);

    // psi^bitrev(k) (forward) or psi^-bitrev(k) (inverse) with Shoup companions, N entries each
    void twiddles(uint32_t limb_id, bool inverse, std::vector<u64> &w, std::vector<u64> &w_sh) const;

    // limb id (index into moduli) of slot i of a polynomial with nl Q-limbs (+K P-limbs when with_p)
    uint32_t limb_of(uint32_t i, uint32_t nl) const { return i < nl ? i : L + (i - nl); }
    uint32_t num_parts(uint32_t nl) const {
        uint32_t p = (nl + alpha - 1) / alpha;
        return p > beta ? beta : p;
    }

    // ModUp table of digit `part` of a ciphertext with nl limbs: digit limbs -> complement in Q_nl u P
    BaseConvTable modup_table(uint32_t nl, uint32_t part) const;
    // ModDown table: P -> first nl Q limbs
    BaseConvTable moddown_table(uint32_t nl) const;
    // [P^-1]_{q_i} and [P]_{q_i}
    u64 p_inv_mod(uint32_t limb_id) const;
    u64 p_mod(uint32_t limb_id) const;
    // [q_l^-1]_{q_i}
    u64 q_inv_mod(uint32_t l, uint32_t i) const;
    // LeveledSHECKKSRNS::GetElementForEvalMult
    std::vector<u64> const_factors(uint32_t nl, uint32_t level, double operand) const;
};

}  // namespace mk
