// params.cpp -- see params.hpp.  Host only; no HIP calls.
#include "params.hpp"

#include <cmath>
#include <stdexcept>
#include <string>

namespace mk {

u64 h_mulmod(u64 a, u64 b, u64 m) { return (u64)((u128)a * b % m); }

u64 h_powmod(u64 a, u64 e, u64 m) {
    u64 acc = 1 % m, base = a % m;
    for (; e; e >>= 1) {
        if (e & 1) acc = h_mulmod(acc, base, m);
        base = h_mulmod(base, base, m);
    }
    return acc;
}

u64 h_invmod(u64 a, u64 m) {
    // extended Euclid on signed 128-bit cofactors (m need not be prime)
    __int128 t = 0, nt = 1, r = m, nr = a % m;
    while (nr != 0) {
        __int128 qq = r / nr;
        __int128 tmp = t - qq * nt; t = nt; nt = tmp;
        tmp = r - qq * nr; r = nr; nr = tmp;
    }
    if (r != 1) throw std::invalid_argument("h_invmod: not invertible");
    if (t < 0) t += m;
    return (u64)t;
}

u64 h_shoup(u64 w, u64 q) { return (u64)(((u128)w << 64) / q); }

bool h_is_prime(u64 n) {
    if (n < 4) return n == 2 || n == 3;
    if (!(n & 1)) return false;
    for (u64 p : {3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull})
        if (n % p == 0) return n == p;
    u64 odd = n - 1;
    unsigned twos = 0;
    while (!(odd & 1)) { odd >>= 1; ++twos; }
    // witnesses sufficient for all n < 2^64
    for (u64 a : {2ull, 325ull, 9375ull, 28178ull, 450775ull, 9780504ull, 1795265022ull}) {
        u64 x = h_powmod(a % n, odd, n);
        if (a % n == 0 || x == 1 || x == n - 1) continue;
        bool witness = true;
        for (unsigned s = 1; s < twos && witness; ++s) {
            x = h_mulmod(x, x, n);
            if (x == n - 1) witness = false;
        }
        if (witness) return false;
    }
    return true;
}

// OpenFHE's RootOfUnity returns the smallest primitive root of the requested order.
// `order` is a power of two here: x^((q-1)/order) is primitive iff its (order/2)-th
// power is -1; all primitive roots are its odd powers.
u64 h_min_primitive_root(u64 order, u64 q) {
    if ((q - 1) % order) throw std::invalid_argument("modulus is not 1 mod 2N");
    u64 cof = (q - 1) / order, g = 0;
    for (u64 x = 2; !g; ++x) {
        u64 cand = h_powmod(x, cof, q);
        if (h_powmod(cand, order >> 1, q) == q - 1) g = cand;
    }
    u64 step = h_mulmod(g, g, q), walk = g, smallest = g;
    for (u64 i = 1; i < (order >> 1); ++i) {
        walk = h_mulmod(walk, step, q);
        if (walk < smallest) smallest = walk;
    }
    return smallest;
}

namespace {

// walks primes congruent to 1 modulo `step` (the cyclotomic order 2N)
struct PrimeWalk {
    u64 step;
    u64 down(u64 from) const {  // PreviousPrime
        u64 c = from - step;
        while (!h_is_prime(c)) c -= step;
        return c;
    }
    u64 up(u64 from) const {  // NextPrime
        u64 c = from + step;
        while (!h_is_prime(c)) c += step;
        return c;
    }
    u64 first_at(uint32_t bits) const {  // FirstPrime: starts at 2^bits + step + 1 when step | 2^bits
        u64 base = 1ull << bits;
        u64 c = base + (step - base % step) + 1;
        while (!h_is_prime(c)) c += step;
        return c;
    }
    u64 last_below(uint32_t bits) const {  // LastPrime
        u64 c = 1ull << bits;
        u64 r = c % step;
        if (r == 0) c -= step;
        c = c - r + 1;
        while (!h_is_prime(c)) c -= step;
        return c;
    }
};

uint32_t bit_length_of_product(const u64 *m, uint32_t cnt) {
    std::vector<u64> acc{1};
    for (uint32_t i = 0; i < cnt; ++i) {
        u64 carry = 0;
        for (auto &word : acc) {
            u128 t = (u128)word * m[i] + carry;
            word = (u64)t;
            carry = (u64)(t >> 64);
        }
        if (carry) acc.push_back(carry);
    }
    return (uint32_t)(acc.size() - 1) * 64 + (64 - (uint32_t)__builtin_clzll(acc.back()));
}

bool contains(const std::vector<u64> &v, size_t from, size_t to, u64 x) {
    for (size_t i = from; i < to; ++i)
        if (v[i] == x) return true;
    return false;
}

uint32_t reverse_bits(uint32_t x, uint32_t bits) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < bits; ++i, x >>= 1) r = (r << 1) | (x & 1);
    return r;
}

u64 product_mod(const std::vector<u64> &mods, const std::vector<uint32_t> &ids, int skip, u64 m) {
    u64 acc = 1 % m;
    for (size_t k = 0; k < ids.size(); ++k)
        if ((int)k != skip) acc = h_mulmod(acc, mods[ids[k]] % m, m);
    return acc;
}

}  // namespace

void ParamSet::generate(uint32_t log_n_, uint32_t depth, uint32_t sbits, uint32_t fbits, uint32_t dn,
                        uint32_t abits, uint32_t ebits) {
    if (log_n_ < 8 || log_n_ > 17) throw std::invalid_argument("log_n must be in [8,17]");
    if (depth < 1 || depth > 30) throw std::invalid_argument("mult_depth must be in [1,30]");
    if (sbits < 20 || sbits > 59 || fbits <= sbits || fbits > 60)
        throw std::invalid_argument("need 20 <= scaling_bits < first_bits <= 60");
    if (abits < 30 || abits > 60 || ebits < 18 || ebits > 30) throw std::invalid_argument("bad aux/extra bits");
    if (dn < 1) throw std::invalid_argument("dnum must be >= 1");
    log_n = log_n_; n = 1u << log_n_;
    mult_depth = depth; scaling_bits = sbits; first_bits = fbits; dnum = dn; aux_bits = abits; extra_bits = ebits;

    const PrimeWalk walk{2ull * n};
    L = depth + 2;  // depth+1 scaling levels + the FLEXIBLEAUTOEXT extra limb
    moduli.assign(L, 0);
    const uint32_t body = L - 1;  // limbs 0..body-1 are the classic FLEXIBLEAUTO chain

    // the chain is filled from the top: q_{body-1} first, then alternately just
    // below / just above the running scaling factor so that sf stays near 2^sbits
    moduli[body - 1] = walk.first_at(sbits);
    double running = (double)moduli[body - 1];
    for (int i = (int)body - 2, turn = 0; i >= 1; --i, ++turn) {
        running = running * running / (double)moduli[i + 1];
        u64 centre = (u64)std::llround(running);
        u64 aligned = centre - centre % walk.step + 1;  // = 1 mod 2N, <= centre
        u64 pick;
        if (turn % 2 == 0) {
            pick = aligned - walk.step;
            do pick = walk.down(pick); while (contains(moduli, i + 1, body, pick));
        } else {
            pick = aligned + walk.step;
            do pick = walk.up(pick); while (contains(moduli, i + 1, body, pick));
        }
        moduli[i] = pick;
    }
    moduli[0] = walk.last_below(fbits);
    moduli[L - 1] = walk.first_at(ebits - 1);

    // HYBRID key switching: digits of alpha limbs, K special primes of abits bits
    alpha = (L + dn - 1) / dn;
    beta = (L + alpha - 1) / alpha;
    if ((int)L - (int)(alpha * (beta - 1)) <= 0) throw std::invalid_argument("dnum does not partition Q");
    uint32_t widest = 0;
    for (uint32_t j = 0; j < beta; ++j) {
        uint32_t lo = j * alpha, hi = std::min(L, lo + alpha);
        widest = std::max(widest, bit_length_of_product(&moduli[lo], hi - lo));
    }
    K = (widest + abits - 1) / abits;
    D = L + K;
    u64 cursor = walk.first_at(abits);
    for (uint32_t i = 0; i < K; ++i) {
        do cursor = walk.down(cursor); while (contains(moduli, 0, L, cursor));
        moduli.push_back(cursor);
    }
    if (alpha > 8 || K > 8) throw std::invalid_argument("digit size / #special primes above 8 unsupported");

    roots.resize(D);
    limb.resize(D);
    for (uint32_t i = 0; i < D; ++i) {
        u64 q = moduli[i];
        roots[i] = h_min_primitive_root(walk.step, q);
        LimbConst &c = limb[i];
        c.q = q; c.q2 = 2 * q;
        c.k = 64 - (uint32_t)__builtin_clzll(q);
        c.sh = c.k - 2;
        c.mu = (u64)(((u128)1 << (62 + c.k)) / q);
        c.c64 = (u64)(((u128)1 << 64) % q);
        c.ninv = h_invmod(n % q, q);
        c.ninv_sh = h_shoup(c.ninv, q);
        c.qd = (double)q;
        c.qinv = (double)(1.0L / (long double)q);
        c.ninv_d = (double)c.ninv;
        c.ninv_qd = (double)((long double)c.ninv / (long double)q);
        c.fp = 0;  // the engine decides (needs both passes on the radix kernels)
        c.pm = 0;  // the engine decides (integer limbs of the form 2^k - c)
        c.pm_c = 0;
        c.pad_ = 0;
    }

    // scaling factors (FLEXIBLEAUTOEXT): sf[0] = extra limb, sf[1] = q_{L-2}, then sf^2/q going down
    sf.assign(L, 0.0);
    sf_big.assign(L, 0.0);
    sf[0] = (double)moduli[L - 1];
    sf[1] = (double)moduli[L - 2];
    for (uint32_t k = 2; k < L; ++k) sf[k] = sf[k - 1] * sf[k - 1] / (double)moduli[L - k];
    sf_big[0] = sf[0] * sf[1];
    for (uint32_t k = 1; k + 1 < L; ++k) sf_big[k] = sf[k] * sf[k];
}

void ParamSet::twiddles(uint32_t id, bool inverse, std::vector<u64> &w, std::vector<u64> &w_sh) const {
    u64 q = moduli[id];
    u64 base = inverse ? h_invmod(roots[id], q) : roots[id];
    w.assign(n, 0);
    w_sh.assign(n, 0);
    u64 pw = 1;
    for (uint32_t e = 0; e < n; ++e) {
        w[reverse_bits(e, log_n)] = pw;
        pw = h_mulmod(pw, base, q);
    }
    for (uint32_t e = 0; e < n; ++e) w_sh[e] = h_shoup(w[e], q);
}

static BaseConvTable make_table(const std::vector<u64> &mods, std::vector<uint32_t> src, std::vector<uint32_t> dst) {
    BaseConvTable t;
    t.src = std::move(src);
    t.dst = std::move(dst);
    const size_t ni = t.src.size(), no = t.dst.size();
    t.hatinv.resize(ni);
    t.hatinv_sh.resize(ni);
    t.hat.resize(ni * no);
    for (size_t i = 0; i < ni; ++i) {
        u64 si = mods[t.src[i]];
        t.hatinv[i] = h_invmod(product_mod(mods, t.src, (int)i, si), si);
        t.hatinv_sh[i] = h_shoup(t.hatinv[i], si);
        for (size_t j = 0; j < no; ++j) t.hat[i * no + j] = product_mod(mods, t.src, (int)i, mods[t.dst[j]]);
    }
    return t;
}

BaseConvTable ParamSet::modup_table(uint32_t nl, uint32_t part) const {
    uint32_t lo = part * alpha, hi = std::min(nl, lo + alpha);
    std::vector<uint32_t> src, dst;
    for (uint32_t i = lo; i < hi; ++i) src.push_back(i);
    for (uint32_t i = 0; i < nl + K; ++i)
        if (i < lo || i >= hi) dst.push_back(limb_of(i, nl));
    return make_table(moduli, src, dst);
}

BaseConvTable ParamSet::moddown_table(uint32_t nl) const {
    std::vector<uint32_t> src, dst;
    for (uint32_t k = 0; k < K; ++k) src.push_back(L + k);
    for (uint32_t i = 0; i < nl; ++i) dst.push_back(i);
    return make_table(moduli, src, dst);
}

u64 ParamSet::p_mod(uint32_t id) const {
    u64 q = moduli[id], acc = 1;
    for (uint32_t k = 0; k < K; ++k) acc = h_mulmod(acc, moduli[L + k] % q, q);
    return acc;
}
u64 ParamSet::p_inv_mod(uint32_t id) const { return h_invmod(p_mod(id), moduli[id]); }
u64 ParamSet::q_inv_mod(uint32_t l, uint32_t i) const { return h_invmod(moduli[l] % moduli[i], moduli[i]); }

std::vector<u64> ParamSet::const_factors(uint32_t nl, uint32_t level, double operand) const {
    // EvalMult(ct, double): integer constant = trunc(operand * sf(level) + 0.5) as a 128-bit integer
    double scale = sf.at(level);
    int log_sf = (int)std::ceil(std::log2(std::fabs(scale)));
    int log_approx = log_sf > 125 ? log_sf - 125 : 0;
    __int128 big = (__int128)(operand / std::pow(2.0, log_approx) * scale + 0.5);
    std::vector<u64> out(nl);
    for (uint32_t i = 0; i < nl; ++i) {
        __int128 m = (__int128)moduli[i];
        __int128 r = big % m;
        if (r < 0) r += m;
        u64 f = (u64)r;
        if (log_approx > 0) f = h_mulmod(f, h_powmod(2, (u64)log_approx, moduli[i]), moduli[i]);
        out[i] = f;
    }
    return out;
}

}  // namespace mk
