/*
 * mkckks_oracle.c -- CPU restatement (plain C + OpenMP) of the multikey-CKKS
 * PRE + aggregation path.  TEST INFRASTRUCTURE ONLY -- see mkckks_oracle.h.
 *
 * Every function names the OpenFHE routine it restates ([upstream], file names
 * only: the dependency is absent from /root/reference) and the reference call
 * site that reaches it.
 */
#include "mkckks_oracle.h"

#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;
typedef __int128 i128;

/* ------------------------------------------------------------------ */
/* scalar modular arithmetic  ([upstream] math/hal/intnat/ubintnat.h)  */
/* ------------------------------------------------------------------ */

static inline u64 mulmod(u64 a, u64 b, u64 m) { return (u64)((u128)a * b % m); }
static inline u64 addmod(u64 a, u64 b, u64 m) { u64 s = a + b; return s >= m ? s - m : s; }
static inline u64 submod(u64 a, u64 b, u64 m) { return a >= b ? a - b : a + m - b; }

static u64 powmod(u64 a, u64 e, u64 m) {
    u64 r = 1 % m;
    a %= m;
    while (e) {
        if (e & 1) r = mulmod(r, a, m);
        a = mulmod(a, a, m);
        e >>= 1;
    }
    return r;
}
static u64 invmod(u64 a, u64 m) { return powmod(a % m, m - 2, m); } /* m prime */

/* NativeIntegerT::PrepModMulConst / ModMulFastConst (Shoup) */
static inline u64 shoup_pre(u64 w, u64 m) { return (u64)(((u128)w << 64) / m); }
static inline u64 mulmod_shoup(u64 a, u64 w, u64 wpre, u64 m) {
    u64 q = (u64)(((u128)a * wpre) >> 64);
    u64 r = a * w - q * m;
    return r >= m ? r - m : r;
}

/* [upstream] nbtheory: MillerRabinPrimalityTest (deterministic bases here) */
int orc_is_prime(u64 n) {
    static const u64 bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    if (n < 2) return 0;
    for (unsigned i = 0; i < 12; i++) {
        if (n == bases[i]) return 1;
        if (n % bases[i] == 0) return 0;
    }
    u64 d = n - 1;
    int r = 0;
    while (!(d & 1)) { d >>= 1; r++; }
    for (unsigned i = 0; i < 12; i++) {
        u64 x = powmod(bases[i], d, n);
        if (x == 1 || x == n - 1) continue;
        int comp = 1;
        for (int k = 1; k < r; k++) {
            x = mulmod(x, x, n);
            if (x == n - 1) { comp = 0; break; }
        }
        if (comp) return 0;
    }
    return 1;
}

/* [upstream] nbtheory: FirstPrime(nBits, m): first candidate 2^nBits + (m - r) + 1 */
static u64 first_prime(uint32_t bits, u64 m) {
    u64 base = (u64)1 << bits;
    u64 r = base % m;
    u64 q = base + (m - r) + 1;
    while (!orc_is_prime(q)) q += m;
    return q;
}
/* [upstream] nbtheory: LastPrime(nBits, m): largest prime < 2^nBits, = 1 mod m */
static u64 last_prime(uint32_t bits, u64 m) {
    u64 q = (u64)1 << bits;
    u64 r = q % m;
    if (r < 1) q -= m;
    q -= r;
    q += 1;
    while (!orc_is_prime(q)) q -= m;
    return q;
}
static u64 previous_prime(u64 q, u64 m) {
    u64 x = q - m;
    while (!orc_is_prime(x)) x -= m;
    return x;
}
static u64 next_prime(u64 q, u64 m) {
    u64 x = q + m;
    while (!orc_is_prime(x)) x += m;
    return x;
}

/* [upstream] nbtheory: RootOfUnity(m, q) returns the MINIMUM primitive m-th
 * root.  m is a power of two here, so g is primitive iff
 * g^(m/2) = -1, and the primitive roots are the odd powers of g. */
u64 orc_min_root_of_unity(u64 m, u64 q) {
    u64 g = 0;
    for (u64 x = 2; x < q; x++) {
        u64 c = powmod(x, (q - 1) / m, q);
        if (powmod(c, m / 2, q) == q - 1) { g = c; break; }
    }
    u64 g2 = mulmod(g, g, q), cur = g, best = g;
    for (u64 k = 1; k < m / 2; k++) {
        cur = mulmod(cur, g2, q);
        if (cur < best) best = cur;
    }
    return best;
}

/* ------------------------------------------------------------------ */
/* context                                                             */
/* ------------------------------------------------------------------ */

struct orc_ctx {
    uint32_t logn, n, L, K, D, alpha, beta, depth;
    u64 *mod;   /* D moduli: q_0..q_{L-1}, p_0..p_{K-1} */
    u64 *psi;   /* minimal primitive 2N-th roots */
    u64 *ninv;  /* N^-1 mod m */
    u64 **tw, **twp, **itw, **itwp; /* bit-reversed psi powers + Shoup companions */
    double *sf, *sf_big;
    /* encode/decode tables */
    uint32_t *rot_group;
    double *ksi_re, *ksi_im;
};

static uint32_t bitrev(uint32_t x, uint32_t bits) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

/* bit length of a product of up to 16 word-size moduli (small bignum) */
static uint32_t product_bits(const u64 *m, uint32_t cnt) {
    u64 acc[20] = {1};
    uint32_t len = 1;
    for (uint32_t i = 0; i < cnt; i++) {
        u64 carry = 0;
        for (uint32_t k = 0; k < len; k++) {
            u128 t = (u128)acc[k] * m[i] + carry;
            acc[k] = (u64)t;
            carry = (u64)(t >> 64);
        }
        if (carry) acc[len++] = carry;
    }
    uint32_t top = 64 - (uint32_t)__builtin_clzll(acc[len - 1]);
    return (len - 1) * 64 + top;
}

/*
 * [upstream] ckksrns-parametergeneration.cpp ParameterGenerationCKKSRNS::ParamsGenCKKSRNSInternal
 * (FLEXIBLEAUTOEXT branch) + rns-cryptoparameters.cpp PrecomputeCRTTables (HYBRID:
 * digit partition, special primes P) + ckksrns-cryptoparameters.cpp (scaling factors).
 * Reached from server/src/genCC.cpp:68 (GenCryptoContext).
 */
orc_ctx *orc_ctx_new(uint32_t log_n, uint32_t mult_depth, uint32_t scaling_bits,
                     uint32_t first_bits, uint32_t dnum, uint32_t aux_bits,
                     uint32_t extra_bits) {
    orc_ctx *c = (orc_ctx *)calloc(1, sizeof(*c));
    c->logn = log_n;
    c->n = 1u << log_n;
    c->depth = mult_depth;
    const u64 M = 2ull * c->n; /* cyclotomic order */
    const uint32_t L = mult_depth + 2;
    c->L = L;
    u64 *q = (u64 *)calloc(L + 64, sizeof(u64));

    /* scaling primes alternate around 2^scaling_bits so that sf stays close to it */
    const uint32_t vec = L - 1; /* limbs without the extra one */
    q[vec - 1] = first_prime(scaling_bits, M);
    {
        double sf = (double)q[vec - 1];
        uint32_t cnt = 0;
        for (int32_t i = (int32_t)vec - 2; i >= 1; i--) {
            sf = sf * sf / (double)q[i + 1];
            u64 sfi = (u64)llround(sf);
            u64 rem = sfi % M;
            if ((cnt & 1) == 0) {
                u64 cand = sfi - M - rem + 1;
                for (;;) {
                    cand = previous_prime(cand, M);
                    int same = 0;
                    for (uint32_t j = i + 1; j < vec; j++) same |= (cand == q[j]);
                    if (!same) break;
                }
                q[i] = cand;
            } else {
                u64 cand = sfi + M - rem + 1;
                for (;;) {
                    cand = next_prime(cand, M);
                    int same = 0;
                    for (uint32_t j = i + 1; j < vec; j++) same |= (cand == q[j]);
                    if (!same) break;
                }
                q[i] = cand;
            }
            cnt++;
        }
    }
    q[0] = last_prime(first_bits, M);
    q[L - 1] = first_prime(extra_bits - 1, M); /* the FLEXIBLEAUTOEXT extra limb */

    /* HYBRID digit partition */
    c->alpha = (L + dnum - 1) / dnum;
    c->beta = (L + c->alpha - 1) / c->alpha;
    uint32_t max_bits = 0;
    for (uint32_t j = 0; j < c->beta; j++) {
        uint32_t lo = j * c->alpha, hi = lo + c->alpha > L ? L : lo + c->alpha;
        uint32_t b = product_bits(q + lo, hi - lo);
        if (b > max_bits) max_bits = b;
    }
    c->K = (max_bits + aux_bits - 1) / aux_bits;
    c->D = L + c->K;
    {
        u64 prev = first_prime(aux_bits, M);
        for (uint32_t i = 0; i < c->K; i++) {
            u64 p;
            int in_q;
            do {
                p = previous_prime(prev, M);
                in_q = 0;
                for (uint32_t j = 0; j < L; j++) in_q |= (p == q[j]);
                prev = p;
            } while (in_q);
            q[L + i] = p;
        }
    }
    c->mod = q;

    const uint32_t D = c->D, n = c->n;
    c->psi = (u64 *)calloc(D, sizeof(u64));
    c->ninv = (u64 *)calloc(D, sizeof(u64));
    c->tw = (u64 **)calloc(D, sizeof(u64 *));
    c->twp = (u64 **)calloc(D, sizeof(u64 *));
    c->itw = (u64 **)calloc(D, sizeof(u64 *));
    c->itwp = (u64 **)calloc(D, sizeof(u64 *));
#pragma omp parallel for schedule(dynamic)
    for (uint32_t i = 0; i < D; i++) {
        u64 m = q[i];
        u64 psi = orc_min_root_of_unity(M, m);
        u64 ipsi = invmod(psi, m);
        c->psi[i] = psi;
        c->ninv[i] = invmod(n % m, m);
        c->tw[i] = (u64 *)malloc(n * sizeof(u64));
        c->twp[i] = (u64 *)malloc(n * sizeof(u64));
        c->itw[i] = (u64 *)malloc(n * sizeof(u64));
        c->itwp[i] = (u64 *)malloc(n * sizeof(u64));
        u64 pw = 1, ipw = 1;
        for (uint32_t k = 0; k < n; k++) {
            uint32_t r = bitrev(k, log_n);
            c->tw[i][r] = pw;
            c->itw[i][r] = ipw;
            pw = mulmod(pw, psi, m);
            ipw = mulmod(ipw, ipsi, m);
        }
        for (uint32_t k = 0; k < n; k++) {
            c->twp[i][k] = shoup_pre(c->tw[i][k], m);
            c->itwp[i][k] = shoup_pre(c->itw[i][k], m);
        }
    }

    /* scaling factors: ckksrns-cryptoparameters.cpp, FLEXIBLEAUTOEXT */
    c->sf = (double *)calloc(L, sizeof(double));
    c->sf_big = (double *)calloc(L, sizeof(double));
    c->sf[0] = (double)q[L - 1];
    c->sf[1] = (double)q[L - 2];
    for (uint32_t k = 2; k < L; k++) {
        double prev = c->sf[k - 1];
        c->sf[k] = prev * prev / (double)q[L - k];
    }
    c->sf_big[0] = c->sf[0] * c->sf[1];
    for (uint32_t k = 1; k + 1 < L; k++) c->sf_big[k] = c->sf[k] * c->sf[k];

    /* encoding tables: [upstream] dftransform.cpp (rotation group 5^j mod 2N) */
    uint32_t slots = n / 2;
    c->rot_group = (uint32_t *)malloc(slots * sizeof(uint32_t));
    u64 five = 1;
    for (uint32_t j = 0; j < slots; j++) {
        c->rot_group[j] = (uint32_t)five;
        five = five * 5 % M;
    }
    c->ksi_re = (double *)malloc((M + 1) * sizeof(double));
    c->ksi_im = (double *)malloc((M + 1) * sizeof(double));
    for (u64 k = 0; k <= M; k++) {
        double ang = 2.0 * M_PI * (double)k / (double)M;
        c->ksi_re[k] = cos(ang);
        c->ksi_im[k] = sin(ang);
    }
    return c;
}

void orc_ctx_free(orc_ctx *c) {
    if (!c) return;
    for (uint32_t i = 0; i < c->D; i++) {
        free(c->tw[i]); free(c->twp[i]); free(c->itw[i]); free(c->itwp[i]);
    }
    free(c->tw); free(c->twp); free(c->itw); free(c->itwp);
    free(c->mod); free(c->psi); free(c->ninv); free(c->sf); free(c->sf_big);
    free(c->rot_group); free(c->ksi_re); free(c->ksi_im);
    free(c);
}

void orc_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
uint32_t orc_ring_dim(const orc_ctx *c) { return c->n; }
uint32_t orc_num_q(const orc_ctx *c) { return c->L; }
uint32_t orc_num_p(const orc_ctx *c) { return c->K; }
uint32_t orc_alpha(const orc_ctx *c) { return c->alpha; }
uint32_t orc_beta(const orc_ctx *c) { return c->beta; }
void orc_moduli(const orc_ctx *c, u64 *out) { memcpy(out, c->mod, c->D * sizeof(u64)); }
void orc_roots(const orc_ctx *c, u64 *out) { memcpy(out, c->psi, c->D * sizeof(u64)); }
double orc_sf(const orc_ctx *c, uint32_t level) { return c->sf[level]; }
double orc_sf_big(const orc_ctx *c, uint32_t level) { return c->sf_big[level]; }

/* ------------------------------------------------------------------ */
/* NTT  ([upstream] transformnat-impl.h ChineseRemainderTransformFTTNat) */
/* ------------------------------------------------------------------ */

/* ForwardTransformToBitReverseInPlace: Cooley-Tukey, natural in, bit-reversed out */
void orc_ntt_fwd(const orc_ctx *c, uint32_t limb, u64 *a) {
    const u64 m = c->mod[limb];
    const u64 *tw = c->tw[limb], *twp = c->twp[limb];
    uint32_t n = c->n, t = n;
    for (uint32_t mm = 1; mm < n; mm <<= 1) {
        t >>= 1;
        for (uint32_t i = 0; i < mm; i++) {
            u64 w = tw[mm + i], wp = twp[mm + i];
            uint32_t j1 = 2 * i * t;
            for (uint32_t j = j1; j < j1 + t; j++) {
                u64 u = a[j];
                u64 v = mulmod_shoup(a[j + t], w, wp, m);
                a[j] = addmod(u, v, m);
                a[j + t] = submod(u, v, m);
            }
        }
    }
}

/* InverseTransformFromBitReverseInPlace: Gentleman-Sande, bit-reversed in, natural out, then * N^-1 */
void orc_ntt_inv(const orc_ctx *c, uint32_t limb, u64 *a) {
    const u64 m = c->mod[limb];
    const u64 *tw = c->itw[limb], *twp = c->itwp[limb];
    uint32_t n = c->n, t = 1;
    for (uint32_t mm = n; mm > 1; mm >>= 1) {
        uint32_t h = mm >> 1, j1 = 0;
        for (uint32_t i = 0; i < h; i++) {
            u64 w = tw[h + i], wp = twp[h + i];
            for (uint32_t j = j1; j < j1 + t; j++) {
                u64 u = a[j], v = a[j + t];
                a[j] = addmod(u, v, m);
                a[j + t] = mulmod_shoup(submod(u, v, m), w, wp, m);
            }
            j1 += 2 * t;
        }
        t <<= 1;
    }
    u64 ni = c->ninv[limb], nip = shoup_pre(ni, m);
    for (uint32_t j = 0; j < n; j++) a[j] = mulmod_shoup(a[j], ni, nip, m);
}

/* ------------------------------------------------------------------ */
/* coefficient-wise ops                                                */
/* ------------------------------------------------------------------ */

/* [upstream] LeveledSHEBase::EvalAddCore -> DCRTPoly::operator+= (aggregateEncryptedWeights.cpp:82) */
void orc_eval_add(const orc_ctx *c, uint32_t nl, const u64 *a, const u64 *b, u64 *out) {
    const uint32_t n = c->n;
#pragma omp parallel for
    for (uint32_t k = 0; k < 2 * nl; k++) {
        u64 m = c->mod[k % nl];
        const u64 *x = a + (size_t)k * n, *y = b + (size_t)k * n;
        u64 *o = out + (size_t)k * n;
        for (uint32_t j = 0; j < n; j++) o[j] = addmod(x[j], y[j], m);
    }
}

/* [upstream] dcrtpoly-impl.h DropLastElementAndScale, via
 * LeveledSHECKKSRNS::ModReduceInternalInPlace (aggregateEncryptedWeights.cpp:83,
 * first half of EvalMult(ct,double) when noiseScaleDeg==2).
 * c'_i = (c_i - [c_last]_centred) * q_last^-1 mod q_i; the centred lift treats
 * v > floor(q_last/2) as negative (NativeVectorT::SwitchModulus). */
void orc_rescale(const orc_ctx *c, uint32_t nl, const u64 *in, u64 *out) {
    const uint32_t n = c->n, last = nl - 1;
    const u64 ql = c->mod[last], half = ql >> 1;
    for (uint32_t k = 0; k < 2; k++) {
        u64 *lp = (u64 *)malloc(n * sizeof(u64));
        memcpy(lp, in + ((size_t)k * nl + last) * n, n * sizeof(u64));
        orc_ntt_inv(c, last, lp);
#pragma omp parallel for
        for (uint32_t i = 0; i < last; i++) {
            u64 qi = c->mod[i];
            u64 qlinv = invmod(ql % qi, qi), qlinvp = shoup_pre(qlinv, qi);
            u64 *tmp = (u64 *)malloc(n * sizeof(u64));
            for (uint32_t j = 0; j < n; j++) {
                u64 v = lp[j];
                /* SwitchModulus: centred representative mod q_i */
                if (v > half) tmp[j] = submod(v % qi, ql % qi, qi);
                else tmp[j] = v % qi;
            }
            orc_ntt_fwd(c, i, tmp);
            const u64 *src = in + ((size_t)k * nl + i) * n;
            u64 *dst = out + ((size_t)k * last + i) * n;
            for (uint32_t j = 0; j < n; j++)
                dst[j] = mulmod_shoup(submod(src[j], tmp[j], qi), qlinv, qlinvp, qi);
            free(tmp);
        }
        free(lp);
    }
}

/* [upstream] ckksrns-leveledshe.cpp LeveledSHECKKSRNS::GetElementForEvalMult
 * (aggregateEncryptedWeights.cpp:83: EvalMult(ct_sum, 0.5)).  HAVE_INT128 path:
 * large = (int128)(operand * sf(level) + 0.5), reduced per limb. */
void orc_const_factors(const orc_ctx *c, uint32_t nl, uint32_t level, double operand, u64 *factors) {
    double sc = c->sf[level];
    int32_t log_sf = (int32_t)ceil(log2(fabs(sc)));
    int32_t log_valid = log_sf <= 125 ? log_sf : 125;
    int32_t log_approx = log_sf - log_valid;
    double approx = pow(2.0, log_approx);
    i128 large = (i128)(operand / approx * sc + 0.5);
    for (uint32_t i = 0; i < nl; i++) {
        i128 m = (i128)c->mod[i];
        i128 r = large % m;
        if (r < 0) r += m;
        u64 f = (u64)r;
        if (log_approx > 0) { /* scale back up by 2^logApprox inside the CRT */
            u64 two = powmod(2, (u64)log_approx, c->mod[i]);
            f = mulmod(f, two, c->mod[i]);
        }
        factors[i] = f;
    }
}

/* [upstream] LeveledSHECKKSRNS::EvalMultCoreInPlace: cv[k] = cv[k] * factors */
void orc_mult_factors(const orc_ctx *c, uint32_t nl, const u64 *factors, u64 *ct) {
    const uint32_t n = c->n;
#pragma omp parallel for
    for (uint32_t k = 0; k < 2 * nl; k++) {
        u64 m = c->mod[k % nl], f = factors[k % nl], fp = shoup_pre(f, m);
        u64 *x = ct + (size_t)k * n;
        for (uint32_t j = 0; j < n; j++) x[j] = mulmod_shoup(x[j], f, fp, m);
    }
}

/* ------------------------------------------------------------------ */
/* hybrid key switching                                                */
/* ------------------------------------------------------------------ */

/* [upstream] dcrtpoly-impl.h ApproxSwitchCRTBasis: COEFFICIENT-format x over
 * `src` moduli -> y over `dst` moduli,
 *   y_j = sum_i [x_i * (S/s_i)^-1]_{s_i} * [(S/s_i)]_{d_j}  mod d_j
 * with a 128-bit accumulator and one reduction at the end. */
static void approx_switch_basis(uint32_t n, uint32_t nin, const u64 *src, u64 *const *x,
                                uint32_t nout, const u64 *dst, u64 *const *y) {
    u64 hatinv[32], hatinvp[32];
    u64 *hat = (u64 *)malloc((size_t)nin * nout * sizeof(u64));
    for (uint32_t i = 0; i < nin; i++) {
        u64 h = 1;
        for (uint32_t k = 0; k < nin; k++)
            if (k != i) h = mulmod(h, src[k] % src[i], src[i]);
        hatinv[i] = invmod(h, src[i]);
        hatinvp[i] = shoup_pre(hatinv[i], src[i]);
        for (uint32_t j = 0; j < nout; j++) {
            u64 g = 1;
            for (uint32_t k = 0; k < nin; k++)
                if (k != i) g = mulmod(g, src[k] % dst[j], dst[j]);
            hat[i * nout + j] = g;
        }
    }
#pragma omp parallel for
    for (uint32_t r = 0; r < n; r++) {
        u64 t[32];
        for (uint32_t i = 0; i < nin; i++) t[i] = mulmod_shoup(x[i][r], hatinv[i], hatinvp[i], src[i]);
        for (uint32_t j = 0; j < nout; j++) {
            u128 acc = 0;
            for (uint32_t i = 0; i < nin; i++) acc += (u128)t[i] * hat[i * nout + j];
            y[j][r] = (u64)(acc % dst[j]);
        }
    }
    free(hat);
}

/* [upstream] keyswitch-hybrid.cpp KeySwitchHYBRID::EvalKeySwitchPrecomputeCore
 * (changeCipherDomain.cpp:74 -> ReEncrypt -> KeySwitchInPlace -> KeySwitchCore).
 * c1: u64[nl][N] EVALUATION.  digits: u64[nparts][nl+K][N] EVALUATION over
 * Q_l P, own limbs copied, the others ModUp-converted. */
uint32_t orc_modup_digits(const orc_ctx *c, uint32_t nl, const u64 *c1, u64 *digits) {
    const uint32_t n = c->n, K = c->K, L = c->L, alpha = c->alpha;
    uint32_t nparts = (nl + alpha - 1) / alpha;
    if (nparts > c->beta) nparts = c->beta;
    const uint32_t ext = nl + K;
    for (uint32_t part = 0; part < nparts; part++) {
        uint32_t lo = part * alpha, hi = lo + alpha > nl ? nl : lo + alpha;
        uint32_t sz = hi - lo, ncomp = ext - sz;
        u64 *dig = digits + (size_t)part * ext * n;
        u64 *coef = (u64 *)malloc((size_t)sz * n * sizeof(u64));
        u64 *x[32], *y[64], src[32], dst[64];
#pragma omp parallel for
        for (uint32_t i = 0; i < sz; i++) {
            memcpy(coef + (size_t)i * n, c1 + (size_t)(lo + i) * n, n * sizeof(u64));
            orc_ntt_inv(c, lo + i, coef + (size_t)i * n);
        }
        for (uint32_t i = 0; i < sz; i++) { x[i] = coef + (size_t)i * n; src[i] = c->mod[lo + i]; }
        uint32_t idx[64], w = 0;
        for (uint32_t i = 0; i < ext; i++) {
            if (i >= lo && i < hi) continue;
            idx[w] = i < nl ? i : L + (i - nl); /* limb index in the QP table */
            dst[w] = c->mod[idx[w]];
            y[w] = dig + (size_t)i * n;
            w++;
        }
        approx_switch_basis(n, sz, src, x, ncomp, dst, y);
#pragma omp parallel for
        for (uint32_t k = 0; k < ncomp; k++) orc_ntt_fwd(c, idx[k], y[k]);
        for (uint32_t i = lo; i < hi; i++)
            memcpy(dig + (size_t)i * n, c1 + (size_t)i * n, n * sizeof(u64));
        free(coef);
    }
    return nparts;
}

/* [upstream] dcrtpoly-impl.h ApproxModDown (t = 0, CKKS):
 * out_i = (in_i - NTT(BaseConv_{P->Q_l}(INTT(in_P))_i)) * P^-1 mod q_i. */
void orc_moddown(const orc_ctx *c, uint32_t nl, const u64 *in, u64 *out) {
    const uint32_t n = c->n, K = c->K, L = c->L;
    u64 *pc = (u64 *)malloc((size_t)K * n * sizeof(u64));
    u64 *conv = (u64 *)malloc((size_t)nl * n * sizeof(u64));
    u64 *x[32], *y[64];
#pragma omp parallel for
    for (uint32_t k = 0; k < K; k++) {
        memcpy(pc + (size_t)k * n, in + (size_t)(nl + k) * n, n * sizeof(u64));
        orc_ntt_inv(c, L + k, pc + (size_t)k * n);
    }
    for (uint32_t k = 0; k < K; k++) x[k] = pc + (size_t)k * n;
    for (uint32_t i = 0; i < nl; i++) y[i] = conv + (size_t)i * n;
    approx_switch_basis(n, K, c->mod + L, x, nl, c->mod, y);
#pragma omp parallel for
    for (uint32_t i = 0; i < nl; i++) {
        u64 qi = c->mod[i], pinv = 1;
        for (uint32_t k = 0; k < K; k++) pinv = mulmod(pinv, c->mod[L + k] % qi, qi);
        pinv = invmod(pinv, qi);
        u64 pinvp = shoup_pre(pinv, qi);
        orc_ntt_fwd(c, i, y[i]);
        const u64 *src = in + (size_t)i * n;
        u64 *dst = out + (size_t)i * n;
        for (uint32_t j = 0; j < n; j++)
            dst[j] = mulmod_shoup(submod(src[j], y[i][j], qi), pinv, pinvp, qi);
    }
    free(pc);
    free(conv);
}

/* [upstream] keyswitch-hybrid.cpp EvalFastKeySwitchCoreExt + ApproxModDown x2 +
 * KeySwitchRNS::KeySwitchInPlace; reached as PRE::ReEncrypt(ct, ek, nullptr)
 * (INDCPA: no re-randomisation) from changeCipherDomain.cpp:74,89,105. */
void orc_reencrypt(const orc_ctx *c, uint32_t nl, const u64 *ct, const u64 *evk, u64 *out) {
    const uint32_t n = c->n, K = c->K, L = c->L, D = c->D, ext = nl + K;
    const u64 *c0 = ct, *c1 = ct + (size_t)nl * n;
    u64 *digits = (u64 *)malloc((size_t)c->beta * ext * n * sizeof(u64));
    uint32_t nparts = orc_modup_digits(c, nl, c1, digits);
    u64 *ct0 = (u64 *)calloc((size_t)ext * n, sizeof(u64));
    u64 *ct1 = (u64 *)calloc((size_t)ext * n, sizeof(u64));
#pragma omp parallel for
    for (uint32_t i = 0; i < ext; i++) {
        uint32_t idx = i < nl ? i : L + (i - nl); /* evk limb: skips dropped Q limbs */
        u64 m = c->mod[idx];
        u64 *t0 = ct0 + (size_t)i * n, *t1 = ct1 + (size_t)i * n;
        for (uint32_t j = 0; j < nparts; j++) {
            const u64 *d = digits + ((size_t)j * ext + i) * n;
            const u64 *b = evk + (((size_t)j * 2 + 0) * D + idx) * n;
            const u64 *a = evk + (((size_t)j * 2 + 1) * D + idx) * n;
            for (uint32_t r = 0; r < n; r++) {
                t0[r] = addmod(t0[r], mulmod(d[r], b[r], m), m);
                t1[r] = addmod(t1[r], mulmod(d[r], a[r], m), m);
            }
        }
    }
    u64 *o0 = out, *o1 = out + (size_t)nl * n;
    orc_moddown(c, nl, ct0, o0);
    orc_moddown(c, nl, ct1, o1);
#pragma omp parallel for
    for (uint32_t i = 0; i < nl; i++) {
        u64 m = c->mod[i];
        for (uint32_t r = 0; r < n; r++)
            o0[(size_t)i * n + r] = addmod(o0[(size_t)i * n + r], c0[(size_t)i * n + r], m);
    }
    free(digits); free(ct0); free(ct1);
}

/* ------------------------------------------------------------------ */
/* key generation                                                      */
/* ------------------------------------------------------------------ */

static void signed_to_eval(const orc_ctx *c, uint32_t limb, const int32_t *v, u64 *out) {
    u64 m = c->mod[limb];
    for (uint32_t j = 0; j < c->n; j++) {
        int64_t x = v[j];
        out[j] = x >= 0 ? (u64)x % m : m - ((u64)(-x) % m);
        if (out[j] == m) out[j] = 0;
    }
    orc_ntt_fwd(c, limb, out);
}
static void tern_to_eval(const orc_ctx *c, uint32_t limb, const int8_t *v, u64 *out) {
    u64 m = c->mod[limb];
    for (uint32_t j = 0; j < c->n; j++) out[j] = v[j] == 0 ? 0 : (v[j] > 0 ? 1 : m - 1);
    orc_ntt_fwd(c, limb, out);
}

/* [upstream] rns-pke.cpp PKERNS::KeyGenInternal (keyGen.cpp:33): over QP,
 * b = ns*e - a*s (ns = 1 for CKKS), pk = (b, a). */
void orc_keygen(const orc_ctx *c, const int8_t *s_tern, const u64 *a_eval, const int32_t *e,
                u64 *pk, u64 *sk_eval) {
    const uint32_t n = c->n, D = c->D;
#pragma omp parallel for
    for (uint32_t i = 0; i < D; i++) {
        u64 m = c->mod[i];
        u64 *s = sk_eval + (size_t)i * n;
        u64 *ee = (u64 *)malloc(n * sizeof(u64));
        tern_to_eval(c, i, s_tern, s);
        signed_to_eval(c, i, e, ee);
        const u64 *a = a_eval + (size_t)i * n;
        u64 *b = pk + (size_t)i * n, *pa = pk + ((size_t)D + i) * n;
        for (uint32_t j = 0; j < n; j++) {
            b[j] = submod(ee[j], mulmod(a[j], s[j], m), m);
            pa[j] = a[j];
        }
        free(ee);
    }
}

/* [upstream] keyswitch-hybrid.cpp KeySwitchHYBRID::KeySwitchGenInternal(oldKey,
 * newPublicKey), reached via PRERNS::ReKeyGen (REkeyGen.cpp:52):
 *   b_j[i] = pk0[i]*u_j + e0_j (+ [P]_{q_i} * s_old[i] when limb i is in digit j)
 *   a_j[i] = pk1[i]*u_j + e1_j            over all D limbs of QP. */
void orc_rekeygen(const orc_ctx *c, const int8_t *s_old, const u64 *pk_new, const int8_t *u,
                  const int32_t *e0, const int32_t *e1, u64 *evk) {
    const uint32_t n = c->n, D = c->D, L = c->L, K = c->K;
    for (uint32_t part = 0; part < c->beta; part++) {
        uint32_t lo = part * c->alpha, hi = lo + c->alpha > L ? L : lo + c->alpha;
#pragma omp parallel for
        for (uint32_t i = 0; i < D; i++) {
            u64 m = c->mod[i];
            u64 *ue = (u64 *)malloc(n * sizeof(u64));
            u64 *e0e = (u64 *)malloc(n * sizeof(u64));
            u64 *e1e = (u64 *)malloc(n * sizeof(u64));
            u64 *se = (u64 *)malloc(n * sizeof(u64));
            tern_to_eval(c, i, u + (size_t)part * n, ue);
            signed_to_eval(c, i, e0 + (size_t)part * n, e0e);
            signed_to_eval(c, i, e1 + (size_t)part * n, e1e);
            int own = (i >= lo && i < hi);
            u64 pmod = 1;
            if (own) {
                tern_to_eval(c, i, s_old, se);
                for (uint32_t k = 0; k < K; k++) pmod = mulmod(pmod, c->mod[L + k] % m, m);
            }
            const u64 *p0 = pk_new + (size_t)i * n, *p1 = pk_new + ((size_t)D + i) * n;
            u64 *b = evk + (((size_t)part * 2 + 0) * D + i) * n;
            u64 *a = evk + (((size_t)part * 2 + 1) * D + i) * n;
            for (uint32_t j = 0; j < n; j++) {
                u64 bb = addmod(mulmod(p0[j], ue[j], m), e0e[j], m);
                if (own) bb = addmod(bb, mulmod(pmod, se[j], m), m);
                b[j] = bb;
                a[j] = addmod(mulmod(p1[j], ue[j], m), e1e[j], m);
            }
            free(ue); free(e0e); free(e1e); free(se);
        }
    }
}

/* ------------------------------------------------------------------ */
/* encode / encrypt / decrypt / decode                                 */
/* ------------------------------------------------------------------ */

static void bitrev_complex(double *re, double *im, uint32_t size) {
    for (uint32_t i = 1, j = 0; i < size; i++) {
        uint32_t bit = size >> 1;
        for (; j >= bit; bit >>= 1) j -= bit;
        j += bit;
        if (i < j) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
}

/* [upstream] dftransform.cpp DiscreteFourierTransform::FFTSpecialInv */
static void fft_special_inv(const orc_ctx *c, double *re, double *im, uint32_t size) {
    const uint32_t M = 2 * c->n;
    for (uint32_t len = size; len >= 1; len >>= 1) {
        uint32_t lenh = len >> 1, lenq = len << 2, gap = M / lenq;
        for (uint32_t i = 0; i < size; i += len) {
            for (uint32_t j = 0; j < lenh; j++) {
                uint32_t idx = (lenq - (c->rot_group[j] % lenq)) * gap;
                double ur = re[i + j] + re[i + j + lenh], ui = im[i + j] + im[i + j + lenh];
                double vr = re[i + j] - re[i + j + lenh], vi = im[i + j] - im[i + j + lenh];
                double wr = c->ksi_re[idx], wi = c->ksi_im[idx];
                re[i + j] = ur; im[i + j] = ui;
                re[i + j + lenh] = vr * wr - vi * wi;
                im[i + j + lenh] = vr * wi + vi * wr;
            }
        }
    }
    bitrev_complex(re, im, size);
    for (uint32_t i = 0; i < size; i++) { re[i] /= size; im[i] /= size; }
}

/* [upstream] dftransform.cpp DiscreteFourierTransform::FFTSpecial */
static void fft_special(const orc_ctx *c, double *re, double *im, uint32_t size) {
    const uint32_t M = 2 * c->n;
    bitrev_complex(re, im, size);
    for (uint32_t len = 2; len <= size; len <<= 1) {
        uint32_t lenh = len >> 1, lenq = len << 2, gap = M / lenq;
        for (uint32_t i = 0; i < size; i += len) {
            for (uint32_t j = 0; j < lenh; j++) {
                uint32_t idx = (c->rot_group[j] % lenq) * gap;
                double ur = re[i + j], ui = im[i + j];
                double xr = re[i + j + lenh], xi = im[i + j + lenh];
                double wr = c->ksi_re[idx], wi = c->ksi_im[idx];
                double vr = xr * wr - xi * wi, vi = xr * wi + xi * wr;
                re[i + j] = ur + vr; im[i + j] = ui + vi;
                re[i + j + lenh] = ur - vr; im[i + j + lenh] = ui - vi;
            }
        }
    }
}

/* [upstream] ckkspackedencoding.cpp CKKSPackedEncoding::Encode, full packing
 * (slots = N/2 = batch size, encryptModelWeights.cpp:40,82,109): inverse special
 * FFT, * scale, round, residues per limb, NTT. */
void orc_encode(const orc_ctx *c, const double *vals, uint32_t nvals, double scale, uint32_t nl, u64 *pt) {
    const uint32_t n = c->n, slots = n / 2;
    double *re = (double *)calloc(slots, sizeof(double));
    double *im = (double *)calloc(slots, sizeof(double));
    for (uint32_t i = 0; i < nvals && i < slots; i++) re[i] = vals[i];
    fft_special_inv(c, re, im, slots);
    i128 *coef = (i128 *)malloc((size_t)n * sizeof(i128));
    for (uint32_t i = 0; i < slots; i++) {
        double a = re[i] * scale, b = im[i] * scale;
        coef[i] = (i128)(a + (a >= 0 ? 0.5 : -0.5));
        coef[i + slots] = (i128)(b + (b >= 0 ? 0.5 : -0.5));
    }
#pragma omp parallel for
    for (uint32_t l = 0; l < nl; l++) {
        i128 m = (i128)c->mod[l];
        u64 *o = pt + (size_t)l * n;
        for (uint32_t j = 0; j < n; j++) {
            i128 r = coef[j] % m;
            if (r < 0) r += m;
            o[j] = (u64)r;
        }
        orc_ntt_fwd(c, l, o);
    }
    free(re); free(im); free(coef);
}

/* [upstream] rns-pke.cpp PKERNS::Encrypt / EncryptZeroCore
 * (encryptModelWeights.cpp:83): v ternary, e0/e1 gaussian; uses the first nl
 * Q-limbs of the QP public key. */
void orc_encrypt(const orc_ctx *c, uint32_t nl, const u64 *pk, const u64 *pt, const int8_t *v,
                 const int32_t *e0, const int32_t *e1, u64 *ct) {
    const uint32_t n = c->n, D = c->D;
#pragma omp parallel for
    for (uint32_t i = 0; i < nl; i++) {
        u64 m = c->mod[i];
        u64 *ve = (u64 *)malloc(n * sizeof(u64));
        u64 *e0e = (u64 *)malloc(n * sizeof(u64));
        u64 *e1e = (u64 *)malloc(n * sizeof(u64));
        tern_to_eval(c, i, v, ve);
        signed_to_eval(c, i, e0, e0e);
        signed_to_eval(c, i, e1, e1e);
        const u64 *p0 = pk + (size_t)i * n, *p1 = pk + ((size_t)D + i) * n;
        const u64 *mm = pt + (size_t)i * n;
        u64 *o0 = ct + (size_t)i * n, *o1 = ct + ((size_t)nl + i) * n;
        for (uint32_t j = 0; j < n; j++) {
            o0[j] = addmod(addmod(mulmod(p0[j], ve[j], m), e0e[j], m), mm[j], m);
            o1[j] = addmod(mulmod(p1[j], ve[j], m), e1e[j], m);
        }
        free(ve); free(e0e); free(e1e);
    }
}

/* [upstream] rns-pke.cpp PKERNS::DecryptCore (decryptModelWeights.cpp:81):
 * b = c0 + c1*s in EVALUATION, then COEFFICIENT format. */
void orc_decrypt_core(const orc_ctx *c, uint32_t nl, const u64 *ct, const u64 *sk_eval, u64 *m) {
    const uint32_t n = c->n;
#pragma omp parallel for
    for (uint32_t i = 0; i < nl; i++) {
        u64 q = c->mod[i];
        const u64 *c0 = ct + (size_t)i * n, *c1 = ct + ((size_t)nl + i) * n;
        const u64 *s = sk_eval + (size_t)i * n;
        u64 *o = m + (size_t)i * n;
        for (uint32_t j = 0; j < n; j++) o[j] = addmod(c0[j], mulmod(c1[j], s[j], q), q);
        orc_ntt_inv(c, i, o);
    }
}

/* DecryptCore + CRT interpolation (Garner mixed radix, centred) +
 * [upstream] ckkspackedencoding.cpp CKKSPackedEncoding::Decode without the
 * decode-time noise flooding (decryptModelWeights.cpp:81-83,108-109). */
void orc_decrypt_decode(const orc_ctx *c, uint32_t nl, const u64 *ct, const u64 *sk_eval,
                        double scale, double *out) {
    const uint32_t n = c->n, slots = n / 2;
    u64 *m = (u64 *)malloc((size_t)nl * n * sizeof(u64));
    orc_decrypt_core(c, nl, ct, sk_eval, m);
    /* Garner constants: inv[i] = (q_0...q_{i-1})^-1 mod q_i */
    u64 inv[64];
    for (uint32_t i = 1; i < nl; i++) {
        u64 p = 1;
        for (uint32_t k = 0; k < i; k++) p = mulmod(p, c->mod[k] % c->mod[i], c->mod[i]);
        inv[i] = invmod(p, c->mod[i]);
    }
    double *re = (double *)malloc(slots * sizeof(double));
    double *im = (double *)malloc(slots * sizeof(double));
#pragma omp parallel for
    for (uint32_t j = 0; j < n; j++) {
        u64 v[64];
        v[0] = m[j];
        for (uint32_t i = 1; i < nl; i++) {
            u64 qi = c->mod[i];
            /* acc = v_0 + v_1 q_0 + ... mod q_i via Horner from the top digit */
            u64 acc = v[i - 1] % qi;
            for (int32_t k = (int32_t)i - 2; k >= 0; k--)
                acc = addmod(mulmod(acc, c->mod[k] % qi, qi), v[k] % qi, qi);
            v[i] = mulmod(submod(m[(size_t)i * n + j], acc, qi), inv[i], qi);
        }
        /* x > (Q-1)/2  <=>  digits lexicographically above h_i = (q_i-1)/2 */
        int neg = 0;
        for (int32_t i = (int32_t)nl - 1; i >= 0; i--) {
            u64 h = (c->mod[i] - 1) / 2;
            if (v[i] > h) { neg = 1; break; }
            if (v[i] < h) { neg = 0; break; }
        }
        /* x = v_0 + q_0 (v_1 + q_1 (v_2 + ...)); negative: Q - x = (digit-wise complement) + 1 */
        long double acc = 0;
        for (int32_t i = (int32_t)nl - 1; i >= 0; i--) {
            long double d = neg ? (long double)(c->mod[i] - 1 - v[i]) : (long double)v[i];
            acc = acc * (long double)c->mod[i] + d;
        }
        if (neg) acc = -(acc + 1);
        double r = (double)(acc / (long double)scale);
        if (j < slots) re[j] = r; else im[j - slots] = r;
    }
    fft_special(c, re, im, slots);
    memcpy(out, re, slots * sizeof(double));
    free(m); free(re); free(im);
}
