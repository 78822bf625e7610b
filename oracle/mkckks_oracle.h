/*
 * mkckks_oracle.h -- CPU restatement of the PRE + aggregation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and only as the checker / CPU baseline.
 *
 * What it restates: the arithmetic that the reference reaches through
 * OpenFHE (module openfheorg/openfhe-development, version UNPINNED in the
 * reference: lib/openfhe-development/ is an empty directory, evidence points
 * to 1.3.x-1.4.x) at these call sites:
 *   server/src/changeCipherDomain.cpp:74,89,105   cc->ReEncrypt(ct, reKey)
 *   server/src/aggregateEncryptedWeights.cpp:82-83 EvalAdd, EvalMult(ct, 0.5)
 *   client/src/encryptModelWeights.cpp:82-83      MakeCKKSPackedPlaintext+Encrypt
 *   client/src/decryptModelWeights.cpp:81-83      Decrypt + GetRealPackedValue
 *   client/src/keyGen.cpp:33, client/src/REkeyGen.cpp:52, server/src/genCC.cpp:32-79
 * OpenFHE's source is absent from /root/reference, so the published algorithms
 * are restated (file names cited per function, no line numbers available).
 *
 * Parity pinning: the restatement is pinned by the reference's own DATA
 * fixtures (tests/golden): CC.json moduli/roots (parameter KAT), the two
 * clients' secret keys in EVALUATION form (NTT KAT) and the plaintext-in /
 * decrypted-out weight files (end-to-end KAT).  ReEncrypt / rescale / const-mul
 * outputs have no reference vector available (all ciphertext and key blobs are
 * in .MISSING_LARGE_BLOBS): for those rows parity vs OpenFHE is UNPINNED and
 * rests on the algebra + the end-to-end KAT.
 *
 * Layout convention everywhere: limb-major uint64, a polynomial over the first
 * `nl` Q-limbs is u64[nl][N]; a ciphertext is u64[2][nl][N] (c0 then c1); over
 * QP the limb order is q_0..q_{L-1}, p_0..p_{K-1}; an eval key is
 * u64[beta][2][D][N] with component 0 = b_j, component 1 = a_j.
 * EVALUATION format = negacyclic NTT, natural-order input, bit-reversed output,
 * psi = minimal primitive 2N-th root.
 */
#ifndef MKCKKS_ORACLE_H
#define MKCKKS_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

/* genCC.cpp:32-79 -> GenCryptoContext(CCParams<CryptoContextCKKSRNS>) with
 * FLEXIBLEAUTOEXT scaling, HYBRID key switching, uniform ternary secrets.
 * num Q limbs = mult_depth + 2 (one extra_bits-sized limb last). */
orc_ctx *orc_ctx_new(uint32_t log_n, uint32_t mult_depth, uint32_t scaling_bits,
                     uint32_t first_bits, uint32_t dnum, uint32_t aux_bits,
                     uint32_t extra_bits);
void orc_ctx_free(orc_ctx *c);

void orc_set_threads(int n);               /* OpenMP team size for the per-limb loops */
uint32_t orc_ring_dim(const orc_ctx *c);
uint32_t orc_num_q(const orc_ctx *c);     /* L */
uint32_t orc_num_p(const orc_ctx *c);     /* K */
uint32_t orc_alpha(const orc_ctx *c);     /* limbs per digit */
uint32_t orc_beta(const orc_ctx *c);      /* digits at full level */
void orc_moduli(const orc_ctx *c, uint64_t *out);   /* D = L+K values */
void orc_roots(const orc_ctx *c, uint64_t *out);    /* D minimal 2N-th roots */
double orc_sf(const orc_ctx *c, uint32_t level);     /* m_scalingFactorsReal */
double orc_sf_big(const orc_ctx *c, uint32_t level); /* m_scalingFactorsRealBig */

/* number-theory helpers exposed for the parameter KATs */
int orc_is_prime(uint64_t n);
uint64_t orc_min_root_of_unity(uint64_t m, uint64_t q);

/* transformnat-impl.h: ForwardTransformToBitReverse / InverseTransformFromBitReverse.
 * limb = index into QP (0..D-1); in place on N words. */
void orc_ntt_fwd(const orc_ctx *c, uint32_t limb, uint64_t *a);
void orc_ntt_inv(const orc_ctx *c, uint32_t limb, uint64_t *a);

/* EvalAddCore: out = a + b coefficient-wise over nl limbs x 2 polys. */
void orc_eval_add(const orc_ctx *c, uint32_t nl, const uint64_t *a,
                  const uint64_t *b, uint64_t *out);

/* DCRTPolyImpl::DropLastElementAndScale on both polys: in u64[2][nl][N]
 * (EVALUATION) -> out u64[2][nl-1][N]. */
void orc_rescale(const orc_ctx *c, uint32_t nl, const uint64_t *in, uint64_t *out);

/* LeveledSHECKKSRNS::GetElementForEvalMult: per-limb integer for `operand`
 * scaled by sf(level); factors[nl]. */
void orc_const_factors(const orc_ctx *c, uint32_t nl, uint32_t level,
                       double operand, uint64_t *factors);
/* EvalMultCoreInPlace: ct[k][i][:] *= factors[i]. */
void orc_mult_factors(const orc_ctx *c, uint32_t nl, const uint64_t *factors,
                      uint64_t *ct);

/* KeySwitchHYBRID::KeySwitchInPlace as reached from PRE::ReEncrypt(pk=null):
 * out = (c0 + <d,b>/P, <d,a>/P), d = ModUp digits of c1. ct has nl limbs. */
void orc_reencrypt(const orc_ctx *c, uint32_t nl, const uint64_t *ct,
                   const uint64_t *evk, uint64_t *out);
/* pieces of the above exposed for stage-by-stage GPU parity tests:
 * digits out u64[nparts][nl+K][N] (EVALUATION), returns nparts. */
uint32_t orc_modup_digits(const orc_ctx *c, uint32_t nl, const uint64_t *c1,
                          uint64_t *digits);
/* ApproxModDown: in u64[nl+K][N] (EVALUATION) -> out u64[nl][N]. */
void orc_moddown(const orc_ctx *c, uint32_t nl, const uint64_t *in, uint64_t *out);

/* PKERNS::KeyGenInternal over QP: pk = (e - a*s, a). s_tern/e are signed
 * coefficient vectors (COEFFICIENT format), a_eval is u64[D][N] uniform
 * residues taken as already in EVALUATION format. sk_eval out u64[D][N]. */
void orc_keygen(const orc_ctx *c, const int8_t *s_tern, const uint64_t *a_eval,
                const int32_t *e, uint64_t *pk, uint64_t *sk_eval);
/* KeySwitchHYBRID::KeySwitchGenInternal(oldSk, newPk) (REkeyGen.cpp:52):
 * u[beta][N] ternary, e0/e1[beta][N] gaussian. evk out u64[beta][2][D][N]. */
void orc_rekeygen(const orc_ctx *c, const int8_t *s_old, const uint64_t *pk_new,
                  const int8_t *u, const int32_t *e0, const int32_t *e1,
                  uint64_t *evk);

/* CKKSPackedEncoding::Encode: nvals reals into N/2 slots (zero padded), scale
 * `scale`, over nl limbs, result in EVALUATION format u64[nl][N]. */
void orc_encode(const orc_ctx *c, const double *vals, uint32_t nvals,
                double scale, uint32_t nl, uint64_t *pt);
/* PKERNS::Encrypt: ct = (pk0*v + e0 + pt, pk1*v + e1) on the first nl Q-limbs. */
void orc_encrypt(const orc_ctx *c, uint32_t nl, const uint64_t *pk,
                 const uint64_t *pt, const int8_t *v, const int32_t *e0,
                 const int32_t *e1, uint64_t *ct);
/* PKERNS::DecryptCore + CRT interpolation + CKKSPackedEncoding::Decode without
 * the decode-time noise flooding: out N/2 reals = slots / scale. */
void orc_decrypt_decode(const orc_ctx *c, uint32_t nl, const uint64_t *ct,
                        const uint64_t *sk_eval, double scale, double *out);
/* DecryptCore only: m = INTT(c0 + c1*s), u64[nl][N] COEFFICIENT format. */
void orc_decrypt_core(const orc_ctx *c, uint32_t nl, const uint64_t *ct,
                      const uint64_t *sk_eval, uint64_t *m);

#ifdef __cplusplus
}
#endif
#endif
