/*
 * mkckks.h -- C-ABI of libmkckks_hip.so, the MI355X (gfx950) engine for the
 * multikey-CKKS PRE + aggregation hot path of CDACHPCIE25/PPQSFLHE.
 *
 * The reference has no FFI/plugin layer: its hot path is eight C++ main()s that
 * call OpenFHE's CryptoContext.  The drop-in boundary is
 * therefore (1) the CLI/JSON contract, re-created by the C++ hosts under
 * ppqsflhe_amd/host/, and (2) this library, whose entry points replace the
 * OpenFHE calls those mains make.  Each entry point cites the reference
 * call site(s) it stands in for (paths relative to /root/reference).
 *
 * Conventions
 *  - every function returns 0 on success, a negative MKCKKS_E_* code otherwise,
 *    never throws across the ABI; mkckks_last_error() gives the message of the
 *    calling thread's last failure.
 *  - "d_" pointers are DEVICE pointers (HBM) owned by the caller (hipMalloc,
 *    torch tensor data_ptr(), or mkckks_dev_alloc); "h_" pointers are host.
 *  - all work is enqueued on the context's HIP stream (mkckks_set_stream) and is
 *    asynchronous w.r.t. the host unless stated; a context is re-entrant per
 *    (context, stream) pair, not across threads sharing one context.
 *  - polynomials are limb-major uint64: a polynomial over the first nl Q-limbs
 *    is u64[nl][N]; a ciphertext is u64[2][nl][N] (c0 then c1); a batch is
 *    u64[n_ct][2][nl][N]; over QP the limb order is q_0..q_{L-1},p_0..p_{K-1};
 *    a public key is u64[2][D][N] (b then a); an eval (re-encryption) key is
 *    u64[beta][2][D][N] (digit j: b_j then a_j).  EVALUATION format = negacyclic
 *    NTT, natural-order input, bit-reversed output, psi = minimal primitive
 *    2N-th root (what OpenFHE serialises with "f":0).
 */
#ifndef MKCKKS_H
#define MKCKKS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MKCKKS_OK 0
#define MKCKKS_E_INVALID (-1)   /* bad argument / unsupported parameter set   */
#define MKCKKS_E_NODEVICE (-2)  /* no HIP device, or host-only context used   */
#define MKCKKS_E_HIP (-3)       /* HIP runtime error (message has the detail) */
#define MKCKKS_E_NOMEM (-4)
#define MKCKKS_E_INTERNAL (-5)

typedef struct mkckks_ctx mkckks_ctx;

/* server/config/config_cc.json:2-5 knobs + the ones genCC.cpp leaves at
 * OpenFHE defaults (first mod 60, aux 60, extra 20: CC.json 'ab','eb'). */
typedef struct mkckks_params {
    uint32_t log_n;        /* ring dimension N = 2^log_n (CC.json "rd")            */
    uint32_t mult_depth;   /* MultiplicativeDepth; #Q limbs L = mult_depth + 2     */
    uint32_t scaling_bits; /* ScalingModSize                                       */
    uint32_t first_bits;   /* FirstModSize (60)                                    */
    uint32_t dnum;         /* NumLargeDigits of HYBRID key switching ("dnum")      */
    uint32_t aux_bits;     /* auxiliary prime size ("ab", 60)                      */
    uint32_t extra_bits;   /* FLEXIBLEAUTOEXT extra limb size ("eb", 20)           */
    int32_t device;        /* HIP device ordinal; -1 = host-only (tables, no GPU)  */
} mkckks_params;

typedef struct mkckks_info {
    uint32_t ring_dim, num_q, num_p, alpha, beta, slots;
} mkckks_info;

const char *mkckks_last_error(void);
const char *mkckks_version(void);

/* ---- context -------------------------------------------------------------
 * replaces GenCryptoContext(params)+Enable(...) (server/src/genCC.cpp:68-76)
 * and Serial::DeserializeFromFile(cc) (server/src/changeCipherDomain.cpp:33,
 * aggregateEncryptedWeights.cpp:47 and the client mains): builds Q, P, roots,
 * twiddles and all CRT tables and keeps them resident in HBM. */
int mkckks_ctx_create(const mkckks_params *p, mkckks_ctx **out);
int mkckks_ctx_destroy(mkckks_ctx *c);
int mkckks_ctx_info(const mkckks_ctx *c, mkckks_info *out);
int mkckks_ctx_moduli(const mkckks_ctx *c, uint64_t *h_out /*D*/);
int mkckks_ctx_roots(const mkckks_ctx *c, uint64_t *h_out /*D*/);
/* arithmetic the device kernels use per limb (diagnostics, bench.py's instruction-issue ceiling): 0 = 64-bit integer
 * with Shoup/Harvey butterflies, 1 = fp64 (moduli below 1.25 * 2^50), 2 = 64-bit integer with pseudo-Mersenne
 * butterflies (q = 2^k - c; OpenFHE's 60-bit primes).  Results are the same bits in every class. */
#define MKCKKS_ARITH_INT 0
#define MKCKKS_ARITH_FP64 1
#define MKCKKS_ARITH_PM 2
int mkckks_ctx_arith(const mkckks_ctx *c, uint8_t *h_out /*D*/);
/* CryptoParametersCKKSRNS::GetScalingFactorReal / RealBig (level = #dropped limbs) */
int mkckks_scaling_factor(const mkckks_ctx *c, uint32_t level, int big, double *out);
/* stream: a hipStream_t passed as void* (NULL = default stream) */
int mkckks_set_stream(mkckks_ctx *c, void *hip_stream);
int mkckks_sync(mkckks_ctx *c);

/* ---- device memory helpers (so hosts need no HIP headers) ---------------- */
int mkckks_dev_alloc(mkckks_ctx *c, size_t bytes, void **d_out);
int mkckks_dev_free(mkckks_ctx *c, void *d_ptr);
int mkckks_upload(mkckks_ctx *c, void *d_dst, const void *h_src, size_t bytes);   /* synchronous */
int mkckks_download(mkckks_ctx *c, void *h_dst, const void *d_src, size_t bytes); /* synchronous */

/* ---- I/O pipeline of the server hosts (new; SURVEY.md 8f row f1: "removes the I/O wall") -----------------------
 * The reference's server moves every ciphertext through the host, one at a time and synchronously, between
 * Base64Decode + Serial::Deserialize and cc->ReEncrypt / cc->EvalAdd (server/src/changeCipherDomain.cpp:61-117,
 * aggregateEncryptedWeights.cpp:18-30,54-119).  Here a host reads ciphertext payloads straight into PINNED buffers
 * (mkckks_host_alloc) and enqueues them on the context's upload stream while it reads the next ones; results leave on
 * a download stream (PCIe is full duplex).  Every copy gets a ticket (counting up from 1): mkckks_copy_done polls it,
 * mkckks_copy_wait blocks on it (and on every earlier copy of the same direction).  Ordering against the compute
 * stream is explicit: mkckks_fence_uploads makes the compute stream wait for every upload enqueued so far,
 * mkckks_fence_compute makes the download stream wait for all compute enqueued so far.  Host buffers of an
 * asynchronous copy must be pinned and stay untouched until the copy's ticket is done.  One host thread drives a
 * context (reader threads only fill pinned buffers).
 * mkckks_count_noncanonical: nothing in a client's file is trusted and the kernels assume canonical residues; counts
 * the residues of d_ct u64[n_ct][2][nl][N] that are not below their limb's modulus (synchronous; 0 = accept). */
int mkckks_host_alloc(mkckks_ctx *c, size_t bytes, void **h_out);
int mkckks_host_free(mkckks_ctx *c, void *h_ptr);
int mkckks_upload_async(mkckks_ctx *c, void *d_dst, const void *h_src_pinned, size_t bytes, uint64_t *ticket_out);
int mkckks_download_async(mkckks_ctx *c, void *h_dst_pinned, const void *d_src, size_t bytes, uint64_t *ticket_out);
int mkckks_copy_done(mkckks_ctx *c, uint64_t ticket, int *done_out);
int mkckks_copy_wait(mkckks_ctx *c, uint64_t ticket);
int mkckks_fence_uploads(mkckks_ctx *c);
int mkckks_fence_compute(mkckks_ctx *c);
int mkckks_count_noncanonical(mkckks_ctx *c, const uint64_t *d_ct, uint32_t n_ct, uint32_t nl, uint64_t *h_count);

/* ---- transforms: DCRTPoly::SetFormat (OpenFHE ChineseRemainderTransformFTT)
 * d_polys is u64[n_polys][nl(+K)][N], transformed in place.  with_p != 0 means
 * each polynomial carries the K P-limbs after its nl Q-limbs. */
int mkckks_ntt_forward_batch(mkckks_ctx *c, uint64_t *d_polys, uint32_t n_polys, uint32_t nl, int with_p);
int mkckks_ntt_inverse_batch(mkckks_ctx *c, uint64_t *d_polys, uint32_t n_polys, uint32_t nl, int with_p);

/* ---- cc->EvalAdd(ct1, ct2)  (aggregateEncryptedWeights.cpp:82,91,106) ---- */
int mkckks_eval_add_batch(mkckks_ctx *c, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out,
                          uint32_t n_ct, uint32_t nl);
/* n-client generalisation of the same loop: d_out[b] = sum_k d_in[k][b], d_in is
 * u64[n_clients][n_ct][2][nl][N]. */
int mkckks_eval_sum_batch(mkckks_ctx *c, const uint64_t *d_in, uint64_t *d_out, uint32_t n_clients,
                          uint32_t n_ct, uint32_t nl);

/* ---- cc->EvalMult(ct, operand) on a noiseScaleDeg-2 ciphertext
 * (aggregateEncryptedWeights.cpp:83,92,107: EvalMult(ct_sum, 0.5)): rescale
 * (drop limb nl-1) then multiply by round(operand * sf(level+1)).
 * in u64[n_ct][2][nl][N] at `level` -> out u64[n_ct][2][nl-1][N]. */
int mkckks_rescale_mult_const_batch(mkckks_ctx *c, const uint64_t *d_in, uint64_t *d_out, uint32_t n_ct,
                                    uint32_t nl, double operand);
/* the two halves on their own (ModReduceInternalInPlace / EvalMultCoreInPlace) */
int mkckks_rescale_batch(mkckks_ctx *c, const uint64_t *d_in, uint64_t *d_out, uint32_t n_ct, uint32_t nl);
int mkckks_mult_const_batch(mkckks_ctx *c, uint64_t *d_ct, uint32_t n_ct, uint32_t nl, double operand);

/* ---- cc->ReEncrypt(ct, reKey)  (changeCipherDomain.cpp:74,89,105) --------
 * INDCPA proxy re-encryption = hybrid key switch of c1 with the eval key:
 * out = (c0 + <d,b>/P, <d,a>/P).  d_evk is u64[beta][2][D][N] (full level);
 * ciphertexts have nl <= L limbs; in/out may alias. */
int mkckks_reencrypt_batch(mkckks_ctx *c, const uint64_t *d_ct, const uint64_t *d_evk, uint64_t *d_out,
                           uint32_t n_ct, uint32_t nl);
/* the same, folded into a running aggregate: d_acc[b] = d_acc[b] + ReEncrypt(d_ct[b]) coefficient-wise mod q_i
 * (ReEncrypt at changeCipherDomain.cpp:74 followed by EvalAdd at aggregateEncryptedWeights.cpp:82, without the
 * round trip of the re-encrypted ciphertext through HBM).  d_acc must not alias d_ct. */
int mkckks_reencrypt_accumulate_batch(mkckks_ctx *c, const uint64_t *d_ct, const uint64_t *d_evk, uint64_t *d_acc,
                                      uint32_t n_ct, uint32_t nl);
/* n-client form of the server loop (changeCipherDomain x n, then the EvalAdd chain of
 * aggregateEncryptedWeights.cpp:82): d_out[b] = sum_c ReEncrypt(d_cts[c][b], d_evks[c]) coefficient-wise mod q_i.
 * d_cts u64[n_clients][n_ct][2][nl][N], d_evks u64[n_clients][beta][2][D][N] (one re-encryption key per client,
 * all towards the common domain), d_out u64[n_ct][2][nl][N].  Bit-identical to re-encrypting every ciphertext and
 * adding them in any order; the last ModDown pass of all clients and the sum are one kernel. */
int mkckks_reencrypt_sum_batch(mkckks_ctx *c, const uint64_t *d_cts, const uint64_t *d_evks, uint64_t *d_out,
                               uint32_t n_clients, uint32_t n_ct, uint32_t nl);
/* stages of the above, exposed for parity tests and profiling:
 * KeySwitchHYBRID::EvalKeySwitchPrecomputeCore: c1 u64[n][nl][N] ->
 * digits u64[n][nparts][nl+K][N]; ApproxModDown: u64[n][nl+K][N] -> u64[n][nl][N]. */
int mkckks_modup_batch(mkckks_ctx *c, const uint64_t *d_c1, uint64_t *d_digits, uint32_t n, uint32_t nl);
int mkckks_moddown_batch(mkckks_ctx *c, const uint64_t *d_in, uint64_t *d_out, uint32_t n, uint32_t nl);

/* ---- randomness for KeyGen / ReKeyGen / Encrypt, generated in HBM ------------
 * OpenFHE's TernaryUniformGenerator, DiscreteGaussianGenerator (sigma 3.19) and
 * DiscreteUniformGenerator.  Generator: the ChaCha20 block function (RFC 8439) under a
 * 256-bit key (h_key32: 32 HOST bytes, drawn from the OS by the callers -- getrandom(2) in
 * ppqsflhe_amd/host/sampler.hpp): a cryptographic PRF, like OpenFHE's Blake2-based PRNG and
 * unlike a 64-bit-seeded statistical generator.  Counter based: element i of `stream_id`
 * is a pure function of (key, stream_id, i).  Distributional parity only (OpenFHE's PRNG
 * stream cannot be reproduced).  d_out: int8[count] / int32[count] /
 * u64[n_polys][nl(+K)][N] (uniform in [0, q_limb), exact by rejection). */
#define MKCKKS_SAMPLER_KEY_BYTES 32
int mkckks_sample_ternary(mkckks_ctx *c, int8_t *d_out, size_t count, const uint8_t *h_key32, uint32_t stream_id);
int mkckks_sample_gauss(mkckks_ctx *c, int32_t *d_out, size_t count, double sigma, const uint8_t *h_key32,
                        uint32_t stream_id);
int mkckks_sample_uniform(mkckks_ctx *c, uint64_t *d_out, uint32_t n_polys, uint32_t nl, int with_p,
                          const uint8_t *h_key32, uint32_t stream_id);
/* known-answer hook: the 16 output words of one ChaCha20 block (RFC 8439 2.3.2) -> d_out16 (device) */
int mkckks_chacha20_block(mkckks_ctx *c, uint32_t *d_out16, const uint8_t *h_key32, uint32_t counter,
                          const uint32_t *h_nonce3);

/* ---- cc->KeyGen()  (client/src/keyGen.cpp:33) -----------------------------
 * randomness is supplied by the caller (mkckks_sample_* under OS-drawn keys in
 * ppqsflhe_amd/host, or a test's seeded vectors): s ternary int8[N], e int32[N] (COEFFICIENT),
 * a u64[D][N] uniform residues (taken as EVALUATION).
 * d_pk out u64[2][D][N], d_sk out u64[D][N] (EVALUATION). */
int mkckks_keygen(mkckks_ctx *c, const int8_t *d_s, const uint64_t *d_a, const int32_t *d_e,
                  uint64_t *d_pk, uint64_t *d_sk);
/* ---- cc->ReKeyGen(mySk, peerPk)  (client/src/REkeyGen.cpp:52) -------------
 * s_old int8[N]; u int8[beta][N]; e0,e1 int32[beta][N]; d_evk out u64[beta][2][D][N]. */
int mkckks_rekeygen(mkckks_ctx *c, const int8_t *d_s_old, const uint64_t *d_pk_new, const int8_t *d_u,
                    const int32_t *d_e0, const int32_t *d_e1, uint64_t *d_evk);

/* ---- cc->Encrypt(pk, pt)  (client/src/encryptModelWeights.cpp:83,91,110) --
 * d_pt u64[n_ct][nl][N] encoded plaintexts (EVALUATION); v int8[n_ct][N];
 * e0,e1 int32[n_ct][N]; out u64[n_ct][2][nl][N]. */
int mkckks_encrypt_batch(mkckks_ctx *c, const uint64_t *d_pk, const uint64_t *d_pt, const int8_t *d_v,
                         const int32_t *d_e0, const int32_t *d_e1, uint64_t *d_ct, uint32_t n_ct, uint32_t nl);
/* scaled real coefficient vectors -> residues in EVALUATION format over nl limbs
 * (the integer half of CKKSPackedEncoding::Encode; the fp64 canonical embedding
 * runs on the GPU too: mkckks_encode_batch).  Each double (|x| < 2^120) is
 * rounded to the nearest integer (ties away from zero) and that integer is
 * reduced exactly per limb: coef double[n][N] -> u64[n][nl][N]. */
int mkckks_lift_ntt_batch(mkckks_ctx *c, const double *d_coef, uint64_t *d_out, uint32_t n, uint32_t nl);

/* ---- cc->MakeCKKSPackedPlaintext(values)  (encryptModelWeights.cpp:82,90,109) --
 * CKKSPackedEncoding::Encode, full packing: d_vals double[n][N/2] real slot values
 * (zero padded by the caller) -> inverse canonical embedding (fp64 special FFT), x scale,
 * round, residues, NTT -> d_pt u64[n][nl][N].  scale: the plaintext's scaling factor
 * (FLEXIBLEAUTOEXT level 0: mkckks_scaling_factor(ctx, 0, big=1)). */
int mkckks_encode_batch(mkckks_ctx *c, const double *d_vals, uint64_t *d_pt, uint32_t n, uint32_t nl, double scale);
/* ---- pt->GetRealPackedValue()  (decryptModelWeights.cpp:83,92,109) ------------
 * CRT interpolation of d_m u64[n][nl][N] (output of mkckks_decrypt_batch), / scale,
 * canonical embedding -> d_vals double[n][N/2].  Upstream Decode's noise flooding is
 * not applied. */
int mkckks_decode_batch(mkckks_ctx *c, const uint64_t *d_m, double *d_vals, uint32_t n, uint32_t nl, double scale);

/* ---- cc->Decrypt(sk, ct, &pt)  (client/src/decryptModelWeights.cpp:81,90,108)
 * DecryptCore: m = INTT(c0 + c1*s); d_m out u64[n_ct][nl][N] (COEFFICIENT).
 * CRT interpolation + Decode follow on the GPU: mkckks_decode_batch. */
int mkckks_decrypt_batch(mkckks_ctx *c, const uint64_t *d_ct, const uint64_t *d_sk, uint64_t *d_m,
                         uint32_t n_ct, uint32_t nl);

/* ---- multi-GPU aggregation step (new) -----------------------
 * after an RCCL ncclSum over uint64 of `n_terms` canonical residues per word,
 * reduce every word mod its limb modulus: d_ct u64[n_ct][2][nl][N] in place. */
int mkckks_reduce_mod_batch(mkckks_ctx *c, uint64_t *d_ct, uint32_t n_ct, uint32_t nl, uint32_t n_terms);

/* ---- RCCL exchange step behind the C-ABI (SURVEY.md 8b export list, 8e.2) ----
 * Replaces the serial per-client loop of the reference's server (orchestration/server_fns.sh:62-80,
 * orchestration/run.sh:37-43: one changeCipherDomain per client, then one aggregateEncryptedWeights):
 * every GPU re-encrypts and sums ITS clients (mkckks_reencrypt_sum_batch), then
 *   d_shard[b] = ( sum over ranks r of d_partial_r[rank * n_ct_shard + b] )  coefficient-wise mod q_i
 * as ONE ncclReduceScatter(ncclUint64, ncclSum) over xGMI + a word-wise reduction (n_ranks <= 8 canonical residues
 * below 2^61 cannot wrap 2^64).  d_partial: u64[n_ranks * n_ct_shard][2][nl][N] on every rank; d_shard:
 * u64[n_ct_shard][2][nl][N].  Enqueued on the context's stream.  `comm` is an ncclComm_t (as void*): from
 * mkckks_comm_create, or any communicator of the RCCL this process carries whose rank is bound to the context's
 * device.  librccl.so.1 is resolved at first use (dlopen by SONAME: a process that already loaded an RCCL, e.g.
 * PyTorch's, keeps that one); mkckks_comm_library() names the file.
 * Communicator bootstrap: rank 0 calls mkckks_comm_unique_id, ships the MKCKKS_COMM_ID_BYTES bytes to the other
 * ranks by any side channel, every rank calls mkckks_comm_create (collective: ncclCommInitRank). */
#define MKCKKS_COMM_ID_BYTES 128
int mkckks_comm_unique_id(void *h_id_out /* MKCKKS_COMM_ID_BYTES */);
int mkckks_comm_create(mkckks_ctx *c, const void *h_id, int n_ranks, int rank, void **comm_out);
int mkckks_comm_destroy(mkckks_ctx *c, void *comm);
int mkckks_reduce_scatter_sum_mod(mkckks_ctx *c, void *comm, const uint64_t *d_partial, uint64_t *d_shard,
                                  uint32_t n_ct_shard, uint32_t nl, uint32_t n_ranks);
const char *mkckks_comm_library(void);

/* ---- diagnostics: in-kernel phase stamps of a -DMK_STAMP=1 build created under MKCKKS_STAMPS=1 (tools/stamps.py);
 * h_out: 2^20 words; *n_out = words written, 0 in a product build */
int mkckks_debug_stamps(mkckks_ctx *c, unsigned long long *h_out, uint32_t region, size_t *n_out);

/* ---- introspection for tests: copy a CRT table to the host ---------------- */
int mkckks_ctx_twiddles(const mkckks_ctx *c, uint32_t limb, int inverse, uint64_t *h_out /*N*/);

#ifdef __cplusplus
}
#endif
#endif /* MKCKKS_H */
