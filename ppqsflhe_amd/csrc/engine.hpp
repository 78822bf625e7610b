// engine.hpp -- device-resident context + batched operations behind the C-ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "modarith.hpp"
#include "ntt_kernels.hpp"
#include "params.hpp"

namespace mk {

struct HipError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct NoDevice : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define MK_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            throw mk::HipError(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" __FILE__ \
                               ":" + std::to_string(__LINE__) + ")");                             \
    } while (0)

constexpr int MAX_CONV_IN = 8;    // alpha, K <= 8
constexpr int MAX_CONV_OUT = 40;  // complement limbs of one digit

// device image of one BaseConvTable, passed to k_conv_col by value (constant/SGPR space)
struct DevConv {
    uint32_t n_in, n_out;
    uint32_t src_id[MAX_CONV_IN];     // limb ids of the sources
    uint32_t src_slot[MAX_CONV_IN];   // slot of each source inside the input polynomial
    uint32_t dst_id[MAX_CONV_OUT];    // limb ids of the targets
    uint32_t dst_slot[MAX_CONV_OUT];  // slot of each target inside the output polynomial
    u64 hatinv[MAX_CONV_IN], hatinv_sh[MAX_CONV_IN];
    const u64 *hat;        // device, [n_in][n_out]: [S/s_i]_t
    const double *hat_d;   // device, [n_in][n_out]: the same as doubles (used for fp64-class targets)
    const double *hatq_d;  // device, [n_in][n_out]: [S/s_i]_t / t
};

// the stream a pass is launched on
struct Lanes {
    hipStream_t main = nullptr;
};

// Switches of the library (environment variables MKCKKS_*), read once when a context is created.  Defaults are the
// measured best; every non-default value has a parity test (tests/test_gpu_parity.py).
struct Knobs {
    uint32_t chunk = 16;         // MKCKKS_CHUNK: ciphertexts per workspace chunk
    uint32_t qsum_group = 8;     // MKCKKS_QSUM_GROUP: clients per pass of the merged n-client flow (one forward transform of
                                 // the summed ModDown conversions per group: 8 is +2.7 % against 4, 2 is -6.6 %)
    bool cu_affine = true;       // MKCKKS_CU_AFFINE=0: plain XCD-aware placement; default: workgroups that share operand tiles on the same CU (group_member)
    bool generic_ntt = false;    // MKCKKS_GENERIC_NTT=1: LDS-stage kernels for both passes
    bool no_pm = false;          // MKCKKS_NO_PM=1: Shoup butterflies on the integer limbs instead of the pseudo-Mersenne ones
    bool no_fp64 = false;        // MKCKKS_NO_FP64=1: integer arithmetic on every limb
    static Knobs from_env();
};

class Engine {
public:
    explicit Engine(const ParamSet &ps, int device);
    ~Engine();
    Engine(const Engine &) = delete;
    Engine &operator=(const Engine &) = delete;

    const ParamSet &params() const { return ps_; }
    bool has_device() const { return device_ >= 0; }
    int device() const { return device_; }
    void set_stream(hipStream_t s) { stream_ = s; }
    hipStream_t stream() const { return stream_; }
    void sync();

    void *dev_alloc(size_t bytes);
    void dev_free(void *p);
    void upload(void *d, const void *h, size_t bytes);
    void download(void *h, const void *d, size_t bytes);
    // I/O pipeline of the hosts: pinned host buffers and copies on the context's two copy streams (one per direction,
    // PCIe is full duplex), ordered against the compute stream by fences; every copy gets a ticket the host can poll
    void *host_alloc(size_t bytes);
    void host_free(void *p);
    uint64_t upload_async(void *d, const void *h, size_t bytes);
    uint64_t download_async(void *h, const void *d, size_t bytes);
    bool copy_done(uint64_t ticket);
    void copy_wait(uint64_t ticket);
    void fence_uploads();   // compute stream: wait for every upload enqueued so far
    void fence_compute();   // download stream: wait for all compute enqueued so far
    // residues of ciphertexts u64[n_ct][2][nl][N] that are not below their limb's modulus (0 = all canonical); synchronous
    uint64_t count_noncanonical(const u64 *ct, uint32_t n_ct, uint32_t nl);

    // transforms, in place on u64[n_polys][ext][N]; ext = nl (+K when with_p)
    void ntt_forward(u64 *d, uint32_t n_polys, uint32_t nl, bool with_p);
    void ntt_inverse(u64 *d, uint32_t n_polys, uint32_t nl, bool with_p);

    void eval_add(const u64 *a, const u64 *b, u64 *out, uint32_t n_ct, uint32_t nl);
    void eval_sum(const u64 *in, u64 *out, uint32_t n_clients, uint32_t n_ct, uint32_t nl);
    void rescale(const u64 *in, u64 *out, uint32_t n_ct, uint32_t nl, const std::vector<u64> *factors);
    void mult_const(u64 *ct, uint32_t n_ct, uint32_t nl, const std::vector<u64> &factors);
    void reduce_mod(u64 *ct, uint32_t n_ct, uint32_t nl, uint32_t n_terms);
    // RCCL exchange of the per-GPU partial sums: shard[b] = (sum over ranks of partial_r[rank * n_ct_shard + b]) mod q,
    // partial u64[n_ranks * n_ct_shard][2][nl][N] on every rank, shard u64[n_ct_shard][2][nl][N] (comm: ncclComm_t)
    void reduce_scatter_sum_mod(void *comm, const u64 *partial, u64 *shard, uint32_t n_ct_shard, uint32_t nl,
                                uint32_t n_ranks);

    // out[b] = sum over clients of ReEncrypt(cts[c][b], evks[c]); cts [C][n_ct][2][nl][N], evks [C][beta][2][D][N]
    void reencrypt_sum(const u64 *cts, const u64 *evks, u64 *out, uint32_t n_clients, uint32_t n_ct, uint32_t nl);
    void modup(const u64 *c1, u64 *digits, uint32_t n, uint32_t nl);
    void moddown(const u64 *in, u64 *out, uint32_t n, uint32_t nl);
    // accumulate: out[b] += ReEncrypt(ct[b]) (coefficient-wise, mod q) -- the fold into a running aggregate
    void reencrypt(const u64 *ct, const u64 *evk, u64 *out, uint32_t n_ct, uint32_t nl, bool accumulate = false);

    void keygen(const int8_t *s, const u64 *a, const int32_t *e, u64 *pk, u64 *sk);
    void rekeygen(const int8_t *s_old, const u64 *pk_new, const int8_t *u, const int32_t *e0,
                  const int32_t *e1, u64 *evk);
    void encrypt(const u64 *pk, const u64 *pt, const int8_t *v, const int32_t *e0, const int32_t *e1,
                 u64 *ct, uint32_t n_ct, uint32_t nl);
    void lift_ntt(const double *coef, u64 *out, uint32_t n, uint32_t nl);
    void decrypt(const u64 *ct, const u64 *sk, u64 *m, uint32_t n_ct, uint32_t nl);
    // counter-based samplers (ChaCha20 block function under a 256-bit key): element i of stream sid is a pure
    // function of (key, sid, i), independent of launch shape
    void sample_ternary(int8_t *out, size_t count, const uint8_t *key32, uint32_t sid);
    void sample_gauss(int32_t *out, size_t count, double sigma, const uint8_t *key32, uint32_t sid);
    void sample_uniform(u64 *out, uint32_t items, uint32_t nl, bool with_p, const uint8_t *key32, uint32_t sid);
    void chacha_block(uint32_t *d_out16, const uint8_t *key32, uint32_t counter, const uint32_t nonce[3]);  // KAT hook
    // CKKS canonical embedding on the device: vals [n][N/2] reals <-> plaintexts / decrypted polynomials
    void encode(const double *vals, u64 *pt, uint32_t n, uint32_t nl, double scale);
    void decode(const u64 *m, double *vals, uint32_t n, uint32_t nl, double scale);

    void host_twiddles(uint32_t limb, bool inverse, std::vector<u64> &out) const;
    // diagnostic builds (-DMK_STAMP=1, MKCKKS_STAMPS=1): copy one region of in-kernel phase stamps (2^20 words) to the
    // host and clear it; 0 words in a product build
    size_t debug_stamps(unsigned long long *h_out, uint32_t region);

private:
    void need_device() const;
    Lanes lanes() const;
    void check_nl(uint32_t nl) const;
    u64 *workspace(size_t words);  // grow-only scratch arena (stream-ordered reuse)
    const DevConv &modup_conv(uint32_t nl, uint32_t part);
    const DevConv &moddown_conv(uint32_t nl);
    const u64 *limb_vector(const std::string &key, const std::vector<u64> &vals);  // cached small device arrays
    void ntt_launch(u64 *d, uint32_t n_polys, uint32_t nl, uint32_t ext, bool inverse, const u64 *scale,
                    const u64 *scale_sh);
    void reencrypt_chunk(const u64 *ct, const u64 *evk, u64 *out, uint32_t n_ct, uint32_t nl, bool accumulate);
    const u64 *folded_scale(uint32_t nl);
    const u64 *p_inverse(uint32_t nl);
    void modup_core(const u64 *c1, size_t c1_stride, u64 *coef, u64 *dig, uint32_t cnt, uint32_t nl,
                    bool rows_int_only = false, uint32_t in_group = 0, size_t in_gstride = 0);
    bool qsum_ok(uint32_t nl) const;
    void reencrypt_sum_merged(const u64 *cts, const u64 *evks, u64 *out, uint32_t n_clients, uint32_t n_ct, uint32_t nl);
    const u64 *p_doubles();
    // returns true when the inverse ROW pass of the P limbs was done on the fly into `pc` (ModDown then starts with
    // the inverse column pass)
    bool keyswitch_digits(const u64 *c1, size_t ct_stride, const u64 *evk, u64 *coef, u64 *dig, u64 *til, u64 *pc,
                          uint32_t cnt, uint32_t nl);
    void moddown_core(const u64 *til, u64 *pc, u64 *conv, u64 *out, size_t out_stride, const u64 *add,
                      size_t add_stride, uint32_t cnt, uint32_t nl, bool accumulate, bool p_rows_done = false);
    void moddown_convert(const u64 *til, u64 *pc, u64 *conv, uint32_t cnt, uint32_t nl, bool rows_done);
    int conv_src_mode(const DevConv &cv) const;

    ParamSet ps_;
    int device_ = -1;
    hipStream_t stream_ = nullptr;
    NttTables tabs_{};
    LimbConst *d_limb_ = nullptr;
    u64 *d_tw_ = nullptr, *d_tw_sh_ = nullptr, *d_itw_ = nullptr, *d_itw_sh_ = nullptr;
    unsigned long long *d_stamps_ = nullptr;
    u64 *d_twb_ = nullptr, *d_itwb_ = nullptr;  // packed round-B tables of the row kernels (NttTables::twb)
    std::vector<uint8_t> fp_of_;  // per limb id: 1 = fp64 kernel instance
    // copies: tickets count up from 1; ticket t's completion event is ring slot t % COPY_RING
    static constexpr uint32_t COPY_RING = 64;
    uint64_t enqueue_copy(void *dst, const void *src, size_t bytes, bool up);
    void copy_streams();
    hipStream_t up_stream_ = nullptr, down_stream_ = nullptr;
    hipEvent_t copy_ev_[COPY_RING] = {};
    hipEvent_t ev_fence_ = nullptr;
    uint64_t next_ticket_ = 1, done_ticket_ = 0;  // every ticket <= done_ticket_ is known complete
    bool skip_rows_ = false;  // modup_core: leave the row pass of the converted digits to the fused kernels
    uint32_t *d_rot_ = nullptr;
    void *d_ksi_ = nullptr;
    u64 *ws_ = nullptr;
    size_t ws_words_ = 0;
    Knobs knobs_;
    std::map<std::pair<uint32_t, uint32_t>, DevConv> modup_cache_;
    std::map<uint32_t, DevConv> moddown_cache_;
    std::map<std::string, u64 *> vec_cache_;
    std::vector<void *> owned_;
};

}  // namespace mk
