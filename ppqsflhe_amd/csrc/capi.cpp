// capi.cpp -- extern "C" surface of libmkckks_hip.so (declared in include/mkckks.h).
// Thin: argument checks, exception -> status code translation, nothing else.
#include <cstring>
#include <new>
#include <string>

#include "../../include/mkckks.h"
#include "comm.hpp"
#include "engine.hpp"

struct mkckks_ctx {
    mk::Engine *eng;
};

namespace {
thread_local std::string g_err;

template <typename F>
int guarded(F &&f) {
    try {
        f();
        return MKCKKS_OK;
    } catch (const mk::NoDevice &e) {
        g_err = e.what();
        return MKCKKS_E_NODEVICE;
    } catch (const mk::HipError &e) {
        g_err = e.what();
        return MKCKKS_E_HIP;
    } catch (const std::invalid_argument &e) {
        g_err = e.what();
        return MKCKKS_E_INVALID;
    } catch (const std::out_of_range &e) {
        g_err = e.what();
        return MKCKKS_E_INVALID;
    } catch (const std::bad_alloc &) {
        g_err = "out of host memory";
        return MKCKKS_E_NOMEM;
    } catch (const std::exception &e) {
        g_err = e.what();
        return MKCKKS_E_INTERNAL;
    } catch (...) {
        g_err = "unknown error";
        return MKCKKS_E_INTERNAL;
    }
}

void need(bool ok, const char *what) {
    if (!ok) throw std::invalid_argument(what);
}
}  // namespace

extern "C" {

const char *mkckks_last_error(void) { return g_err.c_str(); }
const char *mkckks_version(void) { return "mkckks-hip 0.2 (gfx950)"; }

int mkckks_ctx_create(const mkckks_params *p, mkckks_ctx **out) {
    return guarded([&] {
        need(p && out, "null argument");
        mk::ParamSet ps;
        ps.generate(p->log_n, p->mult_depth, p->scaling_bits, p->first_bits, p->dnum, p->aux_bits, p->extra_bits);
        auto *eng = new mk::Engine(ps, p->device);
        *out = new mkckks_ctx{eng};
    });
}

int mkckks_ctx_destroy(mkckks_ctx *c) {
    return guarded([&] {
        if (!c) return;
        delete c->eng;
        delete c;
    });
}

int mkckks_ctx_info(const mkckks_ctx *c, mkckks_info *out) {
    return guarded([&] {
        need(c && out, "null argument");
        const mk::ParamSet &ps = c->eng->params();
        *out = mkckks_info{ps.n, ps.L, ps.K, ps.alpha, ps.beta, ps.n / 2};
    });
}

int mkckks_ctx_moduli(const mkckks_ctx *c, uint64_t *h_out) {
    return guarded([&] {
        need(c && h_out, "null argument");
        const auto &m = c->eng->params().moduli;
        std::memcpy(h_out, m.data(), m.size() * sizeof(uint64_t));
    });
}

int mkckks_ctx_arith(const mkckks_ctx *c, uint8_t *h_out) {
    return guarded([&] {
        need(c && h_out, "null argument");
        const auto &limb = c->eng->params().limb;
        for (size_t i = 0; i < limb.size(); ++i)
            h_out[i] = limb[i].fp ? MKCKKS_ARITH_FP64 : (limb[i].pm ? MKCKKS_ARITH_PM : MKCKKS_ARITH_INT);
    });
}

int mkckks_ctx_roots(const mkckks_ctx *c, uint64_t *h_out) {
    return guarded([&] {
        need(c && h_out, "null argument");
        const auto &r = c->eng->params().roots;
        std::memcpy(h_out, r.data(), r.size() * sizeof(uint64_t));
    });
}

int mkckks_scaling_factor(const mkckks_ctx *c, uint32_t level, int big, double *out) {
    return guarded([&] {
        need(c && out, "null argument");
        const mk::ParamSet &ps = c->eng->params();
        *out = big ? ps.sf_big.at(level) : ps.sf.at(level);
    });
}

int mkckks_ctx_twiddles(const mkckks_ctx *c, uint32_t limb, int inverse, uint64_t *h_out) {
    return guarded([&] {
        need(c && h_out, "null argument");
        need(limb < c->eng->params().D, "limb out of range");
        std::vector<uint64_t> w;
        c->eng->host_twiddles(limb, inverse != 0, w);
        std::memcpy(h_out, w.data(), w.size() * sizeof(uint64_t));
    });
}

int mkckks_set_stream(mkckks_ctx *c, void *s) {
    return guarded([&] {
        need(c, "null context");
        c->eng->set_stream(reinterpret_cast<hipStream_t>(s));
    });
}
int mkckks_sync(mkckks_ctx *c) {
    return guarded([&] {
        need(c, "null context");
        c->eng->sync();
    });
}

int mkckks_dev_alloc(mkckks_ctx *c, size_t bytes, void **d_out) {
    return guarded([&] {
        need(c && d_out, "null argument");
        *d_out = c->eng->dev_alloc(bytes);
    });
}
int mkckks_dev_free(mkckks_ctx *c, void *d) {
    return guarded([&] {
        need(c, "null context");
        if (d) c->eng->dev_free(d);
    });
}
int mkckks_upload(mkckks_ctx *c, void *d, const void *h, size_t bytes) {
    return guarded([&] {
        need(c && d && h, "null argument");
        c->eng->upload(d, h, bytes);
    });
}
int mkckks_download(mkckks_ctx *c, void *h, const void *d, size_t bytes) {
    return guarded([&] {
        need(c && d && h, "null argument");
        c->eng->download(h, d, bytes);
    });
}

int mkckks_host_alloc(mkckks_ctx *c, size_t bytes, void **h_out) {
    return guarded([&] {
        need(c && h_out, "null argument");
        *h_out = c->eng->host_alloc(bytes);
    });
}
int mkckks_host_free(mkckks_ctx *c, void *h) {
    return guarded([&] {
        need(c, "null context");
        c->eng->host_free(h);
    });
}
int mkckks_upload_async(mkckks_ctx *c, void *d, const void *h, size_t bytes, uint64_t *ticket_out) {
    return guarded([&] {
        need(c && (bytes == 0 || (d && h)), "null argument");
        const uint64_t t = c->eng->upload_async(d, h, bytes);
        if (ticket_out) *ticket_out = t;
    });
}
int mkckks_download_async(mkckks_ctx *c, void *h, const void *d, size_t bytes, uint64_t *ticket_out) {
    return guarded([&] {
        need(c && (bytes == 0 || (d && h)), "null argument");
        const uint64_t t = c->eng->download_async(h, d, bytes);
        if (ticket_out) *ticket_out = t;
    });
}
int mkckks_copy_done(mkckks_ctx *c, uint64_t ticket, int *done_out) {
    return guarded([&] {
        need(c && done_out, "null argument");
        *done_out = c->eng->copy_done(ticket) ? 1 : 0;
    });
}
int mkckks_copy_wait(mkckks_ctx *c, uint64_t ticket) {
    return guarded([&] {
        need(c, "null context");
        c->eng->copy_wait(ticket);
    });
}
int mkckks_fence_uploads(mkckks_ctx *c) {
    return guarded([&] {
        need(c, "null context");
        c->eng->fence_uploads();
    });
}
int mkckks_fence_compute(mkckks_ctx *c) {
    return guarded([&] {
        need(c, "null context");
        c->eng->fence_compute();
    });
}
int mkckks_debug_stamps(mkckks_ctx *c, unsigned long long *h_out, uint32_t region, size_t *n_out) {
    return guarded([&] {
        need(c && h_out && n_out, "null argument");
        *n_out = c->eng->debug_stamps(h_out, region);
    });
}
int mkckks_count_noncanonical(mkckks_ctx *c, const uint64_t *d_ct, uint32_t n_ct, uint32_t nl, uint64_t *h_count) {
    return guarded([&] {
        need(c && h_count && (n_ct == 0 || d_ct), "null argument");
        *h_count = c->eng->count_noncanonical(d_ct, n_ct, nl);
    });
}

int mkckks_ntt_forward_batch(mkckks_ctx *c, uint64_t *d, uint32_t n_polys, uint32_t nl, int with_p) {
    return guarded([&] {
        need(c && d, "null argument");
        c->eng->ntt_forward(d, n_polys, nl, with_p != 0);
    });
}
int mkckks_ntt_inverse_batch(mkckks_ctx *c, uint64_t *d, uint32_t n_polys, uint32_t nl, int with_p) {
    return guarded([&] {
        need(c && d, "null argument");
        c->eng->ntt_inverse(d, n_polys, nl, with_p != 0);
    });
}

int mkckks_eval_add_batch(mkckks_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out, uint32_t n_ct,
                          uint32_t nl) {
    return guarded([&] {
        need(c && a && b && out, "null argument");
        c->eng->eval_add(a, b, out, n_ct, nl);
    });
}
int mkckks_eval_sum_batch(mkckks_ctx *c, const uint64_t *in, uint64_t *out, uint32_t n_clients, uint32_t n_ct,
                          uint32_t nl) {
    return guarded([&] {
        need(c && in && out, "null argument");
        c->eng->eval_sum(in, out, n_clients, n_ct, nl);
    });
}

int mkckks_rescale_mult_const_batch(mkckks_ctx *c, const uint64_t *in, uint64_t *out, uint32_t n_ct, uint32_t nl,
                                    double operand) {
    return guarded([&] {
        need(c && in && out, "null argument");
        const mk::ParamSet &ps = c->eng->params();
        need(nl >= 2 && nl <= ps.L, "nl out of range");
        // after dropping limb nl-1 the ciphertext sits at level L-(nl-1)
        std::vector<uint64_t> f = ps.const_factors(nl - 1, ps.L - (nl - 1), operand);
        c->eng->rescale(in, out, n_ct, nl, &f);
    });
}
int mkckks_rescale_batch(mkckks_ctx *c, const uint64_t *in, uint64_t *out, uint32_t n_ct, uint32_t nl) {
    return guarded([&] {
        need(c && in && out, "null argument");
        c->eng->rescale(in, out, n_ct, nl, nullptr);
    });
}
int mkckks_mult_const_batch(mkckks_ctx *c, uint64_t *ct, uint32_t n_ct, uint32_t nl, double operand) {
    return guarded([&] {
        need(c && ct, "null argument");
        const mk::ParamSet &ps = c->eng->params();
        need(nl >= 1 && nl <= ps.L, "nl out of range");
        c->eng->mult_const(ct, n_ct, nl, ps.const_factors(nl, ps.L - nl, operand));
    });
}

int mkckks_reencrypt_batch(mkckks_ctx *c, const uint64_t *ct, const uint64_t *evk, uint64_t *out, uint32_t n_ct,
                           uint32_t nl) {
    return guarded([&] {
        need(c && ct && evk && out, "null argument");
        c->eng->reencrypt(ct, evk, out, n_ct, nl);
    });
}
int mkckks_reencrypt_accumulate_batch(mkckks_ctx *c, const uint64_t *ct, const uint64_t *evk, uint64_t *acc,
                                      uint32_t n_ct, uint32_t nl) {
    return guarded([&] {
        need(c && ct && evk && acc, "null argument");
        c->eng->reencrypt(ct, evk, acc, n_ct, nl, true);
    });
}
int mkckks_reencrypt_sum_batch(mkckks_ctx *c, const uint64_t *cts, const uint64_t *evks, uint64_t *out, uint32_t n_clients,
                               uint32_t n_ct, uint32_t nl) {
    return guarded([&] {
        need(c && cts && evks && out, "null argument");
        c->eng->reencrypt_sum(cts, evks, out, n_clients, n_ct, nl);
    });
}
int mkckks_modup_batch(mkckks_ctx *c, const uint64_t *c1, uint64_t *digits, uint32_t n, uint32_t nl) {
    return guarded([&] {
        need(c && c1 && digits, "null argument");
        c->eng->modup(c1, digits, n, nl);
    });
}
int mkckks_moddown_batch(mkckks_ctx *c, const uint64_t *in, uint64_t *out, uint32_t n, uint32_t nl) {
    return guarded([&] {
        need(c && in && out, "null argument");
        c->eng->moddown(in, out, n, nl);
    });
}

int mkckks_keygen(mkckks_ctx *c, const int8_t *s, const uint64_t *a, const int32_t *e, uint64_t *pk, uint64_t *sk) {
    return guarded([&] {
        need(c && s && a && e && pk && sk, "null argument");
        c->eng->keygen(s, a, e, pk, sk);
    });
}
int mkckks_rekeygen(mkckks_ctx *c, const int8_t *s_old, const uint64_t *pk_new, const int8_t *u, const int32_t *e0,
                    const int32_t *e1, uint64_t *evk) {
    return guarded([&] {
        need(c && s_old && pk_new && u && e0 && e1 && evk, "null argument");
        c->eng->rekeygen(s_old, pk_new, u, e0, e1, evk);
    });
}
int mkckks_encrypt_batch(mkckks_ctx *c, const uint64_t *pk, const uint64_t *pt, const int8_t *v, const int32_t *e0,
                         const int32_t *e1, uint64_t *ct, uint32_t n_ct, uint32_t nl) {
    return guarded([&] {
        need(c && pk && pt && v && e0 && e1 && ct, "null argument");
        c->eng->encrypt(pk, pt, v, e0, e1, ct, n_ct, nl);
    });
}
int mkckks_lift_ntt_batch(mkckks_ctx *c, const double *coef, uint64_t *out, uint32_t n, uint32_t nl) {
    return guarded([&] {
        need(c && coef && out, "null argument");
        c->eng->lift_ntt(coef, out, n, nl);
    });
}
int mkckks_sample_ternary(mkckks_ctx *c, int8_t *out, size_t count, const uint8_t *h_key32, uint32_t stream_id) {
    return guarded([&] {
        need(c && out && h_key32, "null argument");
        c->eng->sample_ternary(out, count, h_key32, stream_id);
    });
}
int mkckks_sample_gauss(mkckks_ctx *c, int32_t *out, size_t count, double sigma, const uint8_t *h_key32,
                        uint32_t stream_id) {
    return guarded([&] {
        need(c && out && h_key32, "null argument");
        c->eng->sample_gauss(out, count, sigma, h_key32, stream_id);
    });
}
int mkckks_sample_uniform(mkckks_ctx *c, uint64_t *out, uint32_t n_polys, uint32_t nl, int with_p, const uint8_t *h_key32,
                          uint32_t stream_id) {
    return guarded([&] {
        need(c && out && h_key32, "null argument");
        c->eng->sample_uniform(out, n_polys, nl, with_p != 0, h_key32, stream_id);
    });
}
int mkckks_chacha20_block(mkckks_ctx *c, uint32_t *d_out16, const uint8_t *h_key32, uint32_t counter,
                          const uint32_t *h_nonce3) {
    return guarded([&] {
        need(c && d_out16 && h_key32 && h_nonce3, "null argument");
        c->eng->chacha_block(d_out16, h_key32, counter, h_nonce3);
    });
}
int mkckks_encode_batch(mkckks_ctx *c, const double *vals, uint64_t *pt, uint32_t n, uint32_t nl, double scale) {
    return guarded([&] {
        need(c && vals && pt, "null argument");
        need(scale > 0, "scale must be positive");
        c->eng->encode(vals, pt, n, nl, scale);
    });
}
int mkckks_decode_batch(mkckks_ctx *c, const uint64_t *m, double *vals, uint32_t n, uint32_t nl, double scale) {
    return guarded([&] {
        need(c && m && vals, "null argument");
        need(scale > 0, "scale must be positive");
        c->eng->decode(m, vals, n, nl, scale);
    });
}
int mkckks_decrypt_batch(mkckks_ctx *c, const uint64_t *ct, const uint64_t *sk, uint64_t *m, uint32_t n_ct,
                         uint32_t nl) {
    return guarded([&] {
        need(c && ct && sk && m, "null argument");
        c->eng->decrypt(ct, sk, m, n_ct, nl);
    });
}
int mkckks_reduce_mod_batch(mkckks_ctx *c, uint64_t *ct, uint32_t n_ct, uint32_t nl, uint32_t n_terms) {
    return guarded([&] {
        need(c && ct, "null argument");
        c->eng->reduce_mod(ct, n_ct, nl, n_terms);
    });
}

int mkckks_comm_unique_id(void *h_id) {
    return guarded([&] {
        need(h_id, "null argument");
        mk::comm_unique_id(h_id);
    });
}
int mkckks_comm_create(mkckks_ctx *c, const void *h_id, int n_ranks, int rank, void **comm_out) {
    return guarded([&] {
        need(c && h_id && comm_out, "null argument");
        if (!c->eng->has_device()) throw mk::NoDevice("host-only context: no communicator");
        *comm_out = mk::comm_create(c->eng->device(), h_id, n_ranks, rank);
    });
}
int mkckks_comm_destroy(mkckks_ctx *c, void *comm) {
    return guarded([&] {
        need(c != nullptr, "null argument");
        mk::comm_destroy(comm);
    });
}
int mkckks_reduce_scatter_sum_mod(mkckks_ctx *c, void *comm, const uint64_t *d_partial, uint64_t *d_shard,
                                  uint32_t n_ct_shard, uint32_t nl, uint32_t n_ranks) {
    return guarded([&] {
        need(c && comm && d_partial && d_shard, "null argument");
        c->eng->reduce_scatter_sum_mod(comm, d_partial, d_shard, n_ct_shard, nl, n_ranks);
    });
}
const char *mkckks_comm_library(void) {
    try {
        return mk::comm_library_path();
    } catch (const std::exception &e) {
        g_err = e.what();
        return "";
    }
}

}  // extern "C"
