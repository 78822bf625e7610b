/* cpu_fast.c -- TEST INFRASTRUCTURE, like everything under oracle/: a faster CPU version of orc_reencrypt, used by
 * bench.py's cpu_baseline leg (and checked against the restatement in tests/test_oracle_algebra.py) so that the CPU
 * number beside the GPU number is not the naive one.  Never linked into or called from the product.
 *
 * Same algorithm, same order of operations, same residues as mkckks_oracle.c's orc_reencrypt ([upstream]
 * keyswitch-hybrid.cpp KeySwitchHYBRID::EvalKeySwitchPrecomputeCore / EvalFastKeySwitchCoreExt, dcrtpoly-impl.h
 * ApproxSwitchCRTBasis / ApproxModDown; reached from changeCipherDomain.cpp:74) -- what changes is how the arithmetic is
 * carried out, the way OpenFHE's own native backend does it:
 *   - Harvey lazy butterflies (values below 4q forward, 2q inverse, one correction pass at the end) instead of a
 *     conditional subtraction per operation ([upstream] transformnat-impl.h uses the same Shoup constants);
 *   - every base-conversion table (S/s_i)^-1, [S/s_i]_t, P^-1 with its Shoup companion is built once per (context,
 *     level) instead of once per call;
 *   - products of two variable residues (eval-key inner product) through a Barrett constant instead of the 128-bit
 *     division of `mulmod`; constant-times-variable products through Shoup companions;
 *   - scratch buffers are allocated once per (context, level).
 * The restatement stays the checker: this file includes it, so both share the parameter generation and the twiddle
 * tables, and tests compare fcpu_reencrypt with orc_reencrypt word for word. */
#include "mkckks_oracle.c"

typedef struct { u64 m, mu; uint32_t k; } bar_t;

static bar_t bar_make(u64 m) {
    bar_t b;
    b.m = m;
    b.k = 64 - (uint32_t)__builtin_clzll(m);
    b.mu = (u64)(((u128)1 << (2 * b.k)) / m); /* < 2^(k+1) */
    return b;
}
/* a * b mod m for a, b < m < 2^61: q = floor(floor(x / 2^(k-1)) mu / 2^(k+1)) is the quotient or up to 2 below it */
static inline u64 bar_mulmod(u64 a, u64 b, bar_t B) {
    const u128 x = (u128)a * b;
    const u64 q = (u64)(((u128)(u64)(x >> (B.k - 1)) * B.mu) >> (B.k + 1));
    u64 r = (u64)x - q * B.m;
    while (r >= B.m) r -= B.m;
    return r;
}
static inline u64 shoup_lazy2(u64 a, u64 w, u64 wpre, u64 m) { /* any a: result in [0, 2m) */
    const u64 q = (u64)(((u128)a * wpre) >> 64);
    return a * w - q * m;
}

/* forward: natural in (canonical), bit-reversed out (canonical); lazily below 4m in between */
static void fast_ntt_fwd(const orc_ctx *c, uint32_t limb, u64 *a) {
    const u64 m = c->mod[limb], m2 = 2 * m;
    const u64 *tw = c->tw[limb], *twp = c->twp[limb];
    const uint32_t n = c->n;
    uint32_t t = n;
    for (uint32_t mm = 1; mm < n; mm <<= 1) {
        t >>= 1;
        for (uint32_t i = 0; i < mm; i++) {
            const u64 w = tw[mm + i], wp = twp[mm + i];
            u64 *x = a + 2 * i * t, *y = x + t;
            for (uint32_t j = 0; j < t; j++) {
                u64 u = x[j];
                u = u >= m2 ? u - m2 : u;
                const u64 v = shoup_lazy2(y[j], w, wp, m);
                x[j] = u + v;
                y[j] = u - v + m2;
            }
        }
    }
    for (uint32_t j = 0; j < n; j++) {
        u64 v = a[j];
        v = v >= m2 ? v - m2 : v;
        a[j] = v >= m ? v - m : v;
    }
}
/* inverse: bit-reversed in, natural out, scaled by N^-1 (times an extra constant `sc` when sc != 0), canonical */
static void fast_ntt_inv(const orc_ctx *c, uint32_t limb, u64 *a, u64 sc) {
    const u64 m = c->mod[limb], m2 = 2 * m;
    const u64 *tw = c->itw[limb], *twp = c->itwp[limb];
    const uint32_t n = c->n;
    uint32_t t = 1;
    for (uint32_t mm = n; mm > 1; mm >>= 1) {
        const uint32_t h = mm >> 1;
        for (uint32_t i = 0; i < h; i++) {
            const u64 w = tw[h + i], wp = twp[h + i];
            u64 *x = a + 2 * i * t, *y = x + t;
            for (uint32_t j = 0; j < t; j++) {
                const u64 u = x[j], v = y[j];
                u64 s = u + v;
                x[j] = s >= m2 ? s - m2 : s;
                y[j] = shoup_lazy2(u - v + m2, w, wp, m);
            }
        }
        t <<= 1;
    }
    u64 f = c->ninv[limb];
    if (sc) f = mulmod(f, sc, m);
    const u64 fp = shoup_pre(f, m);
    for (uint32_t j = 0; j < n; j++) {
        const u64 v = shoup_lazy2(a[j], f, fp, m);
        a[j] = v >= m ? v - m : v;
    }
}

/* tables and scratch of one (context, level) */
typedef struct {
    const orc_ctx *c;
    uint32_t nl, nparts, ext;
    /* ModUp, per part: sources [lo, hi), per source (S/s_i)^-1 mod s_i; per (source, target slot) [S/s_i]_t + Shoup */
    uint32_t lo[8], hi[8];
    u64 hatinv[8][8];
    u64 *hat[8], *hatp[8]; /* [sz][ext] indexed by ext slot (own slots unused) */
    /* ModDown: per P limb (P/p_k)^-1 mod p_k; per (k, i) [P/p_k]_{q_i} + Shoup; per i P^-1 mod q_i + Shoup */
    u64 phatinv[8];
    u64 *phat, *phatp, *pinv, *pinvp;
    bar_t *bar;            /* [ext] */
    u64 *coef, *digits, *til0, *til1, *pc, *conv;
} fast_ws;

static fast_ws g_ws[8];
static int g_nws = 0;

static uint32_t limb_of(const orc_ctx *c, uint32_t nl, uint32_t slot) { return slot < nl ? slot : c->L + (slot - nl); }

static fast_ws *fast_prepare(const orc_ctx *c, uint32_t nl) {
    for (int i = 0; i < g_nws; i++)
        if (g_ws[i].c == c && g_ws[i].nl == nl) return &g_ws[i];
    if (g_nws == 8) g_nws = 0; /* bench and tests use a handful of levels */
    fast_ws *w = &g_ws[g_nws++];
    memset(w, 0, sizeof *w);
    const uint32_t n = c->n, K = c->K, L = c->L, alpha = c->alpha;
    w->c = c;
    w->nl = nl;
    w->ext = nl + K;
    w->nparts = (nl + alpha - 1) / alpha;
    if (w->nparts > c->beta) w->nparts = c->beta;
    for (uint32_t p = 0; p < w->nparts; p++) {
        const uint32_t lo = p * alpha, hi = lo + alpha > nl ? nl : lo + alpha, sz = hi - lo;
        w->lo[p] = lo;
        w->hi[p] = hi;
        w->hat[p] = (u64 *)calloc((size_t)sz * w->ext, sizeof(u64));
        w->hatp[p] = (u64 *)calloc((size_t)sz * w->ext, sizeof(u64));
        for (uint32_t i = 0; i < sz; i++) {
            const u64 si = c->mod[lo + i];
            u64 h = 1;
            for (uint32_t k = 0; k < sz; k++)
                if (k != i) h = mulmod(h, c->mod[lo + k] % si, si);
            w->hatinv[p][i] = invmod(h, si);
            for (uint32_t s = 0; s < w->ext; s++) {
                if (s >= lo && s < hi) continue;
                const u64 t = c->mod[limb_of(c, nl, s)];
                u64 g = 1;
                for (uint32_t k = 0; k < sz; k++)
                    if (k != i) g = mulmod(g, c->mod[lo + k] % t, t);
                w->hat[p][i * w->ext + s] = g;
                w->hatp[p][i * w->ext + s] = shoup_pre(g, t);
            }
        }
    }
    w->phat = (u64 *)calloc((size_t)K * nl, sizeof(u64));
    w->phatp = (u64 *)calloc((size_t)K * nl, sizeof(u64));
    w->pinv = (u64 *)calloc(nl, sizeof(u64));
    w->pinvp = (u64 *)calloc(nl, sizeof(u64));
    for (uint32_t k = 0; k < K; k++) {
        const u64 pk = c->mod[L + k];
        u64 h = 1;
        for (uint32_t j = 0; j < K; j++)
            if (j != k) h = mulmod(h, c->mod[L + j] % pk, pk);
        w->phatinv[k] = invmod(h, pk);
        for (uint32_t i = 0; i < nl; i++) {
            const u64 qi = c->mod[i];
            u64 g = 1;
            for (uint32_t j = 0; j < K; j++)
                if (j != k) g = mulmod(g, c->mod[L + j] % qi, qi);
            w->phat[k * nl + i] = g;
            w->phatp[k * nl + i] = shoup_pre(g, qi);
        }
    }
    for (uint32_t i = 0; i < nl; i++) {
        const u64 qi = c->mod[i];
        u64 pinv = 1;
        for (uint32_t k = 0; k < K; k++) pinv = mulmod(pinv, c->mod[L + k] % qi, qi);
        w->pinv[i] = invmod(pinv, qi);
        w->pinvp[i] = shoup_pre(w->pinv[i], qi);
    }
    w->bar = (bar_t *)calloc(w->ext, sizeof(bar_t));
    for (uint32_t s = 0; s < w->ext; s++) w->bar[s] = bar_make(c->mod[limb_of(c, nl, s)]);
    w->coef = (u64 *)malloc((size_t)alpha * n * sizeof(u64));
    w->digits = (u64 *)malloc((size_t)w->nparts * w->ext * n * sizeof(u64));
    w->til0 = (u64 *)malloc((size_t)w->ext * n * sizeof(u64));
    w->til1 = (u64 *)malloc((size_t)w->ext * n * sizeof(u64));
    w->pc = (u64 *)malloc((size_t)K * n * sizeof(u64));
    w->conv = (u64 *)malloc((size_t)nl * n * sizeof(u64));
    return w;
}

/* ApproxModDown of one accumulator over Q_l P: out_i = (in_i - NTT(conv_i)) P^-1.  The factor (P/p_k)^-1 of the
 * conversion rides on the inverse transform's N^-1 scaling. */
static void fast_moddown(fast_ws *w, const u64 *in, u64 *out) {
    const orc_ctx *c = w->c;
    const uint32_t n = c->n, K = c->K, L = c->L, nl = w->nl;
#pragma omp parallel for
    for (uint32_t k = 0; k < K; k++) {
        memcpy(w->pc + (size_t)k * n, in + (size_t)(nl + k) * n, n * sizeof(u64));
        fast_ntt_inv(c, L + k, w->pc + (size_t)k * n, w->phatinv[k]);
    }
#pragma omp parallel for
    for (uint32_t i = 0; i < nl; i++) {
        const u64 qi = c->mod[i];
        u64 *y = w->conv + (size_t)i * n;
        for (uint32_t r = 0; r < n; r++) {
            u64 acc = 0;
            for (uint32_t k = 0; k < K; k++) {
                u64 v = shoup_lazy2(w->pc[(size_t)k * n + r], w->phat[k * nl + i], w->phatp[k * nl + i], qi);
                v = v >= qi ? v - qi : v;
                acc = addmod(acc, v, qi);
            }
            y[r] = acc;
        }
        fast_ntt_fwd(c, i, y);
        const u64 *src = in + (size_t)i * n;
        u64 *dst = out + (size_t)i * n;
        const u64 pi = w->pinv[i], pip = w->pinvp[i];
        for (uint32_t r = 0; r < n; r++) {
            const u64 v = shoup_lazy2(submod(src[r], y[r], qi), pi, pip, qi);
            dst[r] = v >= qi ? v - qi : v;
        }
    }
}

/* == orc_reencrypt(c, nl, ct, evk, out), word for word */
void fcpu_reencrypt(const orc_ctx *c, uint32_t nl, const u64 *ct, const u64 *evk, u64 *out) {
    fast_ws *w = fast_prepare(c, nl);
    const uint32_t n = c->n, D = c->D, ext = w->ext;
    const u64 *c0 = ct, *c1 = ct + (size_t)nl * n;
    for (uint32_t p = 0; p < w->nparts; p++) {
        const uint32_t lo = w->lo[p], hi = w->hi[p], sz = hi - lo;
        u64 *dig = w->digits + (size_t)p * ext * n;
#pragma omp parallel for
        for (uint32_t i = 0; i < sz; i++) {
            memcpy(w->coef + (size_t)i * n, c1 + (size_t)(lo + i) * n, n * sizeof(u64));
            fast_ntt_inv(c, lo + i, w->coef + (size_t)i * n, w->hatinv[p][i]);
        }
#pragma omp parallel for
        for (uint32_t s = 0; s < ext; s++) {
            u64 *y = dig + (size_t)s * n;
            if (s >= lo && s < hi) {
                memcpy(y, c1 + (size_t)s * n, n * sizeof(u64));
                continue;
            }
            const uint32_t idx = limb_of(c, nl, s);
            const u64 t = c->mod[idx];
            for (uint32_t r = 0; r < n; r++) {
                u64 acc = 0;
                for (uint32_t i = 0; i < sz; i++) {
                    u64 v = shoup_lazy2(w->coef[(size_t)i * n + r], w->hat[p][i * ext + s], w->hatp[p][i * ext + s], t);
                    v = v >= t ? v - t : v;
                    acc = addmod(acc, v, t);
                }
                y[r] = acc;
            }
            fast_ntt_fwd(c, idx, y);
        }
    }
#pragma omp parallel for
    for (uint32_t s = 0; s < ext; s++) {
        const uint32_t idx = limb_of(c, nl, s);
        const u64 m = c->mod[idx];
        const bar_t B = w->bar[s];
        u64 *t0 = w->til0 + (size_t)s * n, *t1 = w->til1 + (size_t)s * n;
        for (uint32_t j = 0; j < w->nparts; j++) {
            const u64 *d = w->digits + ((size_t)j * ext + s) * n;
            const u64 *b = evk + (((size_t)j * 2 + 0) * D + idx) * n;
            const u64 *a = evk + (((size_t)j * 2 + 1) * D + idx) * n;
            if (j == 0) {
                for (uint32_t r = 0; r < n; r++) {
                    t0[r] = bar_mulmod(d[r], b[r], B);
                    t1[r] = bar_mulmod(d[r], a[r], B);
                }
            } else {
                for (uint32_t r = 0; r < n; r++) {
                    t0[r] = addmod(t0[r], bar_mulmod(d[r], b[r], B), m);
                    t1[r] = addmod(t1[r], bar_mulmod(d[r], a[r], B), m);
                }
            }
        }
    }
    u64 *o0 = out, *o1 = out + (size_t)nl * n;
    fast_moddown(w, w->til0, o0);
    fast_moddown(w, w->til1, o1);
#pragma omp parallel for
    for (uint32_t i = 0; i < nl; i++) {
        const u64 m = c->mod[i];
        for (uint32_t r = 0; r < n; r++) o0[(size_t)i * n + r] = addmod(o0[(size_t)i * n + r], c0[(size_t)i * n + r], m);
    }
}

void fcpu_ntt_fwd(const orc_ctx *c, uint32_t limb, u64 *a) { fast_ntt_fwd(c, limb, a); }
void fcpu_ntt_inv(const orc_ctx *c, uint32_t limb, u64 *a) { fast_ntt_inv(c, limb, a, 0); }
