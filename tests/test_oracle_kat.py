"""Pins the oracle (oracle/mkckks_oracle.c) against the reference's own data fixtures.

No GPU. These are the only result-pinning vectors the reference holds for this
path: CC.json parameters, both clients' secret keys in EVALUATION
form, and plaintext-in / decrypted-out weights.
"""
import json
import os

import numpy as np
import pytest

from oracle.oracle import OracleContext, sample_gauss, sample_ternary, sample_uniform


def test_parameter_kat(ref_ctx, golden_dir):
    # server/storage/CC.json:29-216 (P1-P3): ring dim, 4 Q moduli, their minimal 2N-th roots
    cc = json.load(open(os.path.join(golden_dir, "cc_params.json")))
    assert ref_ctx.N == cc["ring_dim"]
    assert ref_ctx.L == len(cc["moduli"]) == 4
    assert [int(x) for x in ref_ctx.moduli[:4]] == cc["moduli"]
    assert [int(x) for x in ref_ctx.roots[:4]] == cc["roots"]
    # composite modulus 'cm' = product of the 4 limbs (160 bits)
    prod = 1
    for m in cc["moduli"]:
        prod *= m
    words = cc["composite_modulus_words"]
    assert prod == sum(int(w) << (64 * i) for i, w in enumerate(words))
    assert prod.bit_length() == cc["composite_modulus_bits"]
    # P7: pk / re-key live over QP = 6 limbs, beta = 2 digits of alpha = 2
    assert (ref_ctx.K, ref_ctx.alpha, ref_ctx.beta) == (2, 2, 2)
    # P5: special primes = next two primes = 1 (mod 2N) below q0
    assert [int(x) for x in ref_ctx.moduli[4:]] == [1152921504606683137, 1152921504606584833]
    for q, r in zip(ref_ctx.moduli, ref_ctx.roots):
        q, r = int(q), int(r)
        assert (q - 1) % (2 * ref_ctx.N) == 0
        assert pow(r, ref_ctx.N, q) == q - 1


@pytest.mark.parametrize("client", [1, 2])
def test_ntt_kat_secret_keys(ref_ctx, golden_dir, client):
    # client_{1,2}-private.key "s": 4 limbs x 16384 residues in EVALUATION format (P4).
    k = np.load(os.path.join(golden_dir, "sk_ntt_kat.npz"))
    ev = k[f"sk{client}_eval"]
    assert ev.shape == (4, ref_ctx.N)
    assert np.array_equal(k[f"sk{client}_moduli"], ref_ctx.moduli[:4])
    tern = None
    for l in range(4):
        q = int(ref_ctx.moduli[l])
        co = ref_ctx.ntt_inv(l, ev[l])
        s = np.where(co > q // 2, -1, co.astype(np.int64)).astype(np.int64)
        assert np.all((co == 0) | (co == 1) | (co == q - 1)), "inverse NTT must give a ternary polynomial"
        if tern is None:
            tern = s
        else:
            assert np.array_equal(tern, s), "same ternary vector in every limb"
        # forward transform of the recovered coefficients reproduces the fixture bit for bit
        assert np.array_equal(ref_ctx.ntt_fwd(l, co), ev[l])
    counts = np.bincount(tern + 1, minlength=3)
    expect = {1: [5487, 5285, 5612], 2: [5442, 5446, 5496]}[client]
    assert counts.tolist() == expect


def _keys(ctx, rng):
    s = sample_ternary(rng, ctx.N)
    pk, sk = ctx.keygen(s, sample_uniform(rng, ctx.moduli, ctx.N), sample_gauss(rng, ctx.N))
    return s, pk, sk


def _rekey(ctx, rng, s_old, pk_new):
    u = np.stack([sample_ternary(rng, ctx.N) for _ in range(ctx.beta)])
    e0 = np.stack([sample_gauss(rng, ctx.N) for _ in range(ctx.beta)])
    e1 = np.stack([sample_gauss(rng, ctx.N) for _ in range(ctx.beta)])
    return ctx.rekeygen(s_old, pk_new, u, e0, e1)


def _enc(ctx, rng, pk, vals):
    pt = ctx.encode(vals, ctx.sf_big(0), ctx.L)
    return ctx.encrypt(pk, pt, sample_ternary(rng, ctx.N), sample_gauss(rng, ctx.N), sample_gauss(rng, ctx.N))


def test_end_to_end_kat(ref_ctx, golden_dir):
    """P8: encrypt c1,c2 -> PRE c1->c2 -> EvalAdd -> EvalMult(0.5) -> PRE c2->c1 -> decrypt.

    Same flow as orchestration/run.sh:28-44.  The reference's decrypted output equals the
    plaintext mean within 1.4e-8 (tensors) -- the scheme's own precision at Delta ~ 2^40;
    the restatement must land in the same band (tolerance 2^-25 ~ 3e-8).
    """
    ctx = ref_ctx
    W = np.load(os.path.join(golden_dir, "e2e_weights.npz"))
    rng = np.random.default_rng(20250919)
    s1, pk1, sk1 = _keys(ctx, rng)
    s2, pk2, sk2 = _keys(ctx, rng)
    rk12 = _rekey(ctx, rng, s1, pk2)
    rk21 = _rekey(ctx, rng, s2, pk1)
    tol = 2.0 ** -25
    for name in ("param_2", "param_6", "param_7", "param_1"):
        v1 = W[f"sample_c1_{name}_values"]
        v2 = W[f"sample_c2_{name}_values"]
        mean = (v1 + v2) / 2
        # the reference's own outputs sit within the scheme's precision of the mean
        assert np.abs(W[f"decrypted_c1_{name}_values"] - mean).max() < tol
        assert np.abs(W[f"decrypted_c2_{name}_values"] - mean).max() < tol
        ct1 = _enc(ctx, rng, pk1, v1)
        ct2 = _enc(ctx, rng, pk2, v2)
        assert ct1.shape == (2, 4, ctx.N)  # P6: fresh ct has 4 limbs
        ct1to2 = ctx.reencrypt(ct1, rk12)
        summed = ctx.eval_add(ct1to2, ct2)
        resc = ctx.rescale(summed)
        assert resc.shape == (2, 3, ctx.N)  # P6: 3 limbs after EvalMult(.,0.5)
        agg = ctx.mult_factors(resc, ctx.const_factors(3, 1, 0.5))
        scale = ctx.sf(1) * ctx.sf(1)
        back = ctx.reencrypt(agg, rk21)
        d2 = ctx.decrypt_decode(agg, sk2, scale)[: v1.size]
        d1 = ctx.decrypt_decode(back, sk1, scale)[: v1.size]
        assert np.abs(d2 - mean).max() < tol
        assert np.abs(d1 - mean).max() < tol
        # and within the same band of the reference's decrypted files
        assert np.abs(d1 - W[f"decrypted_c1_{name}_values"]).max() < 2 * tol
        assert np.abs(d2 - W[f"decrypted_c2_{name}_values"]).max() < 2 * tol


def test_mean_std_scalars(ref_ctx, golden_dir):
    # encryptModelWeights.cpp:82,90: {mean} / {std_dev} packed as 1 value + zero padding;
    # decryptModelWeights.cpp:82-83: SetLength(1), first slot returned.
    ctx = ref_ctx
    W = np.load(os.path.join(golden_dir, "e2e_weights.npz"))
    rng = np.random.default_rng(7)
    s1, pk1, sk1 = _keys(ctx, rng)
    for name in ("param_0", "param_5"):
        ms1 = W[f"sample_c1_{name}_mean_std"]
        ms2 = W[f"sample_c2_{name}_mean_std"]
        ref = W[f"decrypted_c1_{name}_mean_std"]
        for k in range(2):
            a = _enc(ctx, rng, pk1, np.array([ms1[k]]))
            b = _enc(ctx, rng, pk1, np.array([ms2[k]]))
            r = ctx.rescale(ctx.eval_add(a, b))
            agg = ctx.mult_factors(r, ctx.const_factors(3, 1, 0.5))
            d = ctx.decrypt_decode(agg, sk1, ctx.sf(1) ** 2)
            assert abs(d[0] - (ms1[k] + ms2[k]) / 2) < 2.0 ** -25
            assert abs(ref[k] - (ms1[k] + ms2[k]) / 2) < 2.0 ** -25
            assert np.abs(d[1:]).max() < 2.0 ** -25  # padding slots decrypt to ~0
