"""GPU parity: every C-ABI entry point vs the oracle, bit-exact on the same seeded inputs.

Runs on a real MI355X (`-m gpu`).  All calls go through libmkckks_hip.so via ctypes
(ppqsflhe_amd.binding); the oracle (oracle/liboracle.so) is only the checker.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle.oracle import OracleContext, sample_gauss, sample_ternary, sample_uniform  # noqa: E402

CONFIGS = {
    # name: (log_n, depth, scaling_bits, first_bits, dnum)
    "tiny": (10, 3, 40, 60, 2),       # L=5, K=3, alpha=3, beta=2
    "c1": (12, 1, 40, 60, 2),         # BASELINE configs[0]: N=2^12, depth 1
    "ref": (14, 2, 40, 60, 2),        # the reference's own CC.json
    "c3": (16, 10, 50, 60, 3),        # BASELINE configs[1..2]: N=2^16, L=12, dnum=3
    "c5s": (12, 18, 50, 60, 3),       # BASELINE configs[4] limb structure (L=20, alpha=7, K=7) at a small ring
    "n17": (17, 2, 50, 60, 2),        # BASELINE configs[4] ring dimension (radix column pass + generic row pass)
    "n11": (11, 2, 40, 60, 2),        # odd log N: generic kernels for both passes, integer arithmetic only
}


@pytest.fixture(scope="module")
def ctxs():
    from ppqsflhe_amd import Context
    cache = {}

    def get(name):
        if name not in cache:
            a = CONFIGS[name]
            cache[name] = (Context(a[0], a[1], a[2], a[3], dnum=a[4], device=0),
                           OracleContext(a[0], a[1], a[2], a[3], dnum=a[4]))
        return cache[name]

    yield get
    for g, _ in cache.values():
        g.close()


def rand_polys(rng, ctx, limb_ids, count):
    out = np.empty((count, len(limb_ids), ctx.N), dtype=np.uint64)
    for j, l in enumerate(limb_ids):
        out[:, j, :] = rng.integers(0, int(ctx.moduli[l]), size=(count, ctx.N), dtype=np.uint64)
    return out


def rand_ct(rng, ctx, nl, count):
    return rand_polys(rng, ctx, list(range(nl)) * 2, count).reshape(count, 2, nl, ctx.N)


@pytest.mark.parametrize("name", ["tiny", "c1", "ref", "c3"])
def test_ntt_roundtrip_and_parity(ctxs, name):
    g, o = ctxs(name)
    rng = np.random.default_rng(11)
    ids = list(range(g.L)) + list(range(g.L, g.D))
    x = rand_polys(rng, g, ids, 2)  # [2][D][N]
    d = g.to_device(x)
    g.ntt_forward(d, 2, g.L, with_p=True)
    fwd = d.to_host()
    for p in range(2):
        for j, l in enumerate(ids):
            assert np.array_equal(fwd[p, j], o.ntt_fwd(l, x[p, j])), (name, p, l)
    g.ntt_inverse(d, 2, g.L, with_p=True)
    assert np.array_equal(d.to_host(), x)
    # inverse parity on arbitrary (non-transform) input
    d.upload(x)
    g.ntt_inverse(d, 2, g.L, with_p=True)
    inv = d.to_host()
    for j, l in enumerate(ids):
        assert np.array_equal(inv[0, j], o.ntt_inv(l, x[0, j]))


@pytest.mark.parametrize("log_n,generic", [(9, False), (11, False), (13, False), (17, False), (12, True), (16, True)])
def test_ntt_other_sizes_and_generic_kernels(log_n, generic, monkeypatch):
    # odd log N and N = 2^17 take the generic LDS-stage kernels for one or both passes; MKCKKS_GENERIC_NTT=1
    # forces them everywhere.  Both schedules must give the same bits as the oracle.
    from ppqsflhe_amd import Context
    if generic:
        monkeypatch.setenv("MKCKKS_GENERIC_NTT", "1")
    g = Context(log_n, 1, 40, 60, dnum=2, device=0)
    o = OracleContext(log_n, 1, 40, 60, dnum=2)
    rng = np.random.default_rng(21)
    ids = list(range(g.L)) + list(range(g.L, g.D))
    x = rand_polys(rng, g, ids, 1)
    d = g.to_device(x)
    g.ntt_forward(d, 1, g.L, with_p=True)
    fwd = d.to_host()
    for j, l in enumerate(ids):
        assert np.array_equal(fwd[0, j], o.ntt_fwd(l, x[0, j])), (log_n, l)
    g.ntt_inverse(d, 1, g.L, with_p=True)
    assert np.array_equal(d.to_host(), x)
    g.close()


@pytest.mark.parametrize("name", ["tiny", "c3", "n17", "n11"])
def test_ntt_extreme_residues(ctxs, name):
    """Inputs that drive the lazy ranges of the butterflies to their bounds: every residue q - 1, alternating 0 / q - 1
    in several periods (sums and differences of maximal terms at every stage), and q - 1 at a single position."""
    g, o = ctxs(name)
    ids = list(range(g.L)) + list(range(g.L, g.D))
    pats = []
    idx = np.arange(g.N)
    for period in (1, 2, 16, 256, g.N // 2):
        pats.append(((idx // period) % 2 == 0))
    pats.append(np.ones(g.N, dtype=bool))
    pats.append(idx == g.N - 1)
    x = np.zeros((len(pats), len(ids), g.N), dtype=np.uint64)
    for pi, m in enumerate(pats):
        for j, l in enumerate(ids):
            x[pi, j, m] = int(g.moduli[l]) - 1
    d = g.to_device(x)
    g.ntt_forward(d, len(pats), g.L, with_p=True)
    fwd = d.to_host()
    for pi in range(len(pats)):
        for j, l in enumerate(ids):
            assert np.array_equal(fwd[pi, j], o.ntt_fwd(l, x[pi, j])), (name, pi, l)
    d.upload(x)
    g.ntt_inverse(d, len(pats), g.L, with_p=True)
    inv = d.to_host()
    for pi in range(len(pats)):
        for j, l in enumerate(ids):
            assert np.array_equal(inv[pi, j], o.ntt_inv(l, x[pi, j])), (name, pi, l)


@pytest.mark.parametrize("name,nl,C", [("c3", 12, 2), ("ref", 4, 3), ("n17", 4, 2)])
def test_reencrypt_sum_extreme_residues(ctxs, name, nl, C):
    """The whole key switch on maximal operands: ciphertexts and keys with every residue q - 1 (client 0) and with
    alternating 0 / q - 1 (the others)."""
    g, o = ctxs(name)
    B = 1
    cts = np.zeros((C, B, 2, nl, g.N), dtype=np.uint64)
    evks = np.zeros((C, g.beta, 2, g.D, g.N), dtype=np.uint64)
    idx = np.arange(g.N)
    for c in range(C):
        m = np.ones(g.N, dtype=bool) if c == 0 else ((idx // (1 << (3 * c))) % 2 == 0)
        for l in range(nl):
            cts[c, :, :, l, m] = int(g.moduli[l]) - 1
        for l in range(g.D):
            evks[c, :, :, l, m] = int(g.moduli[l]) - 1
    d_out = g.empty((B, 2, nl, g.N))
    g.reencrypt_sum(g.to_device(cts), g.to_device(evks), d_out, C, B, nl)
    acc = o.reencrypt(cts[0, 0], evks[0])
    for c in range(1, C):
        acc = o.eval_add(acc, o.reencrypt(cts[c, 0], evks[c]))
    assert np.array_equal(d_out.to_host()[0], acc)


@pytest.mark.parametrize("first_bits", [55, 58])
def test_first_modulus_below_60_bits(first_bits):
    """q_0 of 55 / 58 bits is still an integer-class limb of the form 2^k - c: the pseudo-Mersenne butterflies take their
    shifts from k.  Transforms and one key switch against the oracle."""
    from ppqsflhe_amd import Context
    g = Context(14, 2, 40, first_bits, dnum=2, device=0)
    o = OracleContext(14, 2, 40, first_bits, dnum=2)
    try:
        assert int(g.moduli[0]).bit_length() == first_bits
        rng = np.random.default_rng(first_bits)
        ids = list(range(g.L)) + list(range(g.L, g.D))
        x = rand_polys(rng, g, ids, 1)
        x[0, 0, ::2] = int(g.moduli[0]) - 1
        d = g.to_device(x)
        g.ntt_forward(d, 1, g.L, with_p=True)
        fwd = d.to_host()
        for j, l in enumerate(ids):
            assert np.array_equal(fwd[0, j], o.ntt_fwd(l, x[0, j])), l
        g.ntt_inverse(d, 1, g.L, with_p=True)
        assert np.array_equal(d.to_host(), x)
        nl = g.L
        ct = rand_ct(rng, g, nl, 1)
        evk = rand_polys(rng, g, list(range(g.D)) * (2 * g.beta), 1).reshape(g.beta, 2, g.D, g.N)
        d_out = g.empty((1, 2, nl, g.N))
        g.reencrypt(g.to_device(ct), g.to_device(evk), d_out, 1, nl)
        assert np.array_equal(d_out.to_host()[0], o.reencrypt(ct[0], evk))
    finally:
        g.close()


def test_ntt_golden_secret_keys(ctxs, golden_dir):
    # the reference's own vectors (client_{1,2}-private.key, P4) through the HIP kernels
    import os
    g, _ = ctxs("ref")
    k = np.load(os.path.join(golden_dir, "sk_ntt_kat.npz"))
    for c in (1, 2):
        ev = k[f"sk{c}_eval"]
        d = g.to_device(ev[None])
        g.ntt_inverse(d, 1, 4)
        co = d.to_host()[0]
        for l in range(4):
            q = int(g.moduli[l])
            assert np.all((co[l] == 0) | (co[l] == 1) | (co[l] == q - 1))
        g.ntt_forward(d, 1, 4)
        assert np.array_equal(d.to_host()[0], ev)


@pytest.mark.parametrize("name,nl", [("tiny", 5), ("tiny", 2), ("ref", 4), ("c3", 12)])
def test_eval_add_sum_reduce(ctxs, name, nl):
    g, o = ctxs(name)
    rng = np.random.default_rng(12)
    B, C = 3, 5
    cts = np.stack([rand_ct(rng, g, nl, B) for _ in range(C)])  # [C][B][2][nl][N]
    d_in = g.to_device(cts)
    d_out = g.empty((B, 2, nl, g.N))
    g.eval_add(d_in.view(0, cts[0].shape), d_in.view(cts[0].size, cts[0].shape), d_out, B, nl)
    got = d_out.to_host()
    for b in range(B):
        assert np.array_equal(got[b], o.eval_add(cts[0, b], cts[1, b]))
    g.eval_sum(d_in, d_out, C, B, nl)
    got = d_out.to_host()
    for b in range(B):
        acc = cts[0, b]
        for c in range(1, C):
            acc = o.eval_add(acc, cts[c, b])
        assert np.array_equal(got[b], acc)
    # integer-sum collective emulation: plain u64 sum then reduce_mod
    raw = cts.sum(axis=0, dtype=np.uint64)
    d_raw = g.to_device(raw)
    g.reduce_mod(d_raw, B, nl, C)
    assert np.array_equal(d_raw.to_host(), got)


@pytest.mark.parametrize("name,nl", [("tiny", 5), ("tiny", 3), ("c1", 3), ("ref", 4), ("c3", 12)])
def test_rescale_and_mult_const(ctxs, name, nl):
    g, o = ctxs(name)
    rng = np.random.default_rng(13)
    B = 2
    ct = rand_ct(rng, g, nl, B)
    d_in = g.to_device(ct)
    d_out = g.empty((B, 2, nl - 1, g.N))
    g.rescale(d_in, d_out, B, nl)
    got = d_out.to_host()
    exp = [o.rescale(ct[b]) for b in range(B)]
    for b in range(B):
        assert np.array_equal(got[b], exp[b])
    level = g.L - (nl - 1)
    for operand in (0.5, 1.0 / 64, -0.25):
        g.rescale_mult_const(d_in, d_out, B, nl, operand)
        got = d_out.to_host()
        f = o.const_factors(nl - 1, level, operand)
        for b in range(B):
            assert np.array_equal(got[b], o.mult_factors(exp[b], f))
    d_r = g.to_device(np.stack(exp))
    g.mult_const(d_r, B, nl - 1, 0.5)
    f = o.const_factors(nl - 1, level, 0.5)
    assert np.array_equal(d_r.to_host()[1], o.mult_factors(exp[1], f))


def make_keys(o, rng):
    N = o.N
    s = sample_ternary(rng, N)
    a = sample_uniform(rng, o.moduli, N)
    e = sample_gauss(rng, N)
    return s, a, e


def make_rk_rand(o, rng):
    u = np.stack([sample_ternary(rng, o.N) for _ in range(o.beta)])
    e0 = np.stack([sample_gauss(rng, o.N) for _ in range(o.beta)])
    e1 = np.stack([sample_gauss(rng, o.N) for _ in range(o.beta)])
    return u, e0, e1


@pytest.mark.parametrize("name", ["tiny", "c1", "ref", "c3"])
def test_keygen_rekeygen_encrypt_decrypt(ctxs, name):
    g, o = ctxs(name)
    rng = np.random.default_rng(14)
    N, D, L = g.N, g.D, g.L
    s1, a1, e1 = make_keys(o, rng)
    s2, a2, e2 = make_keys(o, rng)
    pk1_o, sk1_o = o.keygen(s1, a1, e1)
    pk2_o, sk2_o = o.keygen(s2, a2, e2)
    d_pk, d_sk = g.empty((2, D, N)), g.empty((D, N))
    g.keygen(g.to_device(s1), g.to_device(a1), g.to_device(e1), d_pk, d_sk)
    assert np.array_equal(d_pk.to_host(), pk1_o)
    assert np.array_equal(d_sk.to_host(), sk1_o)
    u, r0, r1 = make_rk_rand(o, rng)
    evk_o = o.rekeygen(s1, pk2_o, u, r0, r1)
    d_evk = g.empty((g.beta, 2, D, N))
    g.rekeygen(g.to_device(s1), g.to_device(pk2_o), g.to_device(u), g.to_device(r0), g.to_device(r1), d_evk)
    assert np.array_equal(d_evk.to_host(), evk_o)
    # encrypt at full level and at a reduced level (first nl limbs of the QP public key)
    for nl in (L, max(2, L - 2)):
        B = 2
        vals = rng.uniform(-0.3, 0.3, size=(B, N // 2))
        scale = o.sf_big(0) if nl == L else 2.0 ** 30
        pts = np.stack([o.encode(vals[b], scale, nl) for b in range(B)])
        v = np.stack([sample_ternary(rng, N) for _ in range(B)])
        f0 = np.stack([sample_gauss(rng, N) for _ in range(B)])
        f1 = np.stack([sample_gauss(rng, N) for _ in range(B)])
        d_ct = g.empty((B, 2, nl, N))
        g.encrypt(d_pk, g.to_device(pts), g.to_device(v), g.to_device(f0), g.to_device(f1), d_ct, B, nl)
        got = d_ct.to_host()
        for b in range(B):
            assert np.array_equal(got[b], o.encrypt(pk1_o, pts[b], v[b], f0[b], f1[b]))
        d_m = g.empty((B, nl, N))
        g.decrypt(d_ct, d_sk, d_m, B, nl)
        m = d_m.to_host()
        for b in range(B):
            assert np.array_equal(m[b], o.decrypt_core(got[b], sk1_o))


@pytest.mark.parametrize("name", ["tiny", "ref", "c3"])
def test_lift_ntt_matches_oracle_encode(ctxs, name):
    # integer half of Encode: rounded scaled coefficients -> residues -> NTT.  The oracle's encode does
    # fp64 embedding + the same rounding, so feeding its coefficient doubles must give identical residues.
    g, o = ctxs(name)
    rng = np.random.default_rng(15)
    N = g.N
    coef = np.rint(rng.normal(0, 2.0 ** 45, size=(2, N)))
    coef[0, :4] = [2.0 ** 70 + 2.0 ** 30, -(2.0 ** 69), 0.5, -0.5]  # beyond int64; ties away from zero
    nl = g.L
    d_out = g.empty((2, nl, N))
    g.lift_ntt(g.to_device(coef), d_out, 2, nl)
    got = d_out.to_host()
    for b in range(2):
        ints = [int(np.sign(x) * np.floor(abs(x) + 0.5)) for x in coef[b]]
        for l in (0, 1, nl - 1):
            q = int(g.moduli[l])
            res = np.array([v % q for v in ints], dtype=np.uint64)
            assert np.array_equal(got[b, l], o.ntt_fwd(l, res))


@pytest.mark.parametrize("name", ["tiny", "ref", "c3"])
def test_encode_decode_on_device(ctxs, name):
    """Encode / Decode (fp64 canonical embedding + CRT interpolation on the GPU) vs the oracle's host versions.
    Floating point: coefficients are ~scale * 2^-4 with a 53-bit mantissa, so two correctly rounded FFTs may differ by
    scale * 2^-50 in a coefficient (a few units at scale 2^60); decoded values must agree far below the scheme's
    precision (tolerance 2^-40 relative to the value range)."""
    g, o = ctxs(name)
    rng = np.random.default_rng(22)
    N, L, slots = g.N, g.L, g.N // 2
    B = 2
    vals = rng.uniform(-0.3, 0.3, size=(B, slots))
    vals[1, 1:] = 0.0  # the {mean} plaintext shape: one value, zero padding
    scale = o.sf_big(0)
    d_pt = g.empty((B, L, N))
    g.encode(g.to_device(vals), d_pt, B, L, scale)
    pt = d_pt.to_host()
    for b in range(B):
        ref = o.encode(vals[b], scale, L)
        for l in (0, 1, L - 1):
            q = int(g.moduli[l])
            d = (o.ntt_inv(l, pt[b, l]).astype(object) - o.ntt_inv(l, ref[l]).astype(object)) % q
            d = np.array([int(x) if x <= q // 2 else int(x) - q for x in d])
            assert np.abs(d).max() <= scale * 2.0 ** -50 + 1, (name, b, l, np.abs(d).max())
    # decode: CRT + embedding of the exact plaintext polynomial (coefficient form of the oracle's encoding)
    nl = max(2, L - 1)
    m = np.stack([np.stack([o.ntt_inv(l, o.encode(vals[b], 2.0 ** 45, nl)[l]) for l in range(nl)]) for b in range(B)])
    d_vals = g.empty((B, slots), dtype=np.float64)
    g.decode(g.to_device(m), d_vals, B, nl, 2.0 ** 45)
    got = d_vals.to_host()
    assert np.abs(got - vals).max() < 2.0 ** -35  # encoding rounding at scale 2^45 (~sqrt(N) * 2^-46), not a GPU error
    # round trip entirely on the device at the real scale, through decrypt of a trivial encryption (c1 = 0)
    ct = np.zeros((B, 2, L, N), dtype=np.uint64)
    ct[:, 0] = pt
    d_m = g.empty((B, L, N))
    g.decrypt(g.to_device(ct), g.to_device(np.zeros((g.D, N), dtype=np.uint64)), d_m, B, L)
    g.decode(d_m, d_vals, B, L, scale)
    assert np.abs(d_vals.to_host() - vals).max() < 2.0 ** -40
    # agreement with the oracle's decoder on a noisy decrypted polynomial
    s_t = sample_ternary(rng, N)
    pk, sk = o.keygen(s_t, sample_uniform(rng, o.moduli, N), sample_gauss(rng, N))
    c = o.encrypt(pk, o.encode(vals[0], scale, L), sample_ternary(rng, N), sample_gauss(rng, N), sample_gauss(rng, N))
    g.decode(g.to_device(o.decrypt_core(c, sk)[None]), d_vals.view(0, (1, slots)), 1, L, scale)
    assert np.abs(d_vals.to_host()[0] - o.decrypt_decode(c, sk, scale)).max() < 2.0 ** -40


@pytest.mark.parametrize("name,nl", [("tiny", 5), ("tiny", 4), ("tiny", 3), ("tiny", 1), ("c1", 3), ("c1", 2),
                                     ("ref", 4), ("ref", 3), ("c3", 12), ("c3", 11), ("c3", 5),
                                     ("c5s", 20), ("c5s", 15), ("c5s", 8), ("n17", 4), ("n11", 4), ("n11", 3)])
def test_modup_moddown_reencrypt(ctxs, name, nl):
    g, o = ctxs(name)
    rng = np.random.default_rng(16)
    N, K, D = g.N, g.K, g.D
    B = 3 if N <= 1 << 14 else 2
    ext = nl + K
    ct = rand_ct(rng, g, nl, B)
    evk = rand_polys(rng, g, list(range(D)) * (2 * g.beta), 1).reshape(g.beta, 2, D, N)
    nparts = g.num_parts(nl)
    # ModUp digits
    c1 = np.ascontiguousarray(ct[:, 1])
    d_dig = g.empty((B, nparts, ext, N))
    g.modup(g.to_device(c1), d_dig, B, nl)
    dig = d_dig.to_host()
    for b in range(B):
        assert np.array_equal(dig[b], o.modup_digits(c1[b])), (name, nl, b)
    # ModDown
    ids = list(range(nl)) + list(range(g.L, D))
    x = rand_polys(rng, g, ids, B)
    d_md = g.empty((B, nl, N))
    g.moddown(g.to_device(x), d_md, B, nl)
    md = d_md.to_host()
    for b in range(B):
        assert np.array_equal(md[b], o.moddown(x[b]))
    # full ReEncrypt, out of place and in place
    d_ct, d_evk, d_out = g.to_device(ct), g.to_device(evk), g.empty((B, 2, nl, N))
    g.reencrypt(d_ct, d_evk, d_out, B, nl)
    got = d_out.to_host()
    exp = [o.reencrypt(ct[b], evk) for b in range(B)]
    for b in range(B):
        assert np.array_equal(got[b], exp[b]), (name, nl, b)
    g.reencrypt(d_ct, d_evk, d_ct, B, nl)
    assert np.array_equal(d_ct.to_host(), got)


@pytest.mark.parametrize("name,nl", [("tiny", 5), ("ref", 4), ("c3", 12)])
def test_reencrypt_accumulate(ctxs, name, nl):
    # ReEncrypt folded into a running aggregate == EvalAdd of the individually re-encrypted ciphertexts
    g, o = ctxs(name)
    rng = np.random.default_rng(19)
    B, C = 2, 3
    cts = np.stack([rand_ct(rng, g, nl, B) for _ in range(C)])
    evks = [rand_polys(rng, g, list(range(g.D)) * (2 * g.beta), 1).reshape(g.beta, 2, g.D, g.N) for _ in range(C)]
    d_acc = g.empty((B, 2, nl, g.N))
    for c in range(C):
        d_ct, d_evk = g.to_device(cts[c]), g.to_device(evks[c])
        if c == 0:
            g.reencrypt(d_ct, d_evk, d_acc, B, nl)
        else:
            g.reencrypt_accumulate(d_ct, d_evk, d_acc, B, nl)
    got = d_acc.to_host()
    for b in range(B):
        acc = o.reencrypt(cts[0, b], evks[0])
        for c in range(1, C):
            acc = o.eval_add(acc, o.reencrypt(cts[c, b], evks[c]))
        assert np.array_equal(got[b], acc)
    from ppqsflhe_amd import MkckksError
    with pytest.raises(MkckksError):
        g.reencrypt_accumulate(d_acc, g.to_device(evks[0]), d_acc, B, nl)  # aliasing is refused


@pytest.mark.parametrize("name,nl,C,B", [("tiny", 5, 3, 2), ("tiny", 3, 1, 1), ("ref", 4, 2, 3), ("c3", 12, 3, 2),
                                         ("c5s", 20, 2, 1), ("n11", 4, 3, 2), ("tiny", 5, 2, 19),
                                         # the N=2^16 sum kernel takes clients two at a time: even / odd counts, one
                                         # and two full pairs, a lower level (nl < L)
                                         ("c3", 12, 4, 1), ("c3", 12, 5, 1), ("c3", 11, 2, 2), ("c3", 12, 1, 1),
                                         # N = 2^17: 512-point rows, three-round fused sum / inner-product kernels
                                         ("n17", 4, 3, 2), ("n17", 3, 2, 1),
                                         # merged n-client flow: more clients than one group (8 + 1: the running sum
                                         # continues from `out`), more indices than one workspace chunk (16 + 3), one
                                         # client, a level with a single digit and with a partial last digit
                                         ("ref", 4, 9, 2), ("ref", 4, 3, 19), ("ref", 2, 2, 2), ("ref", 3, 1, 1),
                                         ("c3", 9, 2, 1), ("c3", 4, 3, 1), ("c3", 2, 2, 1)])
def test_reencrypt_sum(ctxs, name, nl, C, B):
    # sum over clients of ReEncrypt(ct_c, evk_c) in one call == EvalAdd chain of the individual re-encryptions
    g, o = ctxs(name)
    rng = np.random.default_rng(23)
    cts = np.stack([rand_ct(rng, g, nl, B) for _ in range(C)])
    evks = np.stack([rand_polys(rng, g, list(range(g.D)) * (2 * g.beta), 1).reshape(g.beta, 2, g.D, g.N)
                     for _ in range(C)])
    d_out = g.empty((B, 2, nl, g.N))
    g.reencrypt_sum(g.to_device(cts), g.to_device(evks), d_out, C, B, nl)
    got = d_out.to_host()
    for b in range(B if B < 4 else 3):
        acc = o.reencrypt(cts[0, b], evks[0])
        for c in range(1, C):
            acc = o.eval_add(acc, o.reencrypt(cts[c, b], evks[c]))
        assert np.array_equal(got[b], acc), (name, nl, b)
    if B >= 4:  # spot-check the last chunk too
        b = B - 1
        acc = o.reencrypt(cts[0, b], evks[0])
        for c in range(1, C):
            acc = o.eval_add(acc, o.reencrypt(cts[c, b], evks[c]))
        assert np.array_equal(got[b], acc)


@pytest.mark.parametrize("env", [{"MKCKKS_QSUM_GROUP": "2"},      # merged flow, clients in groups of 2 (running sum in out)
                                 {"MKCKKS_QSUM_GROUP": "1"},
                                 {"MKCKKS_CHUNK": "2"},           # workspace chunks of 2 ciphertext indices
                                 {"MKCKKS_CU_AFFINE": "0"},       # plain XCD-aware placement of the workgroups that share tiles
                                 {"MKCKKS_NO_PM": "1"},           # Shoup butterflies on q_0 and the P limbs (any 60-bit modulus)
                                 {"MKCKKS_NO_FP64": "1"},         # integer arithmetic on every limb: no merged flow, per-client loop
                                 {"MKCKKS_NO_FP64": "1", "MKCKKS_NO_PM": "1"},
                                 {"MKCKKS_GENERIC_NTT": "1"}])    # LDS-stage kernels for both passes, nothing fused
def test_unfused_kernel_paths_stay_bit_exact(ctxs, monkeypatch, env):
    """The switches the library keeps select tuning parameters (chunk, client group, placement) or the arithmetic /
    kernel families that other moduli and ring sizes fall back to (Shoup instead of pseudo-Mersenne, integer instead of
    fp64, LDS-stage transforms): a context created under the switch must give the same bits as the default context and
    as the oracle at N = 2^16.  Switches are read once, when a context is created."""
    from ppqsflhe_amd import Context
    g, o = ctxs("c3")
    nl, C, B = 12, 3, 1
    rng = np.random.default_rng(77)
    cts = np.stack([rand_ct(rng, g, nl, B) for _ in range(C)])
    evks = np.stack([rand_polys(rng, g, list(range(g.D)) * (2 * g.beta), 1).reshape(g.beta, 2, g.D, g.N)
                     for _ in range(C)])
    d_ref = g.empty((B, 2, nl, g.N))
    g.reencrypt_sum(g.to_device(cts), g.to_device(evks), d_ref, C, B, nl)
    want = d_ref.to_host()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    a = CONFIGS["c3"]
    g2 = Context(a[0], a[1], a[2], a[3], dnum=a[4], device=0)
    try:
        d_cts, d_evks = g2.to_device(cts), g2.to_device(evks)
        d_out = g2.empty((B, 2, nl, g.N))
        g2.reencrypt_sum(d_cts, d_evks, d_out, C, B, nl)
        assert np.array_equal(d_out.to_host(), want)
        d_one = g2.empty((B, 2, nl, g.N))
        g2.reencrypt(g2.to_device(cts[0]), g2.to_device(evks[0]), d_one, B, nl)
        assert np.array_equal(d_one.to_host()[0], o.reencrypt(cts[0, 0], evks[0]))
    finally:
        g2.close()


def test_fused_path_across_workspace_chunks(ctxs, monkeypatch):
    """N = 2^16 with a workspace chunk of 2 ciphertexts: 3 ciphertexts per client take one full and one partial chunk
    through the fused kernels (the merged n-client flow of reencrypt_sum, and the single-client reencrypt)."""
    from ppqsflhe_amd import Context
    _, o = ctxs("c3")
    a = CONFIGS["c3"]
    monkeypatch.setenv("MKCKKS_CHUNK", "2")
    g = Context(a[0], a[1], a[2], a[3], dnum=a[4], device=0)
    try:
        nl, C, B = 12, 2, 3
        rng = np.random.default_rng(5)
        cts = np.stack([rand_ct(rng, g, nl, B) for _ in range(C)])
        evks = np.stack([rand_polys(rng, g, list(range(g.D)) * (2 * g.beta), 1).reshape(g.beta, 2, g.D, g.N)
                         for _ in range(C)])
        d_out = g.empty((B, 2, nl, g.N))
        g.reencrypt_sum(g.to_device(cts), g.to_device(evks), d_out, C, B, nl)
        got = d_out.to_host()
        d_one = g.empty((B, 2, nl, g.N))
        g.reencrypt(g.to_device(cts[1]), g.to_device(evks[1]), d_one, B, nl)
        one = d_one.to_host()
        for b in (0, 2):  # first ciphertext of the full chunk, the lone ciphertext of the partial one
            r0, r1 = o.reencrypt(cts[0, b], evks[0]), o.reencrypt(cts[1, b], evks[1])
            assert np.array_equal(one[b], r1)
            assert np.array_equal(got[b], o.eval_add(r0, r1))
    finally:
        g.close()


def test_fused_kernels_are_deterministic(ctxs):
    """The row kernels synchronise LDS hand-offs at wave level only (no workgroup barrier): 25 repetitions of the fused
    N = 2^16 path on the same inputs, with other work in flight on the device, must give identical bits every time."""
    g, _ = ctxs("c3")
    nl, C, B = 12, 4, 4
    rng = np.random.default_rng(99)
    cts = np.stack([rand_ct(rng, g, nl, B) for _ in range(C)])
    evks = np.stack([rand_polys(rng, g, list(range(g.D)) * (2 * g.beta), 1).reshape(g.beta, 2, g.D, g.N)
                     for _ in range(C)])
    d_cts, d_evks = g.to_device(cts), g.to_device(evks)
    d_out = g.empty((B, 2, nl, g.N))
    g.reencrypt_sum(d_cts, d_evks, d_out, C, B, nl)
    want = d_out.to_host()
    for _ in range(25):
        g.reencrypt_sum(d_cts, d_evks, d_out, C, B, nl)
        assert np.array_equal(d_out.to_host(), want)


def test_device_samplers(ctxs):
    """Philox samplers in HBM: distributional checks (OpenFHE's PRNG stream is not reproducible), determinism per
    (seed, stream), independence across streams, exact range of the uniform limbs."""
    g, _ = ctxs("ref")
    n = 1 << 18
    d_t = g.empty((n,), dtype=np.int8)
    g.sample_ternary(d_t, n, 1234, 0)
    t = d_t.to_host()
    assert set(np.unique(t)) == {-1, 0, 1}
    counts = np.bincount(t + 1, minlength=3)
    assert np.abs(counts - n / 3).max() < 5 * np.sqrt(n * 2 / 9)  # 5 sigma of a multinomial cell
    g.sample_ternary(d_t, n, 1234, 0)
    assert np.array_equal(d_t.to_host(), t)  # deterministic
    g.sample_ternary(d_t, n, 1234, 1)
    assert not np.array_equal(d_t.to_host(), t)  # another stream
    assert abs(np.corrcoef(d_t.to_host().astype(float), t.astype(float))[0, 1]) < 0.02
    d_g = g.empty((n,), dtype=np.int32)
    g.sample_gauss(d_g, n, 3.19, 99, 0)
    e = d_g.to_host().astype(np.float64)
    assert abs(e.mean()) < 5 * 3.19 / np.sqrt(n)
    assert abs(e.var() - 3.19 ** 2) < 0.15  # discrete Gaussian: variance = sigma^2 up to ~1e-8
    assert np.abs(e).max() <= 39
    # P(0) of D_{Z,sigma}
    S = sum(np.exp(-k * k / (2 * 3.19 ** 2)) for k in range(-60, 61))
    assert abs((e == 0).mean() - 1 / S) < 5 * np.sqrt((1 / S) / n)
    d_u = g.empty((2, g.D, g.N))
    g.sample_uniform(d_u, 2, g.L, True, 7, 3)
    u = d_u.to_host()
    for i in range(g.D):
        q = int(g.moduli[i])
        assert u[:, i].max() < q
        assert abs(u[:, i].astype(np.float64).mean() / q - 0.5) < 5 * np.sqrt(1 / 12 / (2 * g.N))
    assert not np.array_equal(u[0], u[1])


def test_empty_batches_are_noops(ctxs):
    g, _ = ctxs("tiny")
    d = g.empty((1, 2, g.L, g.N))
    evk = g.empty((g.beta, 2, g.D, g.N))
    before = d.upload(np.ones((1, 2, g.L, g.N), dtype=np.uint64)).to_host()
    g.reencrypt(d, evk, d, 0, g.L)
    g.eval_add(d, d, d, 0, g.L)
    g.eval_sum(d, d, 3, 0, g.L)
    g.ntt_forward(d, 0, g.L)
    g.sync()
    assert np.array_equal(d.to_host(), before)


def test_reencrypt_batch_larger_than_chunk(ctxs):
    g, o = ctxs("tiny")
    rng = np.random.default_rng(17)
    B, nl = 37, g.L
    ct = rand_ct(rng, g, nl, B)
    evk = rand_polys(rng, g, list(range(g.D)) * (2 * g.beta), 1).reshape(g.beta, 2, g.D, g.N)
    d_out = g.empty((B, 2, nl, g.N))
    g.reencrypt(g.to_device(ct), g.to_device(evk), d_out, B, nl)
    got = d_out.to_host()
    for b in range(B):
        assert np.array_equal(got[b], o.reencrypt(ct[b], evk))


def test_full_round_known_answer(ctxs, golden_dir):
    """The reference's round (run.sh:28-44) on the GPU at the reference's own parameters, checked against the
    reference's plaintext-in / decrypted-out fixture (P8) and bit-exactly against the oracle."""
    import os
    g, o = ctxs("ref")
    W = np.load(os.path.join(golden_dir, "e2e_weights.npz"))
    rng = np.random.default_rng(18)
    N, D, L = g.N, g.D, g.L
    keys = []
    for _ in range(2):
        s, a, e = make_keys(o, rng)
        pk, sk = o.keygen(s, a, e)
        keys.append((s, pk, sk))
    u, r0, r1 = make_rk_rand(o, rng)
    rk12 = o.rekeygen(keys[0][0], keys[1][1], u, r0, r1)
    u, r0, r1 = make_rk_rand(o, rng)
    rk21 = o.rekeygen(keys[1][0], keys[0][1], u, r0, r1)
    v1, v2 = W["sample_c1_param_1_values"], W["sample_c2_param_1_values"]

    def enc(pk, vals):
        pt = o.encode(vals, o.sf_big(0), L)
        return o.encrypt(pk, pt, sample_ternary(rng, N), sample_gauss(rng, N), sample_gauss(rng, N))

    ct1, ct2 = enc(keys[0][1], v1), enc(keys[1][1], v2)
    d1, d2 = g.to_device(ct1[None]), g.to_device(ct2[None])
    d12 = g.empty((1, 2, L, N))
    g.reencrypt(d1, g.to_device(rk12), d12, 1, L)          # changeCipherDomain c1 -> c2
    d_sum = g.empty((1, 2, L, N))
    g.eval_add(d12, d2, d_sum, 1, L)                        # EvalAdd
    d_avg = g.empty((1, 2, L - 1, N))
    g.rescale_mult_const(d_sum, d_avg, 1, L, 0.5)           # EvalMult(., 0.5)
    d_back = g.empty((1, 2, L - 1, N))
    g.reencrypt(d_avg, g.to_device(rk21), d_back, 1, L - 1)  # changeCipherDomain c2 -> c1
    # oracle, same inputs
    s_o = o.eval_add(o.reencrypt(ct1, rk12), ct2)
    avg_o = o.mult_factors(o.rescale(s_o), o.const_factors(L - 1, 1, 0.5))
    back_o = o.reencrypt(avg_o, rk21)
    assert np.array_equal(d_avg.to_host()[0], avg_o)
    assert np.array_equal(d_back.to_host()[0], back_o)
    # decrypt on the GPU, decode with the oracle's decoder, compare with the fixture
    d_m = g.empty((1, L - 1, N))
    g.decrypt(d_back, g.to_device(keys[0][2]), d_m, 1, L - 1)
    assert np.array_equal(d_m.to_host()[0], o.decrypt_core(back_o, keys[0][2]))
    dec = o.decrypt_decode(back_o, keys[0][2], o.sf(1) ** 2)[: v1.size]
    mean = (v1 + v2) / 2
    assert np.abs(dec - mean).max() < 2.0 ** -25
    assert np.abs(dec - W["decrypted_c1_param_1_values"]).max() < 2.0 ** -24


def test_errors_are_loud(ctxs):
    from ppqsflhe_amd import MkckksError
    g, _ = ctxs("tiny")
    d = g.empty((1, 2, g.L, g.N))
    with pytest.raises(MkckksError):
        g.eval_add(d, d, d, 1, g.L + 1)
    with pytest.raises(MkckksError):
        g.rescale(d, d, 1, 1)
    with pytest.raises(MkckksError):
        g.reduce_mod(d, 1, g.L, 9)


def test_config5_full_round_n17_l20():
    """BASELINE configs[4] shape: N=2^17, L=20 limbs, dnum=3 (alpha=7, K=7), ~1e5 weights packed into 2 ciphertexts
    per client, 2 clients: encode+encrypt -> PRE -> aggregate -> PRE back -> decrypt+decode, all on the GPU; key
    switching bit-exact vs the oracle on one ciphertext, decoded result within 2^-25 of the plaintext mean."""
    from ppqsflhe_amd import Context
    g = Context(17, 18, 50, 60, dnum=3, device=0)
    o = OracleContext(17, 18, 50, 60, dnum=3)
    assert (g.L, g.K, g.alpha, g.beta) == (20, 7, 7, 3)
    rng = np.random.default_rng(24)
    N, L, D, slots = g.N, g.L, g.D, g.N // 2
    n_w = 100_000
    B = -(-n_w // slots)  # 2 ciphertexts of 65536 slots
    w = [rng.uniform(-0.3, 0.3, size=n_w) for _ in range(2)]

    def device_keys(seed):
        d_s, d_e, d_a = g.empty((N,), np.int8), g.empty((N,), np.int32), g.empty((D, N))
        g.sample_ternary(d_s, N, seed, 0)
        g.sample_gauss(d_e, N, 3.19, seed, 1)
        g.sample_uniform(d_a, 1, L, True, seed, 2)
        d_pk, d_sk = g.empty((2, D, N)), g.empty((D, N))
        g.keygen(d_s, d_a, d_e, d_pk, d_sk)
        return d_s, d_pk, d_sk

    k1, k2 = device_keys(1), device_keys(2)

    def device_rekey(sk_t, pk_new, seed):
        d_u, d_e0, d_e1 = g.empty((g.beta, N), np.int8), g.empty((g.beta, N), np.int32), g.empty((g.beta, N), np.int32)
        g.sample_ternary(d_u, g.beta * N, seed, 0)
        g.sample_gauss(d_e0, g.beta * N, 3.19, seed, 1)
        g.sample_gauss(d_e1, g.beta * N, 3.19, seed, 2)
        d_evk = g.empty((g.beta, 2, D, N))
        g.rekeygen(sk_t, pk_new, d_u, d_e0, d_e1, d_evk)
        return d_evk

    rk12, rk21 = device_rekey(k1[0], k2[1], 11), device_rekey(k2[0], k1[1], 12)
    scale = g.sf_big(0)

    def device_encrypt(pk, vals, seed):
        padded = np.zeros((B, slots))
        padded.reshape(-1)[:n_w] = vals
        d_pt, d_ct = g.empty((B, L, N)), g.empty((B, 2, L, N))
        g.encode(g.to_device(padded), d_pt, B, L, scale)
        d_v, d_e0, d_e1 = g.empty((B, N), np.int8), g.empty((B, N), np.int32), g.empty((B, N), np.int32)
        g.sample_ternary(d_v, B * N, seed, 0)
        g.sample_gauss(d_e0, B * N, 3.19, seed, 1)
        g.sample_gauss(d_e1, B * N, 3.19, seed, 2)
        g.encrypt(pk, d_pt, d_v, d_e0, d_e1, d_ct, B, L)
        return d_ct

    ct1, ct2 = device_encrypt(k1[1], w[0], 21), device_encrypt(k2[1], w[1], 22)
    d12 = g.empty((B, 2, L, N))
    g.reencrypt(ct1, rk12, d12, B, L)
    # bit-exact key switch vs the oracle on the first ciphertext at full size
    assert np.array_equal(d12.to_host()[0], o.reencrypt(ct1.to_host()[0], rk12.to_host()))
    d_sum, d_avg = g.empty((B, 2, L, N)), g.empty((B, 2, L - 1, N))
    g.eval_add(d12, ct2, d_sum, B, L)
    g.rescale_mult_const(d_sum, d_avg, B, L, 0.5)
    d_back = g.empty((B, 2, L - 1, N))
    g.reencrypt(d_avg, rk21, d_back, B, L - 1)
    out_scale = g.sf(1) ** 2
    mean = (w[0] + w[1]) / 2
    for d_ct, sk in ((d_avg, k2[2]), (d_back, k1[2])):
        d_m, d_vals = g.empty((B, L - 1, N)), g.empty((B, slots), np.float64)
        g.decrypt(d_ct, sk, d_m, B, L - 1)
        g.decode(d_m, d_vals, B, L - 1, out_scale)
        got = d_vals.to_host().reshape(-1)[:n_w]
        assert np.abs(got - mean).max() < 2.0 ** -25
    g.close()


def test_full_size_batch_properties(ctxs):
    """BASELINE configs[1..3] at full batch size (8 clients x 16 ciphertexts, N=2^16, L=12), checked through
    size-independent properties instead of the (slow) oracle: (a) an element of a large batch equals the same
    ciphertext re-encrypted alone (batches only amortise, they never mix items); (b) the fused n-client kernel path
    equals re-encrypt-each-then-add (different kernels, same bits); (c) the accumulate form equals both; (d) the
    u64-sum + reduce_mod collective arithmetic equals eval_sum."""
    import torch
    g, _ = ctxs("c3")
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(77)
    C, B, L, N, D = 8, 16, g.L, g.N, g.D

    def uniform(lead, ids):
        t = torch.empty(*lead, len(ids), N, dtype=torch.int64, device=dev)
        for j, l in enumerate(ids):
            t[..., j, :] = torch.randint(0, int(g.moduli[l]), (*lead, N), generator=gen, device=dev, dtype=torch.int64)
        return t

    cts = uniform((C, B), list(range(L)) * 2).view(C, B, 2, L, N)
    evks = uniform((C,), list(range(D)) * (2 * g.beta)).view(C, g.beta, 2, D, N)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    each = torch.empty_like(cts)
    for c in range(C):
        g.reencrypt(cts[c], evks[c], each[c], B, L)
    # (a) singles vs batch
    single = torch.empty(1, 2, L, N, dtype=torch.int64, device=dev)
    for c, b in ((0, 0), (3, 7), (7, 15)):
        g.reencrypt(cts[c, b:b + 1], evks[c], single, 1, L)
        assert torch.equal(single[0], each[c, b])
    # (b) fused sum vs add of the individual results
    ref = torch.empty(B, 2, L, N, dtype=torch.int64, device=dev)
    g.eval_sum(each, ref, C, B, L)
    fused = torch.empty_like(ref)
    g.reencrypt_sum(cts, evks, fused, C, B, L)
    assert torch.equal(fused, ref)
    # (c) accumulate form
    acc = torch.empty_like(ref)
    g.reencrypt(cts[0], evks[0], acc, B, L)
    for c in range(1, C):
        g.reencrypt_accumulate(cts[c], evks[c], acc, B, L)
    assert torch.equal(acc, ref)
    # (d) integer-sum collective arithmetic: wrap-around int64 sum of 8 canonical terms, then reduce_mod
    raw = each.sum(dim=0)  # int64 wrap == uint64 add
    g.reduce_mod(raw, B, L, C)
    assert torch.equal(raw, ref)
    torch.cuda.synchronize()
    g.set_stream(None)


def test_config4_sixty_four_clients_on_one_rank(ctxs):
    """BASELINE configs[3]'s workload (64 clients, N=2^16, L=12, dnum=3) as far as ONE GPU goes: the server loop of
    server/src/aggregateEncryptedWeights.cpp:68-115 generalised to 64 clients -- changeCipherDomain x 64
    (orchestration/server_fns.sh:62-80), the EvalAdd chain, EvalMult(1/64).  reencrypt_sum runs 8 groups of 8 clients
    with the running sum carried through `out`; index 0 is checked against the oracle's ReEncrypt x 64 + EvalAdd chain +
    rescale * const, index 1 against re-encrypt-each-then-eval_sum on the device (different kernels, same bits).  The
    sharded leg (8 ranks x 8 clients, RCCL reduce-scatter) is covered on CPU by tests/test_sharding_gloo.py at world
    size 8; only its 8-GPU hardware run is left to the driver."""
    import torch
    g, o = ctxs("c3")
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(64)
    C, B, L, N, D = 64, 2, g.L, g.N, g.D

    def uniform(lead, ids):
        t = torch.empty(*lead, len(ids), N, dtype=torch.int64, device=dev)
        for j, l in enumerate(ids):
            t[..., j, :] = torch.randint(0, int(g.moduli[l]), (*lead, N), generator=gen, device=dev, dtype=torch.int64)
        return t

    cts = uniform((C, B), list(range(L)) * 2).view(C, B, 2, L, N)
    evks = uniform((C,), list(range(D)) * (2 * g.beta)).view(C, g.beta, 2, D, N)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    agg = torch.empty(B, 2, L, N, dtype=torch.int64, device=dev)
    g.reencrypt_sum(cts, evks, agg, C, B, L)
    avg = torch.empty(B, 2, L - 1, N, dtype=torch.int64, device=dev)
    g.rescale_mult_const(agg, avg, B, L, 1.0 / C)
    # index 1: every client re-encrypted on its own, n-ary EvalAdd
    each = torch.empty(C, 1, 2, L, N, dtype=torch.int64, device=dev)
    for c in range(C):
        g.reencrypt(cts[c, 1:2], evks[c], each[c], 1, L)
    ref1 = torch.empty(1, 2, L, N, dtype=torch.int64, device=dev)
    g.eval_sum(each, ref1, C, 1, L)
    assert torch.equal(agg[1], ref1[0])
    torch.cuda.synchronize()
    g.set_stream(None)
    # index 0: the oracle's chain
    h_ct = cts[:, 0].cpu().numpy().view(np.uint64)
    acc = None
    for c in range(C):
        r = o.reencrypt(h_ct[c], evks[c].cpu().numpy().view(np.uint64))
        acc = r if acc is None else o.eval_add(acc, r)
    assert np.array_equal(agg[0].cpu().numpy().view(np.uint64), acc)
    exp = o.mult_factors(o.rescale(acc), o.const_factors(L - 1, 1, 1.0 / C))
    assert np.array_equal(avg[0].cpu().numpy().view(np.uint64), exp)


def test_async_copies_pinned_buffers_and_range_check(ctxs):
    """The I/O pipeline entry points of the C-ABI (include/mkckks.h: host_alloc, upload_async / download_async with
    tickets, fences, count_noncanonical): ciphertexts travel pinned buffer -> HBM on the upload stream, are re-encrypted on
    the compute stream behind fence_uploads, and come back on the download stream behind fence_compute -- the same bits as
    the synchronous path.  A residue at or above its modulus is counted on the device."""
    from ppqsflhe_amd import MkckksError
    g, o = ctxs("ref")
    rng = np.random.default_rng(31)
    nl, B = g.L, 6
    cts = rand_ct(rng, g, nl, B)
    evk = rand_polys(rng, g, list(range(g.D)) * (2 * g.beta), 1).reshape(g.beta, 2, g.D, g.N)
    d_evk = g.to_device(evk)
    d_in, d_out = g.empty((B, 2, nl, g.N)), g.empty((B, 2, nl, g.N))
    one = cts[0].nbytes
    pin = g.host_alloc(2 * one)          # two slots: the classic double buffer
    back = g.host_alloc(B * one)
    tickets = []
    for b in range(B):
        slot = pin[(b % 2) * one:(b % 2 + 1) * one]
        if b >= 2:
            g.copy_wait(tickets[b - 2])   # the slot's previous upload is over before it is refilled
        slot[:] = np.frombuffer(cts[b].tobytes(), dtype=np.uint8)
        tickets.append(g.upload_async(d_in.ptr + b * one, slot))
    assert tickets == sorted(tickets) and tickets[0] >= 1
    g.fence_uploads()
    assert g.count_noncanonical(d_in, B, nl) == 0
    g.reencrypt(d_in, d_evk, d_out, B, nl)
    g.fence_compute()
    t = g.download_async(back, d_out)
    g.copy_wait(t)
    assert g.copy_done(t) and all(g.copy_done(x) for x in tickets)
    got = np.frombuffer(back.tobytes(), dtype=np.uint64).reshape(B, 2, nl, g.N)
    for b in (0, 3, 5):
        assert np.array_equal(got[b], o.reencrypt(cts[b], evk))
    # two residues out of range -> counted, nothing else
    bad = cts.copy()
    bad[1, 0, 0, 5] = g.moduli[0]
    bad[4, 1, nl - 1, 9] = np.uint64(2**64 - 1)
    assert g.count_noncanonical(g.to_device(bad), B, nl) == 2
    with pytest.raises(MkckksError):
        g.copy_wait(10**9)               # a ticket that was never issued
    g.host_free(pin)
    g.host_free(back)


def test_rccl_reduce_scatter_through_the_cabi(ctxs):
    """mkckks_reduce_scatter_sum_mod on a ONE-rank communicator: librccl loads beside the library (the copy the process
    already carries, when torch is in it), ncclCommInitRank / ncclReduceScatter(ncclUint64, ncclSum) execute on the
    device, and the word-wise reduction follows on the same stream.  With one rank the exchange is the identity, so the
    result must equal reduce_mod of the input: here the input is the unreduced sum of 5 canonical terms, i.e. what 5
    ranks' partial sums would add up to."""
    g, _ = ctxs("ref")
    rng = np.random.default_rng(31)
    B, nl, terms = 3, g.L, 5
    cts = np.stack([rand_ct(rng, g, nl, B) for _ in range(terms)])
    raw = cts.sum(axis=0, dtype=np.uint64)
    d_raw, d_want, d_got = g.to_device(raw), g.to_device(raw), g.empty((B, 2, nl, g.N))
    g.reduce_mod(d_want, B, nl, terms)
    uid = g.comm_unique_id()
    comm = g.comm_create(uid, 1, 0)
    try:
        assert "rccl" in g.comm_library()
        g.reduce_scatter_sum_mod(comm, d_raw, d_got, B, nl, terms)
        assert np.array_equal(d_got.to_host(), d_want.to_host())
        # in place on the rank's own block (recv == send + rank * count), as bench.py uses it
        g.reduce_scatter_sum_mod(comm, d_raw, d_raw, B, nl, terms)
        assert np.array_equal(d_raw.to_host(), d_want.to_host())
    finally:
        g.comm_destroy(comm)
    from ppqsflhe_amd import MkckksError
    with pytest.raises(MkckksError):
        g.reduce_scatter_sum_mod(None, d_raw, d_got, B, nl, 2)  # no communicator


def _run_bench(extra_args, env_extra, world):
    """bench.py as a FRESH child process group: world ranks on device 0, collective carried by gloo (RCCL refuses two
    ranks on one GPU), rendezvous on 127.0.0.1."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MKCKKS_BENCH_BACKEND="gloo", MKCKKS_BENCH_ONE_DEVICE="1", **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
           "--gpus", str(world), "--steps", "2", "--warmup", "1", "--no-cpu", "--clients", "2", "--cts", "4"] + extra_args
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    return r


@pytest.mark.parametrize("serial", [False, True])
def test_bench_two_ranks_exchange_verifies(serial):
    """bench.py's N>1 code path (BASELINE configs[3] at two ranks): per-rank reencrypt_sum, integer reduce-scatter of the
    partial sums, reduce_mod, rescale of the rank's shard -- pipelined with the next batch (default) and serial --
    checked by --verify against the modular sum of the all-gathered per-rank aggregates, bit for bit."""
    import json
    r = _run_bench(["--verify"], {"MKCKKS_BENCH_SERIAL_EXCHANGE": "1"} if serial else {}, 2)
    assert "--verify ok" in r.stderr, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["sharding"] == "clients x2"
    assert line["config"]["units_per_step"] == 2 * 2 * 4


def test_bench_two_ranks_sharded_by_ciphertext_index():
    """every rank holds all clients' keys and half of the ciphertext indices; no collective in the data path."""
    import json
    r = _run_bench(["--shard", "ct"], {}, 2)
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["sharding"] == "ciphertext index x2"
    assert line["config"]["units_per_step"] == 2 * 2 * 4 and line["config"]["clients_per_gpu"] == 4
    assert "no collective" in line["config"]["workload"]


def _chacha20_block_py(key, counter, nonce):
    """RFC 8439 2.3 in plain Python (test-side restatement of the generator)."""
    def rotl(v, c):
        return ((v << c) & 0xFFFFFFFF) | (v >> (32 - c))

    def qr(x, a, b, c, d):
        x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] = rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & 0xFFFFFFFF; x[b] = rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] = rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & 0xFFFFFFFF; x[b] = rotl(x[b] ^ x[c], 7)

    s = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + [int.from_bytes(key[4 * i:4 * i + 4], "little") for i in range(8)]
    s += [counter] + list(nonce)
    x = list(s)
    for _ in range(10):
        qr(x, 0, 4, 8, 12); qr(x, 1, 5, 9, 13); qr(x, 2, 6, 10, 14); qr(x, 3, 7, 11, 15)
        qr(x, 0, 5, 10, 15); qr(x, 1, 6, 11, 12); qr(x, 2, 7, 8, 13); qr(x, 3, 4, 9, 14)
    return [(a + b) & 0xFFFFFFFF for a, b in zip(x, s)]


def test_sampler_generator_is_chacha20(ctxs):
    """The device samplers run the ChaCha20 block function: RFC 8439 2.3.2 known answer, and the documented mapping
    element i -> 64-bit word i % 8 of block i / 8 (nonce = (block >> 32, stream, attempt)) for the ternary sampler."""
    g, _ = ctxs("tiny")
    key = bytes(range(32))
    got = g.chacha20_block(key, 1, (0x09000000, 0x4a000000, 0x00000000))
    want = [0xe4e7f110, 0x15593bd1, 0x1fdd0f50, 0xc47120a3, 0xc7f4d1c7, 0x0368c033, 0x9aaa2204, 0x4e6cd4c3,
            0x466482d2, 0x09aa9f07, 0x05d7c214, 0xa2028bd9, 0xd19c12b5, 0xb94e16de, 0xe883d0cb, 0x4e3c50a2]
    assert [int(v) for v in got] == want
    assert _chacha20_block_py(key, 1, (0x09000000, 0x4a000000, 0)) == want
    n, sid = 40, 5
    d_t = g.empty((n,), dtype=np.int8)
    g.sample_ternary(d_t, n, key, sid)
    t = d_t.to_host()
    for i in range(n):
        blk = _chacha20_block_py(key, i // 8, (0, sid, 0))
        w = i % 8
        r = (blk[2 * w + 1] << 32) | blk[2 * w]
        assert int(t[i]) == ((r * 3) >> 64) - 1, i
    from ppqsflhe_amd import MkckksError
    with pytest.raises(ValueError):
        g.sample_ternary(d_t, n, b"short", 0)
