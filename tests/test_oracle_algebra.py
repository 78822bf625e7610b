"""Algebraic checks of the oracle against exact Python big-integer arithmetic (small N, no GPU)."""
import numpy as np
import pytest

from oracle.oracle import sample_gauss, sample_ternary, sample_uniform


def _negacyclic_schoolbook(a, b, q):
    n = len(a)
    out = [0] * n
    for i in range(n):
        ai = int(a[i])
        if ai == 0:
            continue
        for j in range(n):
            k = i + j
            t = ai * int(b[j])
            if k >= n:
                out[k - n] = (out[k - n] - t) % q
            else:
                out[k] = (out[k] + t) % q
    return np.array(out, dtype=np.uint64)


def test_ntt_is_negacyclic_convolution():
    # schoolbook negacyclic product at N <= 256
    from oracle.oracle import OracleContext
    ctx = OracleContext(8, 1, 40, 60, dnum=2)
    rng = np.random.default_rng(3)
    for limb in range(ctx.D):
        q = int(ctx.moduli[limb])
        a = rng.integers(0, q, size=ctx.N, dtype=np.uint64)
        b = rng.integers(0, q, size=ctx.N, dtype=np.uint64)
        fa, fb = ctx.ntt_fwd(limb, a), ctx.ntt_fwd(limb, b)
        prod = np.array([(int(x) * int(y)) % q for x, y in zip(fa, fb)], dtype=np.uint64)
        assert np.array_equal(ctx.ntt_inv(limb, prod), _negacyclic_schoolbook(a, b, q))
        assert np.array_equal(ctx.ntt_inv(limb, fa), a)


def test_ntt_eval_points(small_ctx):
    # P4: X[i] = s(psi^(2*bitrev(i)+1))
    ctx = small_ctx
    logn = ctx.N.bit_length() - 1
    rng = np.random.default_rng(4)
    limb = 1
    q, psi = int(ctx.moduli[limb]), int(ctx.roots[limb])
    a = rng.integers(0, q, size=ctx.N, dtype=np.uint64)
    fa = ctx.ntt_fwd(limb, a)
    for i in (0, 1, 2, 5, ctx.N - 1):
        br = int(format(i, f"0{logn}b")[::-1], 2)
        x = pow(psi, 2 * br + 1, q)
        acc = 0
        for c in reversed(a.tolist()):
            acc = (acc * x + int(c)) % q
        assert acc == int(fa[i])


def _crt(res, mods):
    Q = 1
    for m in mods:
        Q *= int(m)
    x = 0
    for r, m in zip(res, mods):
        m = int(m)
        Qi = Q // m
        x += int(r) * Qi * pow(Qi, -1, m)
    return x % Q, Q


def test_rescale_exact(small_ctx):
    # DropLastElementAndScale == (x - [x]_ql centred) / ql exactly, coefficient-wise
    ctx = small_ctx
    rng = np.random.default_rng(5)
    nl = ctx.L
    mods = [int(m) for m in ctx.moduli[:nl]]
    coef = np.stack([rng.integers(0, m, size=ctx.N, dtype=np.uint64) for m in mods])
    ev = np.stack([ctx.ntt_fwd(i, coef[i]) for i in range(nl)])
    ct = np.stack([ev, ev])
    out = ctx.rescale(ct)
    oc = np.stack([ctx.ntt_inv(i, out[0, i]) for i in range(nl - 1)])
    ql = mods[-1]
    for j in (0, 1, 17, ctx.N - 1):
        x, Q = _crt(coef[:, j], mods)
        r = x % ql
        if r > ql // 2:
            r -= ql
        y = (x - r) // ql
        assert (x - r) % ql == 0
        for i in range(nl - 1):
            assert y % mods[i] == int(oc[i, j])
    assert np.array_equal(out[0], out[1])


def test_const_factors(ref_ctx):
    # EvalMult(ct, 0.5): constant = floor(0.5*sf(1) + 0.5) reduced per limb
    f = ref_ctx.const_factors(3, 1, 0.5)
    c = int(0.5 * ref_ctx.sf(1) + 0.5)
    assert [int(x) for x in f] == [c % int(m) for m in ref_ctx.moduli[:3]]
    assert ref_ctx.sf(1) == float(ref_ctx.moduli[2])
    assert ref_ctx.sf(0) == float(ref_ctx.moduli[3])
    f = ref_ctx.const_factors(3, 1, -0.25)
    c = int(-0.25 * ref_ctx.sf(1) + 0.5)
    assert [int(x) for x in f] == [c % int(m) for m in ref_ctx.moduli[:3]]


@pytest.mark.parametrize("nl_drop", [0, 1, 2])
def test_modup_digits_are_congruent(small_ctx, nl_drop):
    # each ModUp digit d_j equals (its own limbs' CRT value + e*Q_j), 0 <= e < size, on every limb
    ctx = small_ctx
    nl = ctx.L - nl_drop
    rng = np.random.default_rng(6)
    coef = np.stack([rng.integers(0, int(ctx.moduli[i]), size=ctx.N, dtype=np.uint64) for i in range(nl)])
    c1 = np.stack([ctx.ntt_fwd(i, coef[i]) for i in range(nl)])
    dig = ctx.modup_digits(c1)
    nparts = -(-nl // ctx.alpha)
    assert dig.shape == (nparts, nl + ctx.K, ctx.N)
    ext_idx = list(range(nl)) + list(range(ctx.L, ctx.D))
    for part in range(nparts):
        lo, hi = part * ctx.alpha, min(nl, (part + 1) * ctx.alpha)
        own = [int(ctx.moduli[i]) for i in range(lo, hi)]
        for j in (0, 3, ctx.N - 1):
            x, Qj = _crt(coef[lo:hi, j], own)
            for pos, limb in enumerate(ext_idx):
                m = int(ctx.moduli[limb])
                got = int(ctx.ntt_inv(limb, dig[part, pos])[j])
                if lo <= pos < hi:
                    assert got == int(coef[pos, j])
                else:
                    assert any((x + e * Qj) % m == got for e in range(hi - lo + 1))


def test_moddown_divides_by_p(small_ctx):
    # ApproxModDown(P * y) == y up to the approximate-conversion slack (exact here: x = P*y has zero P part)
    ctx = small_ctx
    rng = np.random.default_rng(8)
    nl = ctx.L
    P = 1
    for p in ctx.moduli[ctx.L:]:
        P *= int(p)
    y = np.stack([rng.integers(0, int(ctx.moduli[i]), size=ctx.N, dtype=np.uint64) for i in range(nl)])
    x = np.zeros((nl + ctx.K, ctx.N), dtype=np.uint64)
    for i in range(nl):
        m = int(ctx.moduli[i])
        x[i] = np.array([(int(v) * (P % m)) % m for v in y[i]], dtype=np.uint64)
    assert np.array_equal(ctx.moddown(x), y)


@pytest.mark.parametrize("nl_drop", [0, 1, 3])
def test_reencrypt_noise_and_levels(small_ctx, nl_drop):
    # Dec_new(PRE(ct)) - Dec_old(ct) is small (SURVEY.md 8a "formula check"), at full and reduced levels
    # (reduced levels exercise the partial last digit and the evk limb-skipping of EvalFastKeySwitchCoreExt)
    ctx = small_ctx
    rng = np.random.default_rng(9)
    N = ctx.N
    s1 = sample_ternary(rng, N)
    pk1, sk1 = ctx.keygen(s1, sample_uniform(rng, ctx.moduli, N), sample_gauss(rng, N))
    s2 = sample_ternary(rng, N)
    pk2, sk2 = ctx.keygen(s2, sample_uniform(rng, ctx.moduli, N), sample_gauss(rng, N))
    u = np.stack([sample_ternary(rng, N) for _ in range(ctx.beta)])
    e0 = np.stack([sample_gauss(rng, N) for _ in range(ctx.beta)])
    e1 = np.stack([sample_gauss(rng, N) for _ in range(ctx.beta)])
    evk = ctx.rekeygen(s1, pk2, u, e0, e1)
    nl = ctx.L - nl_drop
    vals = rng.uniform(-0.3, 0.3, size=N // 2)
    scale = 2.0 ** 35
    pt = ctx.encode(vals, scale, nl)
    ct = ctx.encrypt(pk1, pt, sample_ternary(rng, N), sample_gauss(rng, N), sample_gauss(rng, N))
    out = ctx.reencrypt(ct, evk)
    m_old = ctx.decrypt_core(ct, sk1)
    m_new = ctx.decrypt_core(out, sk2)
    for i in range(nl):
        q = int(ctx.moduli[i])
        d = (m_new[i].astype(object) - m_old[i].astype(object)) % q
        d = np.array([x - q if x > q // 2 else x for x in d], dtype=object)
        assert max(abs(int(x)) for x in d) < 2 ** 14, "key-switch noise must stay small"
    dec = ctx.decrypt_decode(out, sk2, scale)
    assert np.abs(dec - vals).max() < 1e-5


@pytest.mark.parametrize("cfg", [(10, 3, 40, 60, 2), (14, 2, 40, 60, 2), (12, 18, 50, 60, 3), (14, 2, 40, 55, 2)])
def test_fast_cpu_reencrypt_equals_the_restatement(cfg):
    """oracle/cpu_fast.c (what bench.py's cpu_baseline leg times) carries out the restatement's ReEncrypt with lazy
    butterflies and precomputed constants: every output word must equal orc_reencrypt's, at full level, after dropped
    limbs, with a partial last digit and with a single limb; and its transforms must equal the restatement's."""
    from oracle.oracle import FastCpuContext, OracleContext
    o = OracleContext(*cfg[:4], dnum=cfg[4])
    f = FastCpuContext(*cfg[:4], dnum=cfg[4])
    assert np.array_equal(o.moduli, f.moduli)
    rng = np.random.default_rng(5)
    evk = np.empty((o.beta, 2, o.D, o.N), dtype=np.uint64)
    for l in range(o.D):
        evk[:, :, l, :] = rng.integers(0, int(o.moduli[l]), (o.beta, 2, o.N), dtype=np.uint64)
    for nl in sorted({o.L, o.L - 1, o.alpha + 1, 2, 1}):
        if not 1 <= nl <= o.L:
            continue
        ct = np.stack([np.stack([rng.integers(0, int(o.moduli[l]), o.N, dtype=np.uint64) for l in range(nl)])
                       for _ in range(2)])
        ct[0, 0, ::3] = int(o.moduli[0]) - 1  # boundary residues
        assert np.array_equal(o.reencrypt(ct, evk), f.reencrypt(ct, evk)), (cfg, nl)
    for limb in (0, o.L - 1, o.D - 1):
        x = rng.integers(0, int(o.moduli[limb]), o.N, dtype=np.uint64)
        x[:4] = int(o.moduli[limb]) - 1
        assert np.array_equal(o.ntt_fwd(limb, x), f.ntt_fwd(limb, x))
        assert np.array_equal(o.ntt_inv(limb, x), f.ntt_inv(limb, x))
